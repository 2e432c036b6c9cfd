#!/bin/bash
# round 4, second GPU call: line-search variants of the leap kernel on recorded inputs, census, parity suites, small-config bench lines with the one-launch update
cd $GRAFT_REPO_ROOT; out=gpurun_out/r4b; mkdir -p $out
REPS=2 bash tools/gpu/ab4.sh base product k75 rev10 base > $out/ab.txt 2>&1; cat $out/ab.txt
JUDO_AMD_LIB=$PWD/variants/libjudo_amd_census.so python tools/diag/census_v5.py 2,32 > $out/census.txt 2>&1; tail -8 $out/census.txt
for t in cartpole cylinder_push; do python bench.py --task $t --no-cpu-baseline > $out/bench_$t.json 2> $out/bench_$t.err; python - <<PY
import json; d=json.loads(open("$out/bench_$t.json").read().strip().splitlines()[-1]); print("$t", d["ms_per_step"], {k: d[k] for k in d if "trace" in k or "kernel" in k})
PY
done
timeout 1500 python -m pytest tests/test_gpu_controller.py tests/test_gpu_leap.py tests/test_gpu_leap_self.py tests/test_gpu_edges.py tests/test_gpu_fr3.py -x -q -m gpu > $out/pytest.txt 2>&1; tail -8 $out/pytest.txt

#!/bin/bash
cd $GRAFT_REPO_ROOT; out=gpurun_out/r4d; mkdir -p $out
JUDO_AMD_LIB=$PWD/variants/libjudo_amd_ticks.so python tools/diag/profile_v5b.py > $out/ticks.txt 2>&1; cat $out/ticks.txt
python tools/diag/host_profile.py cartpole 4096 > $out/host_profile_cartpole.txt 2>&1; head -22 $out/host_profile_cartpole.txt
python tools/diag/host_profile.py cylinder_push 16384 2>&1 | head -3
for t in cartpole cylinder_push; do python bench.py --task $t --no-cpu-baseline > $out/bench_$t.json 2> $out/bench_$t.err; python - <<PY
import json; d=json.loads(open("$out/bench_$t.json").read().strip().splitlines()[-1]); print("$t", d["ms_per_step"], d.get("roofline", {}).get("kernel_ms"))
PY
done
timeout 900 python -m pytest tests/test_gpu_controller.py tests/test_gpu_simple.py tests/test_gpu_dist.py -x -q -m gpu > $out/pytest.txt 2>&1; tail -5 $out/pytest.txt

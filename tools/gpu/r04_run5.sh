#!/bin/bash
cd $GRAFT_REPO_ROOT; out=gpurun_out/r4e; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_controller.py tests/test_gpu_simple.py tests/test_gpu_dist.py tests/test_gpu_edges.py -x -q -m gpu > $out/pytest.txt 2>&1; tail -5 $out/pytest.txt
REPS=2 bash tools/gpu/ab4.sh product gat rcp both product > $out/ab.txt 2>&1; cat $out/ab.txt
python tools/diag/host_profile.py cartpole 4096 > $out/host_profile_cartpole.txt 2>&1; head -16 $out/host_profile_cartpole.txt
python tools/diag/host_profile.py cylinder_push 16384 2>&1 | head -3
for t in cartpole cylinder_push; do python bench.py --task $t --no-cpu-baseline --steps 200 > $out/bench_$t.json 2> $out/bench_$t.err; python - <<PY
import json; d=json.loads(open("$out/bench_$t.json").read().strip().splitlines()[-1]); print("$t", d["ms_per_step"], d["plan_step_ms"], d.get("roofline", {}).get("kernel_ms"), d["per_rank"])
PY
done

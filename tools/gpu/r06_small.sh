#!/bin/bash
# GPU box: the closed-form models' plan step as one launch (k_plan_step) against the two launches it replaces (JUDO_AMD_PLAN_STEP_LAUNCHES=2): tests, then the bench lines of both forms
cd $GRAFT_REPO_ROOT
out=gpurun_out/r6_small; rm -rf $out; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_controller.py tests/test_gpu_simple.py tests/test_gpu_dist.py -x -q -m gpu -p no:cacheprovider > $out/pytest.txt 2>&1; echo "pytest rc $?" >> $out/pytest.txt; tail -n 5 $out/pytest.txt
for L in 1 2 1 2; do
  for t in cartpole cylinder_push; do
    JUDO_AMD_PLAN_STEP_LAUNCHES=$L python bench.py --task $t --steps 500 --warmup 20 --no-cpu-baseline > $out/bench_${t}_L$L.json 2> $out/bench_${t}_L$L.log
    python - <<PY
import json
d=json.loads([l for l in open("$out/bench_${t}_L$L.json") if l.startswith("{")][-1])
p=d["per_rank"][0]
print("$t launches=$L", "ms_per_step %.4f" % d["ms_per_step"], "median %.4f" % d["plan_step_ms"]["median"], "kernel %.4f exchange %.4f host+launch %.4f" % (p["kernel_ms"], p["exchange_ms"], p["host_and_launch_ms"]))
PY
  done
done 2>&1 | tee $out/summary.txt

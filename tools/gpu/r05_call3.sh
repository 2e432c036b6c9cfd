#!/bin/bash
# GPU box, round 5, call 3: Spot tests on the optimised broad phase; phase clocks with / without the robot's own pairs; pair order A/B; two ranks on one GPU with small kernels
cd $GRAFT_REPO_ROOT
out=gpurun_out/r5c; rm -rf $out; mkdir -p $out
rm -f gpurun_out/test_margins.jsonl
JUDO_RECORD_MARGINS=1 timeout 900 python -m pytest tests/test_gpu_spot.py tests/test_gpu_policy.py -m gpu -q -W error::RuntimeWarning -p no:cacheprovider > $out/pytest.txt 2>&1; echo "pytest rc $?" >> $out/pytest.txt
tail -n 8 $out/pytest.txt
cp gpurun_out/test_margins.jsonl $out/ 2>/dev/null
for cfg in "1 grouped" "1 plain" "0 grouped"; do
  set -- $cfg
  echo "== SELF=$1 pair order $2" | tee -a $out/time_spot.txt
  SELF=$1 JUDO_AMD_TREE_PAIR_ORDER=$2 timeout 300 python tools/diag/time_spot.py 65536 10 2>&1 | grep -v amdgpu.ids | tee -a $out/time_spot.txt | cut -c1-200
done
echo "== phase clocks (-DJH_V4_PHASES), SELF=1 then SELF=0" | tee -a $out/phases.txt
for sf in 1 0; do SELF=$sf JUDO_AMD_LIB=$PWD/variants/libjudo_amd_v4ph.so timeout 300 python tools/diag/time_spot.py 65536 10 2>&1 | grep -v amdgpu.ids | tee -a $out/phases.txt | cut -c1-160; done
for t in cartpole; do
  JUDO_BENCH_SHARED_GPU=1 timeout 600 python bench.py --gpus 2 --task $t --rollouts 8192 --steps 50 --no-cpu-baseline --no-cube-only --no-steady-state --no-replay --no-with-traces > $out/bench_two_ranks_$t.json 2> $out/bench_two_ranks_$t.log
  python - <<PY
import json
try:
    d=json.loads([l for l in open("$out/bench_two_ranks_$t.json") if l.startswith("{")][-1]); print("two ranks $t:", d["ms_per_step"], d["per_rank"])
except Exception as e: print("failed", e)
PY
done
timeout 300 python tools/diag/time_policy_small.py 2>&1 | grep -v amdgpu.ids | tee $out/policy_small.txt | tail -n 8

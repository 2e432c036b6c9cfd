#!/bin/bash
# GPU box, round 5, call 1: full GPU suite with margin recording and RuntimeWarnings as errors; leap A/B (c3 cache, trace rows); elite re-roll cost; two ranks on one GPU
cd $GRAFT_REPO_ROOT
out=gpurun_out/r5a; rm -rf $out; mkdir -p $out
rm -f gpurun_out/test_margins.jsonl
JUDO_RECORD_MARGINS=1 timeout 900 python -m pytest tests -m gpu -q -W error::RuntimeWarning -p no:cacheprovider > $out/pytest.txt 2>&1; echo "pytest rc $?" >> $out/pytest.txt
tail -n 15 $out/pytest.txt
cp gpurun_out/test_margins.jsonl $out/ 2>/dev/null
R="python tools/diag/ab_fixed_inputs.py replay tools/diag/ab_inputs_leap.npz"
for v in product c3 notrace c3notrace product c3; do
  lib=$PWD/variants/libjudo_amd_$v.so; [ $v == product ] && lib=$PWD/judo_amd/libjudo_amd.so
  echo "== $v: $(JUDO_AMD_LIB=$lib OUT=$out/nom_$v.npy timeout 300 $R 2>&1 | tail -1 | sed 's/.*kernel mean/kernel mean/')" | tee -a $out/ab.txt
done
echo "== product FUSED_TRACES=0: $(FUSED_TRACES=0 timeout 300 $R 2>&1 | tail -1 | sed 's/.*kernel mean/kernel mean/')" | tee -a $out/ab.txt
python - <<'PY' 2>&1 | tee -a $out/ab.txt
import numpy as np
a=np.load("gpurun_out/r5a/nom_product.npy"); b=np.load("gpurun_out/r5a/nom_c3.npy")
print("c3 vs product nominals: bit-identical" if np.array_equal(a,b) else f"c3 vs product nominals differ: max {np.abs(a-b).max():.3e}")
PY
timeout 300 python tools/diag/time_elite_reroll.py 2>&1 | tee $out/elite_reroll.txt
for n in 1 2; do
  JUDO_BENCH_SHARED_GPU=1 timeout 600 python bench.py --gpus $n --rollouts $((8192*n)) --steps 20 --no-cpu-baseline --no-cube-only --no-steady-state --no-replay --no-with-traces > $out/bench_shared_$n.json 2> $out/bench_shared_$n.log
  python - <<PY
import json
try:
    d=json.loads([l for l in open("$out/bench_shared_$n.json") if l.startswith("{")][-1]); print("ranks $n:", d["ms_per_step"], d["per_rank"])
except Exception as e: print("ranks $n failed", e)
PY
done
timeout 600 python bench.py --no-cpu-baseline > $out/bench_default.json 2> $out/bench_default.log; tail -c 1500 $out/bench_default.json

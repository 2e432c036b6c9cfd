#!/bin/bash
# GPU box: A/B of leap (or TASK=fr3_pick) kernel variants on the recorded plan inputs.  usage: tools/gpu/r06_ab.sh <outdir> variant...   (product = the shipped library)
cd $GRAFT_REPO_ROOT
out=gpurun_out/$1; shift; mkdir -p $out
task=${TASK:-leap_cube}; inp=tools/diag/ab_inputs_leap.npz; [ $task == fr3_pick ] && inp=tools/diag/ab_inputs_fr3.npz
R="python tools/diag/ab_fixed_inputs.py replay $inp $task"
for v in "$@"; do
  lib=$PWD/variants/libjudo_amd_$v.so; [ $v == product ] && lib=$PWD/judo_amd/libjudo_amd.so
  echo "== $v: $(JUDO_AMD_LIB=$lib OUT=$out/nom_$v.npy timeout 300 $R 2>&1 | tail -1 | sed 's/.*lstol/lstol/')" | tee -a $out/ab.txt
done
python - "$out" "$@" <<'PY' 2>&1 | tee -a $out/ab.txt
import sys, numpy as np
out, vs = sys.argv[1], sys.argv[2:]
a = np.load(f"{out}/nom_product.npy") if "product" in vs else None
for v in vs:
    if v == "product" or a is None: continue
    b = np.load(f"{out}/nom_{v}.npy")
    print(f"{v} vs product nominals: " + ("bit-identical" if np.array_equal(a, b) else f"max |diff| {np.abs(a - b).max():.3e}, median {np.median(np.abs(a - b)):.2e}"))
PY

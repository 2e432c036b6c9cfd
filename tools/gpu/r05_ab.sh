#!/bin/bash
# GPU box: A/B of leap variants on the recorded plan inputs.  usage: tools/gpu/r05_ab.sh <outdir> variant...   (product = the shipped library)
cd $GRAFT_REPO_ROOT
out=gpurun_out/$1; shift; mkdir -p $out
R="python tools/diag/ab_fixed_inputs.py replay tools/diag/ab_inputs_leap.npz"
for v in "$@"; do
  lib=$PWD/variants/libjudo_amd_$v.so; [ $v == product ] && lib=$PWD/judo_amd/libjudo_amd.so
  echo "== $v: $(JUDO_AMD_LIB=$lib OUT=$out/nom_$v.npy timeout 300 $R 2>&1 | tail -1 | sed 's/.*Newton cap/Newton cap/')" | tee -a $out/ab.txt
done
python - "$out" "$@" <<'PY' 2>&1 | tee -a $out/ab.txt
import sys, numpy as np
out, vs = sys.argv[1], sys.argv[2:]
a = np.load(f"gpurun_out/{out.split('/')[-1]}/nom_product.npy") if "product" in vs else None
for v in vs:
    if v == "product" or a is None: continue
    b = np.load(f"gpurun_out/{out.split('/')[-1]}/nom_{v}.npy")
    print(f"{v} vs product nominals: " + ("bit-identical" if np.array_equal(a, b) else f"max |diff| {np.abs(a - b).max():.3e}"))
PY

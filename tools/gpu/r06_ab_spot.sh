#!/bin/bash
# GPU box: A/B of library variants on the seeded Spot policy rollouts (tools/diag/ab_spot.py).  usage: tools/gpu/r06_ab_spot.sh <outdir> variant...   (product = the shipped library)
cd $GRAFT_REPO_ROOT
out=gpurun_out/$1; shift; mkdir -p $out
for v in "$@"; do
  lib=$PWD/variants/libjudo_amd_$v.so; [ $v == product ] && lib=$PWD/judo_amd/libjudo_amd.so
  echo "== $v" | tee -a $out/ab.txt
  JUDO_AMD_LIB=$lib OUT=$out/s_$v.npy timeout 600 python tools/diag/ab_spot.py 2>&1 | grep -v amdgpu.ids | tee -a $out/ab.txt
done
python - "$out" "$@" <<'PY' 2>&1 | tee -a $out/ab.txt
import sys, numpy as np
out, vs = sys.argv[1], sys.argv[2:]
a = np.load(f"{out}/s_product.npy") if "product" in vs else None
for v in dict.fromkeys(vs):
    if v == "product" or a is None: continue
    b = np.load(f"{out}/s_{v}.npy")
    print(f"{v} vs product states: " + ("bit-identical" if np.array_equal(a, b) else f"max |diff| {np.abs(a - b).max():.3e}, median {np.median(np.abs(a - b)):.2e}"))
PY

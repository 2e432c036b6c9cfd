#!/bin/bash
# GPU box: Spot tree kernel variants (compiler scheduling).  usage: tools/gpu/r05_spot_ab.sh outdir variant...
cd $GRAFT_REPO_ROOT
out=gpurun_out/$1; shift; mkdir -p $out
for v in "$@"; do
  lib=$PWD/variants/libjudo_amd_$v.so; [ $v == product ] && lib=$PWD/judo_amd/libjudo_amd.so
  echo "== $v: $(JUDO_AMD_LIB=$lib timeout 300 python tools/diag/time_spot.py 65536 10 2>&1 | grep -v amdgpu.ids | cut -c1-110 | tr '\n' ' ')" | tee -a $out/spot_ab.txt
  echo "   $v N=24: $(JUDO_AMD_LIB=$lib timeout 300 python tools/diag/time_policy_small.py 2>&1 | grep -v amdgpu.ids | tail -1)" | tee -a $out/spot_ab.txt
done

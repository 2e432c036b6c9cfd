#!/bin/bash
# GPU box: A/B of fr3 variants on the recorded plan inputs.  usage: tools/gpu/r05_ab_fr3.sh <outdir> variant...
cd $GRAFT_REPO_ROOT
out=gpurun_out/$1; shift; mkdir -p $out
R="python tools/diag/ab_fixed_inputs.py replay tools/diag/ab_inputs_fr3.npz fr3_pick"
for v in "$@"; do
  lib=$PWD/variants/libjudo_amd_$v.so; [ $v == product ] && lib=$PWD/judo_amd/libjudo_amd.so
  echo "== $v: $(JUDO_AMD_LIB=$lib REPS=2 timeout 300 $R 2>&1 | tail -1 | sed 's/.*Newton cap/Newton cap/' | cut -c1-260)" | tee -a $out/ab_fr3.txt
done

#!/bin/bash
# GPU box: Newton tolerance x contact-pool sweep on the recorded 40 plan steps of the headline workload (hand self-collision on).
# Reference = tol 1e-6 with the 48-contact pool (build/libjudo_amd_ns3.so = tools/build_variant.sh ns3 -DJH_V5_NSLOT=3).
cd $GRAFT_REPO_ROOT
R="python tools/diag/ab_fixed_inputs.py replay tools/diag/ab_inputs_leap.npz"
mkdir -p gpurun_out/sweep
exec > >(grep -v amdgpu.ids | tee gpurun_out/sweep/sweep.txt) 2>&1
JUDO_AMD_LIB=$PWD/build/libjudo_amd_ns3.so SELF=1 TOL=1e-6 OUT=gpurun_out/sweep/ref.npy $R 2>&1 | tail -2
for tol in 1e-3 1e-4 1e-5 1e-6; do
  echo "== pool 32, tol $tol"; SELF=1 TOL=$tol REF=gpurun_out/sweep/ref.npy $R 2>&1 | tail -3 | cut -c1-400
  echo "== pool 48, tol $tol"; JUDO_AMD_LIB=$PWD/build/libjudo_amd_ns3.so SELF=1 TOL=$tol REF=gpurun_out/sweep/ref.npy $R 2>&1 | tail -3 | cut -c1-400
done
echo "== cube contacts only, pool 32"; for tol in 1e-3 1e-4 1e-6; do SELF=0 TOL=$tol $R 2>&1 | tail -1; done

#!/bin/bash
# GPU box: where the one-workgroup-per-rollout policy launch stops paying against the per-layer launches
cd $GRAFT_REPO_ROOT
out=gpurun_out/r5r2; rm -rf $out; mkdir -p $out
for n in 64 96 128 192 256 512; do
  for mx in 0 1000000; do echo "rows_max=$mx $(JUDO_AMD_POLICY_ROWS_MAX=$mx timeout 300 python tools/diag/time_policy_small.py $n 2>&1 | grep -v amdgpu.ids | tail -n 1)" | tee -a $out/cross.txt; done
done

#!/bin/bash
# GPU box: where the per-layer policy launches stop paying: policy step latency with the switch-over forced on / off
cd $GRAFT_REPO_ROOT
out=gpurun_out/r5q; rm -rf $out; mkdir -p $out
for n in 512 1024 2048 4096 8192 16384 65536; do
  for mx in 0 1000000; do echo "layers_max=$mx $(JUDO_AMD_POLICY_LAYERS_MAX=$mx timeout 300 python tools/diag/time_policy_small.py $n 2>&1 | grep -v amdgpu.ids | tail -n 1)" | tee -a $out/cross.txt; done
done

#!/bin/bash
# GPU box: the fused policy launch (above 2 048 rollouts) after a change: Spot + policy tests (bit identity with the per-layer launches), its latency at 4 096 ... 65 536 rollouts
cd $GRAFT_REPO_ROOT
out=gpurun_out/r5b; rm -rf $out; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_spot.py tests/test_gpu_policy.py -m gpu -q -W error::RuntimeWarning -p no:cacheprovider -s > $out/pytest.txt 2>&1; echo "pytest rc $?" >> $out/pytest.txt
grep -a "policy step N=\|passed\|failed\|rc " $out/pytest.txt | tail -n 6
for n in 4096 8192 16384 65536; do timeout 300 python tools/diag/time_policy_small.py $n 2>&1 | grep -v amdgpu.ids | tee -a $out/policy_big.txt | tail -n 1; done
SELF=1 timeout 300 python tools/diag/time_spot.py 65536 10 2>&1 | grep -v amdgpu.ids | tee $out/time_spot.txt | cut -c1-120

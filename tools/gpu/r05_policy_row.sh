#!/bin/bash
# GPU box: the one-workgroup-per-rollout policy launch (up to 48 rollouts): Spot + policy tests (bit identity of the three paths), latency, the shipped Spot plan steps
cd $GRAFT_REPO_ROOT
out=gpurun_out/r5r; rm -rf $out; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_spot.py tests/test_gpu_policy.py -m gpu -q -W error::RuntimeWarning -p no:cacheprovider > $out/pytest.txt 2>&1; echo "pytest rc $?" >> $out/pytest.txt
tail -n 25 $out/pytest.txt | cut -c1-300
for n in 1 24 256 512 513 2048; do timeout 300 python tools/diag/time_policy_small.py $n 2>&1 | grep -v amdgpu.ids | tee -a $out/policy_row.txt | tail -n 1; done
timeout 600 python -m judo_amd.benchmark --tasks spot_navigate spot_base 2>&1 | grep -v amdgpu.ids | tee $out/sweep.txt | tail -n 12

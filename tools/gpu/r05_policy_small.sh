#!/bin/bash
# GPU box, round 5: the per-layer policy launches for small batches: Spot tests with margins, latency at 24 / 64 / 256 / 512 rollouts, the shipped Spot plan steps
cd $GRAFT_REPO_ROOT
out=gpurun_out/r5p; rm -rf $out; mkdir -p $out
rm -f gpurun_out/test_margins.jsonl
JUDO_RECORD_MARGINS=1 timeout 900 python -m pytest tests/test_gpu_spot.py tests/test_gpu_policy.py -m gpu -q -W error::RuntimeWarning -p no:cacheprovider > $out/pytest.txt 2>&1; echo "pytest rc $?" >> $out/pytest.txt
tail -n 15 $out/pytest.txt
cp gpurun_out/test_margins.jsonl $out/ 2>/dev/null
for n in 24 64 256 512; do timeout 300 python tools/diag/time_policy_small.py $n 2>&1 | grep -v amdgpu.ids | tee -a $out/policy_small.txt | tail -n 1; done
timeout 600 python -m judo_amd.benchmark --tasks spot_navigate spot_base 2>&1 | grep -v amdgpu.ids | tee $out/sweep.txt | tail -n 8
SELF=1 timeout 300 python tools/diag/time_spot.py 65536 10 2>&1 | grep -v amdgpu.ids | tee $out/time_spot.txt | cut -c1-200
timeout 300 python tools/diag/time_policy_small.py 65536 2>&1 | grep -v amdgpu.ids | tee -a $out/policy_small.txt | tail -n 1

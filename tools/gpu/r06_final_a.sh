#!/bin/bash
# GPU box, final build, part A: full GPU suite with margins, Spot timings, the round's profile exhibits (rocprofv3 stats + PMC passes -> profiles/r06_traffic.json via tools/collect_profiles.py)
cd $GRAFT_REPO_ROOT
out=gpurun_out/r6y; rm -rf $out; mkdir -p $out
rm -f gpurun_out/test_margins.jsonl
JUDO_RECORD_MARGINS=1 timeout 900 python -m pytest tests -m gpu -q -W error::RuntimeWarning -p no:cacheprovider > $out/pytest.txt 2>&1; echo "pytest rc $?" >> $out/pytest.txt
tail -n 8 $out/pytest.txt
cp gpurun_out/test_margins.jsonl $out/ 2>/dev/null
for sf in 1 0; do echo "== SELF=$sf" | tee -a $out/time_spot.txt; SELF=$sf timeout 300 python tools/diag/time_spot.py 65536 10 2>&1 | grep -v amdgpu.ids | tee -a $out/time_spot.txt | cut -c1-200; done
for n in 24 256 2048 65536; do timeout 300 python tools/diag/time_policy_small.py $n 2>&1 | grep -v amdgpu.ids | tee -a $out/policy_small.txt | tail -n 1; done
timeout 2400 bash tools/profile_round6.sh r06 > $out/profile.log 2>&1; tail -n 30 $out/profile.log | cut -c1-220

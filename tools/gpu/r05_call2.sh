#!/bin/bash
# GPU box, round 5, call 2: Spot robot self-collision -- per-state diagnostics, then the Spot / policy / dist tests, then throughput with and without it
cd $GRAFT_REPO_ROOT
out=gpurun_out/r5b; rm -rf $out; mkdir -p $out
timeout 300 python tools/diag/debug_spot_self.py > $out/debug_self.txt 2>&1; tail -n 30 $out/debug_self.txt
rm -f gpurun_out/test_margins.jsonl
JUDO_RECORD_MARGINS=1 timeout 900 python -m pytest tests/test_gpu_spot.py tests/test_gpu_policy.py tests/test_gpu_dist.py -m gpu -q -W error::RuntimeWarning -p no:cacheprovider > $out/pytest.txt 2>&1; echo "pytest rc $?" >> $out/pytest.txt
tail -n 25 $out/pytest.txt
cp gpurun_out/test_margins.jsonl $out/ 2>/dev/null
timeout 300 python tools/diag/time_spot.py > $out/time_spot.txt 2>&1; tail -n 12 $out/time_spot.txt
timeout 300 python bench.py --task spot_navigate --steps 3 --warmup 1 --no-cpu-baseline > $out/bench_spot.json 2> $out/bench_spot.log; tail -c 600 $out/bench_spot.json

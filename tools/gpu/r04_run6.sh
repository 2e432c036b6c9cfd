#!/bin/bash
cd $GRAFT_REPO_ROOT; out=gpurun_out/r4f; mkdir -p $out
REPS=2 bash tools/gpu/ab4.sh product hxlazy product > $out/ab.txt 2>&1; cat $out/ab.txt
python tools/diag/host_profile.py cartpole 4096 > $out/host_profile_cartpole.txt 2>&1; head -6 $out/host_profile_cartpole.txt
python tools/diag/host_profile.py cylinder_push 16384 2>&1 | head -5
timeout 1500 python -m pytest tests/test_gpu_leap.py tests/test_gpu_leap_self.py tests/test_gpu_controller.py tests/test_gpu_fr3.py -x -q -m gpu > $out/pytest.txt 2>&1; tail -5 $out/pytest.txt

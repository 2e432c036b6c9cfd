#!/bin/bash
cd $GRAFT_REPO_ROOT; out=gpurun_out/r4c; mkdir -p $out
REPS=2 bash tools/gpu/ab4.sh base product k75 rev10 base > $out/ab.txt 2>&1; cat $out/ab.txt
JUDO_AMD_LIB=$PWD/variants/libjudo_amd_census.so python tools/diag/census_v5.py 2,32 > $out/census.txt 2>&1; tail -8 $out/census.txt
python tools/diag/host_profile.py cartpole 4096 > $out/host_profile_cartpole.txt 2>&1; head -70 $out/host_profile_cartpole.txt
timeout 1500 python -m pytest tests/test_gpu_controller.py tests/test_gpu_leap.py tests/test_gpu_leap_self.py tests/test_gpu_edges.py tests/test_gpu_fr3.py -x -q -m gpu > $out/pytest.txt 2>&1; tail -8 $out/pytest.txt

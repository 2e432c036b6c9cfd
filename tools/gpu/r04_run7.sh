#!/bin/bash
cd $GRAFT_REPO_ROOT; out=gpurun_out/r4g; mkdir -p $out
JUDO_AMD_LIB=$PWD/variants/libjudo_amd_ticks.so python tools/diag/profile_v5b.py > $out/ticks.txt 2>&1; grep -v amdgpu.ids $out/ticks.txt | head -20
TASK=fr3_pick REPS=2 bash tools/gpu/ab4.sh v6old product v6old > $out/ab_fr3.txt 2>&1; cat $out/ab_fr3.txt
timeout 600 python -m pytest tests/test_gpu_fr3.py tests/test_gpu_controller.py -x -q -m gpu > $out/pytest.txt 2>&1; tail -3 $out/pytest.txt
bash tools/gpu/seeds4.sh base product > $out/seeds.txt 2>&1; cat $out/seeds.txt

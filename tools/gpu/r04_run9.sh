#!/bin/bash
cd $GRAFT_REPO_ROOT; out=gpurun_out/r4i; mkdir -p $out
REPS=2 bash tools/gpu/ab4.sh base2 schur link product base2 > $out/ab.txt 2>&1; cat $out/ab.txt
timeout 1500 python -m pytest tests/test_gpu_leap.py tests/test_gpu_leap_self.py tests/test_gpu_edges.py -x -q -m gpu > $out/pytest.txt 2>&1; tail -3 $out/pytest.txt

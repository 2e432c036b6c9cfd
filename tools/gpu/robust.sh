#!/bin/bash
# GPU box: robustness runs -- 300 closed-loop plan steps of the headline workload (and caltech), the jammed-state fuzz of the leap kernel
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/extras
{ python tools/diag/long_closed_loop.py leap_cube 2>&1 | tail -3; python tools/diag/long_closed_loop.py caltech_leap_cube 2>&1 | tail -3; python tools/diag/fuzz_leap.py 2>&1 | tail -8; } | tee gpurun_out/extras/robustness.txt

#!/bin/bash
# GPU box: the GPU suite, then plan-step time at small rollout counts with and without the latency mode (rows of a wave computing the same rollout)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/extras
timeout 1500 python -m pytest tests/ -m gpu -x -q 2>&1 | tail -4
{ JUDO_AMD_LATENCY_SHIFT=0 python tools/diag/time_small_n.py 2>&1 | tail -1; python tools/diag/time_small_n.py 2>&1 | tail -1; } | tee gpurun_out/extras/latency_mode.txt

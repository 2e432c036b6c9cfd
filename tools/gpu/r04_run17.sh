#!/bin/bash
cd $GRAFT_REPO_ROOT; out=gpurun_out/r4q; mkdir -p $out
REPS=2 bash tools/gpu/ab4.sh product > $out/ab.txt 2>&1; cat $out/ab.txt
TASK=fr3_pick REPS=2 bash tools/gpu/ab4.sh product v6o2 product > $out/ab_fr3.txt 2>&1; cat $out/ab_fr3.txt
timeout 1800 python -m pytest tests/ -m gpu -x -q 2>&1 | tail -4 | tee $out/pytest.txt

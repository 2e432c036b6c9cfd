#!/bin/bash
cd $GRAFT_REPO_ROOT; out=gpurun_out/r4p; mkdir -p $out
REPS=1 bash tools/gpu/ab4.sh product keepw keepw_o2 wpb8 wpb2 product > $out/ab.txt 2>&1; cat $out/ab.txt

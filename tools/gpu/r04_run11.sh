#!/bin/bash
cd $GRAFT_REPO_ROOT; out=gpurun_out/r4k; mkdir -p $out
REPS=2 bash tools/gpu/ab4.sh product noslp noslp_defsched product > $out/ab.txt 2>&1; cat $out/ab.txt

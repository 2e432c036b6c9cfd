#!/bin/bash
cd $GRAFT_REPO_ROOT; out=gpurun_out/r4l; mkdir -p $out
REPS=1 bash tools/gpu/ab4.sh n_base n_link n_lskink n_opq2 n_opq0 n_maxilp n_ns1 n_base > $out/ab.txt 2>&1; cat $out/ab.txt
TASK=fr3_pick REPS=2 bash tools/gpu/ab4.sh product v6noslp product > $out/ab_fr3.txt 2>&1; cat $out/ab_fr3.txt

#!/bin/bash
cd $GRAFT_REPO_ROOT; out=gpurun_out/r4n; mkdir -p $out
REPS=1 bash tools/gpu/ab4.sh b_base b_schur0 b_opqlane0 b_bigp b_opq3 b_link0 b_lskink b_opq4 b_base > $out/ab.txt 2>&1; cat $out/ab.txt

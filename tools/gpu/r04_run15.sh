#!/bin/bash
cd $GRAFT_REPO_ROOT; out=gpurun_out/r4o; mkdir -p $out
vs="product"; for v in lsv0 o2 novec nopost; do [ -f variants/libjudo_amd_$v.so ] && vs="$vs $v"; done
REPS=1 bash tools/gpu/ab4.sh $vs product > $out/ab.txt 2>&1; cat $out/ab.txt
TASK=fr3_pick REPS=2 bash tools/gpu/ab4.sh product > $out/ab_fr3.txt 2>&1; cat $out/ab_fr3.txt
timeout 1800 python -m pytest tests/ -m gpu -x -q 2>&1 | tail -4 | tee $out/pytest.txt

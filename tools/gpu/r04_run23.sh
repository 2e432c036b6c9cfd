#!/bin/bash
# GPU box: product with the world-frame assembly: full GPU suite + recorded inputs (leap, fr3).
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/run23
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 | tee gpurun_out/run23/pytest.txt
REPS=2 tools/gpu/ab4.sh product 2>&1 | tee gpurun_out/run23/ab.txt
for v in "$@"; do REPS=2 tools/gpu/ab4.sh $v 2>&1 | tee -a gpurun_out/run23/ab.txt; done

#!/bin/bash
# GPU box: product with the Schur complement in registers: full GPU suite + recorded inputs; fr3 phase split with the Hessian assembly apart.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/run22
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 | tee gpurun_out/run22/pytest.txt
REPS=2 tools/gpu/ab4.sh product 2>&1 | tee gpurun_out/run22/ab.txt
JUDO_AMD_LIB=$PWD/variants/libjudo_amd_v6ph.so python tools/diag/profile_fr3_phases.py 2>&1 | tee gpurun_out/run22/fr3_phases.txt

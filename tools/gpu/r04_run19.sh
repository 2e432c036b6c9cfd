#!/bin/bash
# GPU box: product with the merged pass: full GPU suite, recorded inputs, shader-clock split (ticks build).
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/run19
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 | tee gpurun_out/run19/pytest.txt
REPS=2 tools/gpu/ab4.sh product 2>&1 | tee gpurun_out/run19/ab.txt
JUDO_AMD_LIB=$PWD/variants/libjudo_amd_ticks.so python tools/diag/profile_v5b.py 2>&1 | tee gpurun_out/run19/ticks.txt | tail -40

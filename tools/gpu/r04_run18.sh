#!/bin/bash
# GPU box: merged gradient+Hessian pass of the leap kernel against the product, recorded inputs; leap parity tests on the variant.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/run18
REPS=2 tools/gpu/ab4.sh product merge merge_opq3 merge_opq1 merge_o3 product 2>&1 | tee gpurun_out/run18/ab.txt
JUDO_AMD_LIB=$PWD/variants/libjudo_amd_merge.so timeout 900 python -m pytest tests/test_gpu_leap.py tests/test_gpu_leap_self.py tests/test_gpu_leap_variants.py -m gpu -x -q 2>&1 | tail -5 | tee gpurun_out/run18/pytest_merge.txt

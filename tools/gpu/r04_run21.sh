#!/bin/bash
# GPU box: product with the cube-block row sums: full GPU suite; Schur complement in registers (bit-identical chain sums) on recorded inputs and the leap tests.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/run21
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 | tee gpurun_out/run21/pytest.txt
REPS=2 tools/gpu/ab4.sh product schur3 product 2>&1 | tee gpurun_out/run21/ab.txt
JUDO_AMD_LIB=$PWD/variants/libjudo_amd_schur3.so timeout 900 python -m pytest tests/test_gpu_leap.py tests/test_gpu_leap_self.py -m gpu -q --deselect tests/test_gpu_leap.py::test_leap_two_kernel_generations_agree 2>&1 | tail -4 | tee gpurun_out/run21/pytest_schur3.txt

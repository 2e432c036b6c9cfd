#!/bin/bash
# GPU box: level-2 broad phase with two body pairs per pass: recorded inputs, leap parity tests, pass counters; shader-clock split of the product.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/run24
REPS=2 tools/gpu/ab4.sh product l2dual product 2>&1 | tee gpurun_out/run24/ab.txt
JUDO_AMD_LIB=$PWD/variants/libjudo_amd_l2dual.so timeout 900 python -m pytest tests/test_gpu_leap.py tests/test_gpu_leap_self.py -m gpu -q --deselect tests/test_gpu_leap.py::test_leap_two_kernel_generations_agree 2>&1 | tail -5 | tee gpurun_out/run24/pytest_l2dual.txt
JUDO_AMD_LIB=$PWD/variants/libjudo_amd_l2dual_count.so python tools/diag/count_v5.py 2>&1 | grep "plan step" | tee gpurun_out/run24/count.txt
JUDO_AMD_LIB=$PWD/variants/libjudo_amd_ticks.so python tools/diag/profile_v5b.py 2>&1 | tee gpurun_out/run24/ticks.txt | tail -17

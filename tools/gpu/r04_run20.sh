#!/bin/bash
# GPU box: cube-block row sums, atomic cost probes (the chain-block atomics issued twice), Schur complement in registers; leap parity tests on the candidates.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/run20
REPS=2 tools/gpu/ab4.sh hccrow dupbb dupcb schur3 hccrow 2>&1 | tee gpurun_out/run20/ab.txt
for v in hccrow schur3; do
  JUDO_AMD_LIB=$PWD/variants/libjudo_amd_$v.so timeout 900 python -m pytest tests/test_gpu_leap.py tests/test_gpu_leap_self.py -m gpu -x -q 2>&1 | tail -4 | tee gpurun_out/run20/pytest_$v.txt
done

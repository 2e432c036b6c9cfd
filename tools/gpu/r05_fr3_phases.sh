#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5f
JUDO_AMD_LIB=$PWD/variants/libjudo_amd_v6ph.so timeout 300 python tools/diag/profile_fr3_phases.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5f/phases.txt

cd $GRAFT_REPO_ROOT
R="python tools/diag/ab_fixed_inputs.py replay tools/diag/ab_inputs_fr3.npz fr3_pick"
for v in "$@"; do echo "== $v: $(JUDO_AMD_LIB=$PWD/build/libjudo_amd_$v.so KERNEL_GEN=3 $R 2>&1 | tail -2 | tr '\n' ' ' | sed 's/.*contacts dropped/dropped/')"; done
JUDO_AMD_LIB=$PWD/build/libjudo_amd_$1.so JUDO_AMD_FR3_KERNEL=3 timeout 900 python -m pytest tests/test_gpu_fr3.py -x -q 2>&1 | tail -3

cd $GRAFT_REPO_ROOT
R="python tools/diag/ab_fixed_inputs.py replay tools/diag/ab_inputs_fr3.npz fr3_pick"
for v in "$@"; do echo "== $v: $(JUDO_AMD_LIB=$PWD/build/libjudo_amd_$v.so $R 2>&1 | tail -2 | tr '\n' ' ' | sed 's/.*contacts dropped/dropped/')"; done

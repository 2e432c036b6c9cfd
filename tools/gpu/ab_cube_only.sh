#!/bin/bash
# GPU box: A/B of library variants on the recorded leap_cube plan inputs, cube contacts only (occupancy experiments whose LDS budget has no room for the self-collision arrays).
cd $GRAFT_REPO_ROOT
R="python tools/diag/ab_fixed_inputs.py replay tools/diag/ab_inputs_leap.npz"
for v in "$@"; do
  echo "== $v self=0: $(JUDO_AMD_LIB=$PWD/build/libjudo_amd_$v.so SELF=0 $R 2>&1 | tail -2 | tr '\n' ' ' | sed 's/.*contacts dropped/dropped/; s/returned nominal.*//')"
done

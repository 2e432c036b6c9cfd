#!/bin/bash
# GPU box: A/B of library variants (variants/libjudo_amd_<name>.so; "product" = judo_amd/libjudo_amd.so) on the recorded plan inputs.
# usage: [TASK=leap_cube|fr3_pick] [REPS=2] [SELF=1] tools/gpu/ab4.sh name1 name2 ...
cd $GRAFT_REPO_ROOT
task=${TASK:-leap_cube}; inp=tools/diag/ab_inputs_leap.npz; [ "$task" == "fr3_pick" ] && inp=tools/diag/ab_inputs_fr3.npz
for v in "$@"; do
  lib=$PWD/variants/libjudo_amd_$v.so; [ "$v" == "product" ] && lib=$PWD/judo_amd/libjudo_amd.so
  echo "== $v: $(JUDO_AMD_LIB=$lib REPS=${REPS:-2} python tools/diag/ab_fixed_inputs.py replay $inp $task 2>&1 | grep -E 'kernel mean|dropped|Error|error' | sed 's/.*libjudo_amd_//' | tr '\n' ' ')"
done

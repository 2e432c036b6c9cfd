#!/bin/bash
cd $GRAFT_REPO_ROOT
for ft in 0 1; do for t in leap_cube fr3_pick; do
  f=tools/diag/ab_inputs_leap.npz; [ $t == fr3_pick ] && f=tools/diag/ab_inputs_fr3.npz
  echo "== $t fused_traces=$ft: $(FUSED_TRACES=$ft python tools/diag/ab_fixed_inputs.py replay $f $t 2>&1 | tail -1)"
done; done

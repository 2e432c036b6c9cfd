#!/bin/bash
# GPU box: random-state sweeps of the final build against the fp64 oracle (fr3 after the register Cholesky, leap, the plan-step update, the controller)
cd $GRAFT_REPO_ROOT
out=gpurun_out/r6u; rm -rf $out; mkdir -p $out
for f in fuzz_fr3 fuzz_leap fuzz_update fuzz_controller; do
  echo "== $f" | tee -a $out/fuzz.txt
  timeout 600 python -W error::RuntimeWarning tools/diag/$f.py 2>&1 | grep -v amdgpu.ids | tail -n 12 | tee -a $out/fuzz.txt | cut -c1-220
done

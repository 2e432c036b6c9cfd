cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
timeout 1500 python -m pytest tests/ -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r3/t1.log
cat gpurun_out/r3/t1.log
R="python tools/diag/ab_fixed_inputs.py replay tools/diag/ab_inputs_fr3.npz fr3_pick"
(echo "== fr3 r3e:"; $R 2>&1 | tail -2) | tee gpurun_out/r3/ab_fr3.log
R="python tools/diag/ab_fixed_inputs.py replay tools/diag/ab_inputs_leap.npz leap_cube"
(echo "== leap r3e:"; $R 2>&1 | tail -2) | tee gpurun_out/r3/ab_leap.log

#!/bin/bash
# GPU box: the default bench line over eight seeds of the noise stream (the closed loop decides which contact situations the plan visits)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/extras
for s in 1 2 3 4 5 6 7 1234; do
  python bench.py --seed $s --no-cpu-baseline --no-cube-only --no-with-traces 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('seed $s: %.2f ms per plan step (median %.2f, min %.2f, max %.2f), kernel %.2f ms, Newton %.2f it/step (wave %.2f), dropped %.1e' % (d['ms_per_step'], d['plan_step_ms']['median'], d['plan_step_ms']['min'], d['plan_step_ms']['max'], d['roofline']['kernel_ms'], d['solver']['newton_iters_per_step'], d['solver']['wave_newton_iters_per_step'], d['solver']['contacts_dropped_per_step']))"
done | tee gpurun_out/extras/seed_sweep.txt
python tools/diag/fuzz_leap.py 2>&1 | tail -9 | tee gpurun_out/extras/fuzz_leap.txt

#!/bin/bash
# GPU box: the reference's benchmark statistic (100 timed plan steps after 10 warm-ups) over eight seeds of the noise stream
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/extras
for s in 1 2 3 4 5 6 7 1234; do
  python bench.py --seed $s --steps 100 --warmup 10 --no-cpu-baseline --no-cube-only --no-with-traces 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('seed $s: %.2f ms per plan step (median %.2f, min %.2f, max %.2f), kernel %.2f ms, Newton %.2f it/step (wave %.2f), dropped %.1e' % (d['ms_per_step'], d['plan_step_ms']['median'], d['plan_step_ms']['min'], d['plan_step_ms']['max'], d['roofline']['kernel_ms'], d['solver']['newton_iters_per_step'], d['solver']['wave_newton_iters_per_step'], d['solver']['contacts_dropped_per_step']))"
done | tee gpurun_out/extras/seed_sweep_100.txt

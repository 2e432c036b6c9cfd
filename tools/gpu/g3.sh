cd $GRAFT_REPO_ROOT
for seed in 1234 1 2 3 4 5 6 7; do python bench.py --seed $seed --no-cpu-baseline --no-cube-only --no-with-traces 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('seed $seed', round(d['ms_per_step'],2), 'median', round(d['plan_step_ms']['median'],2), 'min', round(d['plan_step_ms']['min'],1), 'max', round(d['plan_step_ms']['max'],1), 'iters', round(d['solver']['newton_iters_per_step'],2), 'drops', d['solver']['contacts_dropped_per_step'])"; done

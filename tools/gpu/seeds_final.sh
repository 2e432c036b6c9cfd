#!/bin/bash
# GPU box: the closed-loop bench line of the product over eight seeds of the noise stream (20 timed steps, the default) and over four seeds with 100 timed steps after 10 warm-ups
# (the reference's statistic): the closed loop is chaotic, one seed's figure moves by +-9 % from build to build.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/seeds
run() { python bench.py --seed $1 --no-cpu-baseline --no-cube-only --no-with-traces --no-replay --no-steady-state "${@:2}" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.1f %.2f' % (d['ms_per_step'], d['solver']['newton_iters_per_step']))"; }
line="20 timed steps after 3 warm-ups:"; for s in 1 2 3 4 5 6 7 1234; do r=$(run $s); line="$line seed $s: $r;"; done
echo "$line" | python -c "import sys,re; l=sys.stdin.read().strip(); v=[float(x) for x in re.findall(r': ([\d.]+) ', l)]; print(l, '| mean %.1f ms (%.1f-%.1f)' % (sum(v)/len(v), min(v), max(v)))" | tee gpurun_out/seeds/seeds.txt
line="100 timed steps after 10 warm-ups:"; for s in 1 2 3 1234; do r=$(run $s --steps 100 --warmup 10); line="$line seed $s: $r;"; done
echo "$line" | python -c "import sys,re; l=sys.stdin.read().strip(); v=[float(x) for x in re.findall(r': ([\d.]+) ', l)]; print(l, '| mean %.1f ms (%.1f-%.1f)' % (sum(v)/len(v), min(v), max(v)))" | tee -a gpurun_out/seeds/seeds.txt

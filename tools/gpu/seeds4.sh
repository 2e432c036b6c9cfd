#!/bin/bash
# GPU box: the default closed-loop bench line over eight seeds of the noise stream for library variants (the closed loop is chaotic in the last bits: compare means)
# usage: tools/gpu/seeds4.sh name1 name2 ...   ("product" = judo_amd/libjudo_amd.so)
cd $GRAFT_REPO_ROOT
for v in "$@"; do
  lib=$PWD/variants/libjudo_amd_$v.so; [ "$v" == "product" ] && lib=$PWD/judo_amd/libjudo_amd.so
  line="$v:"
  for s in 1 2 3 4 5 6 7 1234; do
    ms=$(JUDO_AMD_LIB=$lib python bench.py --seed $s --no-cpu-baseline --no-cube-only --no-with-traces --no-replay --no-steady-state 2>/dev/null | tail -1 | python -c "import json,sys; print('%.1f' % json.loads(sys.stdin.read())['ms_per_step'])")
    line="$line seed $s: $ms"
  done
  echo "$line" | python -c "import sys,re; l=sys.stdin.read().strip(); v=[float(x) for x in re.findall(r': ([\d.]+)', l)]; print(l, '| mean %.1f ms (%.1f-%.1f)' % (sum(v)/len(v), min(v), max(v)))"
done

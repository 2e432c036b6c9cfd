#!/bin/bash
# GPU box: the Spot tree kernel and the closed-form kernels without the SLP vectorizer (the flag that gave the leap and fr3 kernels 6-7 %)
cd $GRAFT_REPO_ROOT
for v in product v4noslp v4noslpon product; do
  lib=$PWD/variants/libjudo_amd_$v.so; [ "$v" == "product" ] && lib=$PWD/judo_amd/libjudo_amd.so
  echo "== $v: $(JUDO_AMD_LIB=$lib python tools/diag/time_spot.py 65536 10 2>&1 | grep -E 'control steps|physics only' | cut -c1-110 | tr '\n' ' ')"
done
for v in product simplenoslp product; do
  lib=$PWD/variants/libjudo_amd_$v.so; [ "$v" == "product" ] && lib=$PWD/judo_amd/libjudo_amd.so
  for t in cartpole cylinder_push; do
    echo "== $v $t: $(JUDO_AMD_LIB=$lib python bench.py --task $t --steps 200 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4f ms, kernel %.4f' % (d['ms_per_step'], d['roofline']['kernel_ms']))")"
  done
done

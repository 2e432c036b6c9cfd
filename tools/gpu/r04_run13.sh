#!/bin/bash
cd $GRAFT_REPO_ROOT; out=gpurun_out/r4m; mkdir -p $out
REPS=1 bash tools/gpu/ab4.sh n_lo2 n_lo1 n_lo2_nopark n_lo2_noprealloc n_lo2_gat0 n_lo2_s2 n_lo2 > $out/ab.txt 2>&1; cat $out/ab.txt
for v in product v4noslp; do lib=$PWD/variants/libjudo_amd_$v.so; [ "$v" == "product" ] && lib=$PWD/judo_amd/libjudo_amd.so; echo "== spot $v: $(JUDO_AMD_LIB=$lib python tools/diag/time_spot.py 65536 10 2>&1 | grep -v amdgpu | tr '\n' ' ' | cut -c1-200)"; done
for v in product simple_noslp; do lib=$PWD/variants/libjudo_amd_$v.so; [ "$v" == "product" ] && lib=$PWD/judo_amd/libjudo_amd.so; for t in cartpole cylinder_push; do echo "== $t $v: $(JUDO_AMD_LIB=$lib python bench.py --task $t --steps 200 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['kernel_ms'])")"; done; done

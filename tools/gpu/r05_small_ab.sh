#!/bin/bash
# GPU box: small-kernel variants (compiler scheduling): cartpole / cylinder_push plan steps and kernels, the policy step at 24 and 65 536 rollouts
cd $GRAFT_REPO_ROOT
out=gpurun_out/$1; shift; mkdir -p $out
for v in "$@"; do
  lib=$PWD/variants/libjudo_amd_$v.so; [ $v == product ] && lib=$PWD/judo_amd/libjudo_amd.so
  for t in cartpole cylinder_push; do
    JUDO_AMD_LIB=$lib timeout 300 python bench.py --task $t --steps 200 --no-cpu-baseline --no-steady-state --no-replay --no-with-traces > $out/b.json 2>/dev/null
    python - "$v" "$t" "$out/b.json" <<'PY' | tee -a $out/small_ab.txt
import json,sys
d=json.loads([l for l in open(sys.argv[3]) if l.startswith("{")][-1]); r=d["per_rank"][0]
print(f"{sys.argv[1]:14s} {sys.argv[2]:14s} plan step {d['ms_per_step']*1e3:7.1f} us (median {d['plan_step_ms']['median']*1e3:.1f})  kernel {r['kernel_ms']*1e3:.1f}  update {r['exchange_ms']*1e3:.1f}  host {r['host_and_launch_ms']*1e3:.1f}")
PY
  done
  echo "$v policy: $(JUDO_AMD_LIB=$lib timeout 300 python tools/diag/time_policy_small.py 2>&1 | grep -v amdgpu.ids | tail -1)" | tee -a $out/small_ab.txt
done

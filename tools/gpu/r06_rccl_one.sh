#!/bin/bash
# GPU box: the sharded plan step with ONE rank on the nccl backend (RCCL): per_rank.exchange_ms = this rank's record + all_gather_into_tensor through RCCL + merge -- the exchange's floor with the real transport
cd $GRAFT_REPO_ROOT
out=gpurun_out/r6_rccl; rm -rf $out; mkdir -p $out
for t in cartpole leap_cube; do
  r=$([ $t == cartpole ] && echo 4096 || echo 8192); st=$([ $t == cartpole ] && echo 500 || echo 50)
  for mode in 0 1; do
    JUDO_BENCH_RCCL_ONE_RANK=$mode timeout 600 python bench.py --task $t --rollouts $r --steps $st --warmup 20 --no-cpu-baseline --no-cube-only --no-steady-state --no-replay --no-with-traces > $out/bench_${t}_rccl$mode.json 2> $out/bench_${t}_rccl$mode.log
    python - <<PY
import json
d=json.loads([l for l in open("$out/bench_${t}_rccl$mode.json") if l.startswith("{")][-1])
p=d["per_rank"][0]
print("$t rollouts $r", "RCCL one rank (launch -> all-gather -> merge)" if $mode else "one call (no collective)          ", "ms_per_step %.4f kernel %.4f exchange %.4f host %.4f" % (d["ms_per_step"], p["kernel_ms"], p["exchange_ms"], p["host_and_launch_ms"]), p.get("noise_draw_on_side_stream", ""))
PY
  done
done 2>&1 | tee $out/summary.txt
tail -n 3 $out/*.log | grep -v amdgpu.ids | tail -n 12

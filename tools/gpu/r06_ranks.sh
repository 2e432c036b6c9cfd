#!/bin/bash
# GPU box: the sharded plan step with G ranks on ONE GPU (gloo rendezvous, the record staged through the host): per_rank.exchange_ms against G, and the side-stream noise draw on the exchange's clock
cd $GRAFT_REPO_ROOT
out=gpurun_out/r6_ranks; rm -rf $out; mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_dist.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -n 3
for g in 2 4 8; do
  JUDO_BENCH_SHARED_GPU=1 timeout 600 python bench.py --gpus $g --task cartpole --rollouts $((512*g)) --steps 200 --warmup 20 --no-cpu-baseline --no-cube-only --no-steady-state --no-replay --no-with-traces > $out/bench_${g}_ranks_one_gpu_cartpole.json 2> $out/bench_${g}_ranks_one_gpu_cartpole.log
  python - <<PY
import json
d=json.loads([l for l in open("$out/bench_${g}_ranks_one_gpu_cartpole.json") if l.startswith("{")][-1])
p=d["per_rank"][0]
print("G=$g ms_per_step %.3f kernel %.3f exchange %.3f host %.3f" % (d["ms_per_step"], p["kernel_ms"], p["exchange_ms"], p["host_and_launch_ms"]), p.get("noise_draw_on_side_stream"))
PY
done 2>&1 | tee $out/summary.txt
tail -3 $out/*.log | tail -20

#!/bin/bash
# GPU box, round 6 (as round 5, the steady-state leg of the bench line left out under the profiler; small configs over 200 steps): rocprofv3 kernel stats of the DEFAULT bench command (what BENCH_rNN.json runs) + PMC passes (separate runs, as the guide prescribes,
# on a shorter run) for leap_cube (hand self-collision on = default), leap_cube with the cube's contacts only, and fr3_pick; plain bench lines of every
# config; the materialise-mode (HBM-bound) exhibit for all four BASELINE models.
# usage: tools/profile_round3.sh <tag>   -> gpurun_out/prof_<tag>/<case>_summary.txt ...; then tools/collect_profiles.py <tag> here
tag=${1:-r06}
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
out=$root/gpurun_out/prof_$tag; rm -rf $out; mkdir -p $out
run_case() {  # name, bench args
  name=$1; shift
  cmd="python $root/bench.py --no-cpu-baseline --no-cube-only --no-with-traces --no-steady-state --no-replay $*"
  timeout 900 rocprofv3 --kernel-trace --stats -d $out/${name}_stats -o $name -- $cmd > $out/${name}_bench_under_rocprof.json 2> $out/${name}_stats.log
  i=0
  for pmc in FETCH_SIZE WRITE_SIZE "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM SQ_ACTIVE_INST_MISC"; do
    i=$((i+1))
    timeout 900 rocprofv3 --kernel-trace --pmc $pmc -d $out/${name}_pmc_$i -o $name -- $cmd --steps 5 --warmup 3 > /dev/null 2> $out/${name}_pmc_$i.log
  done
  python $root/tools/rocpd_summary.py $(find $out -name "${name}_results.db" | sort) > $out/${name}_summary.txt 2>&1
  find $out -name "${name}_results.db" -delete
  tail -n 1 $out/${name}_bench_under_rocprof.json | cut -c1-260
}
run_case leap_cube --task leap_cube
run_case leap_cube_cube_only --task leap_cube --no-self-collision
run_case fr3_pick --task fr3_pick
# plain bench lines (no profiler), every config
for t in leap_cube fr3_pick; do python $root/bench.py --task $t > $out/bench_$t.json 2> $out/bench_$t.log; tail -n 1 $out/bench_$t.json | cut -c1-200; done
for t in cartpole cylinder_push; do python $root/bench.py --task $t --steps 200 > $out/bench_$t.json 2> $out/bench_$t.log; tail -n 1 $out/bench_$t.json | cut -c1-200; done
python $root/bench.py --task leap_cube --steps 100 --warmup 10 --no-cpu-baseline > $out/bench_leap_cube_100steps.json 2>> $out/bench_leap_cube.log
python $root/bench.py --task leap_cube --rollouts 8192 --no-cpu-baseline --no-cube-only > $out/bench_leap_cube_8192_one_gpu_share_of_8.json 2>> $out/bench_leap_cube.log
# materialise mode (drop-in RolloutBackend.rollout: every state and sensor written once) -- the HBM-bound exhibit, SURVEY 8(d)
for t in cartpole cylinder_push leap_cube fr3_pick; do python $root/bench.py --task $t --mode materialize --steps 5 --warmup 2 > $out/materialize_$t.json 2> $out/materialize_$t.log; tail -n 1 $out/materialize_$t.json | cut -c1-240; done
# the sharded plan step on one GPU (G ranks share cuda:0, gloo rendezvous: the record travels through the host): small kernels (cartpole, 512 rollouts per rank), so that exchange_ms is the
# exchange and not the other ranks' kernels; G = 2, 4, 8 -> per_rank.exchange_ms against G
for g in 2 4 8; do
  JUDO_BENCH_SHARED_GPU=1 timeout 600 python $root/bench.py --gpus $g --task cartpole --rollouts $((512*g)) --steps 200 --warmup 20 --no-cpu-baseline --no-cube-only --no-steady-state --no-replay --no-with-traces > $out/bench_${g}_ranks_one_gpu_cartpole.json 2> $out/bench_${g}_ranks_one_gpu_cartpole.log
  tail -n 1 $out/bench_${g}_ranks_one_gpu_cartpole.json | cut -c1-200
done
# Spot policy rollout (tools/profile_spot.sh) and the reference's benchmark statistic at the shipped rollout counts
bash $root/tools/profile_spot.sh $tag > /dev/null 2>&1
cd $root && timeout 900 python -m judo_amd.benchmark > $out/benchmark_sweep.txt 2>&1; tail -n 12 $out/benchmark_sweep.txt

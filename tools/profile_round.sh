#!/bin/bash
# GPU box: rocprofv3 kernel stats + PMC passes (separate runs, as the guide prescribes) for the headline bench line and the fr3 line.
# usage: tools/profile_round.sh <tag>   -> gpurun_out/prof_<tag>/*.db summarised into gpurun_out/prof_<tag>/*.txt
tag=${1:-r01}
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
out=$root/gpurun_out/prof_$tag; rm -rf $out; mkdir -p $out
for task in leap_cube fr3_pick; do
  cmd="python $root/bench.py --task $task --steps 3 --warmup 2 --no-cpu-baseline"
  timeout 600 rocprofv3 --kernel-trace --stats -d $out/${task}_stats -o $task -- $cmd > $out/${task}_bench_under_rocprof.json 2> $out/${task}_stats.log
  for pmc in FETCH_SIZE WRITE_SIZE "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT"; do
    name=$(echo $pmc | cut -d' ' -f1)
    timeout 600 rocprofv3 --kernel-trace --pmc $pmc -d $out/${task}_pmc_$name -o $task -- $cmd > /dev/null 2> $out/${task}_pmc_$name.log
  done
  python $root/tools/rocpd_summary.py $(find $out -name "${task}_results.db" | sort) > $out/${task}_summary.txt 2>&1
done
ls $out; for f in $out/*_bench_under_rocprof.json; do tail -n 1 $f | cut -c1-200; done

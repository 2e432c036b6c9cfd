#!/bin/bash
# GPU box: rocprofv3 kernel stats + PMC passes (separate runs) for the Spot policy rollout (65 536 rollouts x 10 control steps, tools/diag/time_spot.py).
# usage: tools/profile_spot.sh <tag>   -> gpurun_out/prof_<tag>/spot_*
tag=${1:-r01c}
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
out=$root/gpurun_out/prof_$tag; mkdir -p $out
cmd="python $root/tools/diag/time_spot.py 65536 10"
timeout 600 rocprofv3 --kernel-trace --stats -d $out/spot_stats -o spot -- $cmd > $out/spot_under_rocprof.txt 2> $out/spot_stats.log
for pmc in FETCH_SIZE WRITE_SIZE "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT"; do
  name=$(echo $pmc | cut -d' ' -f1)
  timeout 600 rocprofv3 --kernel-trace --pmc $pmc -d $out/spot_pmc_$name -o spot -- $cmd > /dev/null 2> $out/spot_pmc_$name.log
done
python $root/tools/rocpd_summary.py $(find $out -name "spot_results.db" | sort) > $out/spot_summary.txt 2>&1
ls $out; tail -2 $out/spot_under_rocprof.txt | cut -c1-300

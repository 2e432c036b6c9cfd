#!/bin/bash
# GPU box: rocprofv3 kernel stats + PMC passes (separate runs) for the headline bench line only.
# usage: tools/profile_leap.sh <tag> [JUDO_AMD_LIB|""] [extra bench.py args]   -> gpurun_out/prof_<tag>/leap_cube_summary.txt
tag=${1:-r02}
root=$GRAFT_REPO_ROOT
[ -n "$2" ] && export JUDO_AMD_LIB=$root/$2
cd /tmp && export TMPDIR=/tmp
out=$root/gpurun_out/prof_$tag; rm -rf $out; mkdir -p $out
task=leap_cube
cmd="python $root/bench.py --task $task --steps 3 --warmup 2 --no-cpu-baseline --no-cube-only $3"
timeout 600 rocprofv3 --kernel-trace --stats -d $out/${task}_stats -o $task -- $cmd > $out/${task}_bench_under_rocprof.json 2> $out/${task}_stats.log
i=0
for pmc in FETCH_SIZE WRITE_SIZE "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $pmc -d $out/${task}_pmc_$i -o $task -- $cmd > /dev/null 2> $out/${task}_pmc_$i.log
done
python $root/tools/rocpd_summary.py $(find $out -name "${task}_results.db" | sort) > $out/${task}_summary.txt 2>&1
grep -E "k_leap|kernel  " $out/${task}_summary.txt | cut -c1-60,100-220
tail -n 1 $out/${task}_bench_under_rocprof.json | cut -c1-300

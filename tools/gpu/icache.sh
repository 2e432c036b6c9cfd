#!/bin/bash
# GPU box: instruction-fetch counters of the leap / fr3 kernels (is the 190 KB kernel body bound by the 64 KB instruction cache?)
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
out=$root/gpurun_out/icache; rm -rf $out; mkdir -p $out
rocprofv3 -L 2>/dev/null | grep -o "Name:[[:space:]]*[A-Za-z0-9_]*" | sed 's/Name:[[:space:]]*//' | sort -u > $out/counters.txt
grep -i "ICACHE\|IFETCH\|INST_CACHE\|SQC_" $out/counters.txt | tr '\n' ' '; echo
for name in leap_cube fr3_pick; do
  cmd="python $root/bench.py --no-cpu-baseline --no-cube-only --no-with-traces --task $name --steps 5 --warmup 3"
  i=0
  for pmc in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES" "SQC_TC_INST_REQ SQC_TC_REQ SQC_TC_STALL SQC_ICACHE_INPUT_VALID_READY SQC_ICACHE_INPUT_VALID_READYB"; do
    i=$((i+1))
    timeout 600 rocprofv3 --kernel-trace --pmc $pmc -d $out/${name}_pmc_$i -o $name -- $cmd > /dev/null 2> $out/${name}_pmc_$i.log
  done
  python $root/tools/rocpd_summary.py $(find $out -name "${name}_results.db" | sort) > $out/${name}_summary.txt 2>&1
  find $out -name "${name}_results.db" -delete
  grep -i "k_leap\|k_fr3" $out/${name}_summary.txt | head -40
done

#!/bin/bash
# GPU box: the SQ counter passes of the leap kernel alone (short default bench), for cycles per VALU instruction per SIMD and the SQ_WAIT_ANY share.  usage: tools/gpu/r06_pmc_leap.sh <outname> [JUDO_AMD_LIB]
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
name=${1:-pmc_leap}; out=$root/gpurun_out/$name; rm -rf $out; mkdir -p $out
[ -n "$2" ] && export JUDO_AMD_LIB=$2
cmd="python $root/bench.py --no-cpu-baseline --no-cube-only --no-with-traces --no-steady-state --no-replay --task leap_cube --steps 5 --warmup 3"
i=0
for pmc in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $pmc -d $out/pmc_$i -o leap -- $cmd > /dev/null 2> $out/pmc_$i.log
done
python $root/tools/rocpd_summary.py $(find $out -name "leap_results.db" | sort) > $out/summary.txt 2>&1
find $out -name "*.db" -delete
grep -i "k_leap" $out/summary.txt | head -40

#!/bin/bash
# GPU box: wave-occupancy / wait counters of the leap kernel in library variants (build/libjudo_amd_<name>.so), cube contacts only unless SELF=1.
# usage: tools/gpu/pmc_variant.sh name1 name2 ...
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
out=$root/gpurun_out/pmcv; rm -rf $out; mkdir -p $out
sc="--no-self-collision"; [ "$SELF" == "1" ] && sc=""
for v in "$@"; do
  cmd="python $root/bench.py --no-cpu-baseline --no-cube-only --no-with-traces --task leap_cube $sc --steps 5 --warmup 3"
  i=0
  for pmc in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM" "SQ_WAVES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT"; do
    i=$((i+1))
    JUDO_AMD_LIB=$root/build/libjudo_amd_$v.so timeout 600 rocprofv3 --kernel-trace --pmc $pmc -d $out/${v}_pmc_$i -o $v -- $cmd > /dev/null 2> $out/${v}_pmc_$i.log
  done
  python $root/tools/rocpd_summary.py $(find $out -name "${v}_results.db" | sort) > $out/${v}_summary.txt 2>&1
  find $out -name "${v}_results.db" -delete
  echo "== $v"; grep "k_leap" $out/${v}_summary.txt | grep "SQ_" | awk '{print $(NF-5), $(NF-3), $NF}'
done

#!/bin/bash
# GPU box: PC sampling of the leap kernel on the replayed recorded inputs (variants/libjudo_amd_lines.so = the product with -gline-tables-only).  Output: gpurun_out/pcs/.
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
out=$root/gpurun_out/pcs; rm -rf $out; mkdir -p $out
method=${METHOD:-stochastic}; unit=${UNIT:-cycles}; interval=${INTERVAL:-1048576}
lib=$root/variants/libjudo_amd_${VARIANT:-lines}.so
JUDO_AMD_LIB=$lib PLAN_STEPS=${PLAN_STEPS:-6} timeout 600 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method $method --pc-sampling-unit $unit --pc-sampling-interval $interval --kernel-trace --output-format csv -d $out/raw -o pcs -- \
  python $root/tools/diag/ab_fixed_inputs.py replay $root/tools/diag/ab_inputs_leap.npz > $out/replay.txt 2> $out/log.txt
echo "rc=$?"; tail -3 $out/log.txt; tail -2 $out/replay.txt; find $out/raw -type f | head; du -sh $out/raw
f=$(find $out/raw -name "*pc_sampling*csv" | head -1)
if [ -n "$f" ]; then head -3 $f; wc -l $f; python $root/tools/diag/pcs_summary.py $f > $out/summary.txt 2>&1; head -60 $out/summary.txt; gzip -9 $f; fi
find $out/raw -name "*.csv" -size +20M -delete

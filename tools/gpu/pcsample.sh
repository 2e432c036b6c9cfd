#!/bin/bash
# GPU box: stochastic PC sampling of the leap kernel (where do the waves wait?).  Output under gpurun_out/pcs/.
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
out=$root/gpurun_out/pcs; rm -rf $out; mkdir -p $out
method=${METHOD:-stochastic}; unit=${UNIT:-cycles}; interval=${INTERVAL:-1048576}
timeout 300 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method $method --pc-sampling-unit $unit --pc-sampling-interval $interval --kernel-trace --output-format csv json -d $out/raw -o pcs -- \
  python $root/bench.py --no-cpu-baseline --no-cube-only --no-with-traces --task ${TASK:-leap_cube} --steps 2 --warmup 1 > $out/bench.json 2> $out/log.txt
echo "rc=$?"; tail -5 $out/log.txt; find $out/raw -type f | head; du -sh $out/raw
for f in $(find $out/raw -name "*pc_sampling*csv"); do head -5 $f; wc -l $f; done

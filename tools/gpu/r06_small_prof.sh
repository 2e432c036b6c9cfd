#!/bin/bash
# GPU box: rocprofv3 kernel stats of the closed-form models' plan step (one launch per plan step: k_plan_step + the noise draw)
root=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
out=$root/gpurun_out/prof_small; rm -rf $out; mkdir -p $out
for t in cartpole cylinder_push; do
  timeout 600 rocprofv3 --kernel-trace --stats -d $out/${t}_stats -o $t -- python $root/bench.py --task $t --steps 200 --no-cpu-baseline --no-steady-state --no-with-traces > $out/${t}_bench_under_rocprof.json 2> $out/${t}.log
  python $root/tools/rocpd_summary.py $(find $out -name "${t}_results.db" | sort) > $out/${t}_summary.txt 2>&1
  find $out -name "${t}_results.db" -delete
  head -n 14 $out/${t}_summary.txt | cut -c1-200
done

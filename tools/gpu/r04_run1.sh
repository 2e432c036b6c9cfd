#!/bin/bash
# round 4, first GPU call: baseline / one-slot solver copy on recorded inputs, census of the headline workload, leap parity suites, PC-sampling probe
cd $GRAFT_REPO_ROOT; out=gpurun_out/r4a; mkdir -p $out
python -c "import torch; print(torch.cuda.get_device_name(0))" > $out/dev.txt 2>&1
REPS=2 bash tools/gpu/ab4.sh base product ns1p9 ns1p99 base > $out/ab.txt 2>&1; cat $out/ab.txt
JUDO_AMD_LIB=$PWD/variants/libjudo_amd_census.so python tools/diag/census_v5.py > $out/census.txt 2>&1; tail -12 $out/census.txt
timeout 900 python -m pytest tests/test_gpu_leap.py tests/test_gpu_leap_self.py tests/test_gpu_edges.py -x -q -m gpu > $out/pytest_leap.txt 2>&1; tail -5 $out/pytest_leap.txt
# PC sampling probe
cd /tmp; export TMPDIR=/tmp
rocprofv3 -L > $GRAFT_REPO_ROOT/$out/avail.txt 2>&1; grep -n -i -B2 -A12 "pc.sampl" $GRAFT_REPO_ROOT/$out/avail.txt | head -60
for cfg in "stochastic cycles 1048576" "stochastic cycles 65536" "host_trap time 100"; do
  set -- $cfg; d=$GRAFT_REPO_ROOT/$out/pcs_$1_$3; rm -rf $d; mkdir -p $d
  JUDO_AMD_LIB=$GRAFT_REPO_ROOT/variants/libjudo_amd_dbg.so timeout 240 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method $1 --pc-sampling-unit $2 --pc-sampling-interval $3 --kernel-trace --output-format csv -d $d/raw -o pcs -- \
    python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-cube-only --no-replay --no-with-traces --task leap_cube --steps 2 --warmup 1 > $d/bench.json 2> $d/log.txt
  echo "pcs $cfg rc=$?"; tail -3 $d/log.txt
  f=$(find $d/raw -name "*pc_sampling*csv" | head -1)
  if [ -n "$f" ]; then wc -l $f; head -3 $f; python $GRAFT_REPO_ROOT/tools/diag/pcs_summary.py $f > $d/summary.txt 2>&1; head -50 $d/summary.txt; rm -rf $d/raw; break; fi
  rm -rf $d/raw
done

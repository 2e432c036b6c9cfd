#!/bin/bash
# GPU box: the round-end sequence's tests + the fr3 part of tools/profile_round4.sh (after a change to the fr3 kernel only); merges into gpurun_out/prof_r04 and gpurun_out/final
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/final gpurun_out/prof_r04
timeout 1200 python -m pytest tests/ -m gpu -x -q 2>&1 | tail -4 | tee gpurun_out/final/pytest.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 | tee gpurun_out/final/smoke.txt
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/prof_r04
cd /tmp && export TMPDIR=/tmp
name=fr3_pick
cmd="python $root/bench.py --no-cpu-baseline --no-cube-only --no-with-traces --no-steady-state --task fr3_pick"
rm -rf $out/${name}_stats $out/${name}_pmc_*
timeout 600 rocprofv3 --kernel-trace --stats -d $out/${name}_stats -o $name -- $cmd > $out/${name}_bench_under_rocprof.json 2> $out/${name}_stats.log
i=0
for pmc in FETCH_SIZE WRITE_SIZE "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $pmc -d $out/${name}_pmc_$i -o $name -- $cmd --steps 5 --warmup 3 > /dev/null 2> $out/${name}_pmc_$i.log
done
python $root/tools/rocpd_summary.py $(find $out -name "${name}_results.db" | sort) > $out/${name}_summary.txt 2>&1
find $out -name "${name}_results.db" -delete
python $root/bench.py --task fr3_pick > $out/bench_fr3_pick.json 2> $out/bench_fr3_pick.log; tail -n 1 $out/bench_fr3_pick.json | cut -c1-200
python $root/bench.py --task fr3_pick --mode materialize --steps 5 --warmup 2 > $out/materialize_fr3_pick.json 2> $out/materialize_fr3_pick.log

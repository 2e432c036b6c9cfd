#!/bin/bash
# GPU box, round 4: the driver's round-end sequence (GPU suite, smoke, default bench line), then the round's profiles
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/final
timeout 1800 python -m pytest tests/ -m gpu -x -q 2>&1 | tail -4 | tee gpurun_out/final/pytest.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 | tee gpurun_out/final/smoke.txt
( time python bench.py ) > gpurun_out/final/bench.json 2> gpurun_out/final/bench.err; tail -1 gpurun_out/final/bench.json | cut -c1-600; tail -4 gpurun_out/final/bench.err
bash tools/profile_round4.sh r04 2>&1 | tail -30
bash tools/profile_spot.sh r04 2>&1 | tail -5
cd $GRAFT_REPO_ROOT
python -m judo_amd.benchmark > gpurun_out/prof_r04/benchmark_sweep.txt 2>&1; tail -25 gpurun_out/prof_r04/benchmark_sweep.txt

#!/bin/bash
# GPU box: what the driver runs at round end -- the GPU suite, smoke(), the default bench line
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/final
timeout 1500 python -m pytest tests/ -m gpu -x -q 2>&1 | tail -4 | tee gpurun_out/final/pytest.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 | tee gpurun_out/final/smoke.txt
( time python bench.py ) > gpurun_out/final/bench.json 2> gpurun_out/final/bench.err; tail -1 gpurun_out/final/bench.json | cut -c1-400; tail -4 gpurun_out/final/bench.err

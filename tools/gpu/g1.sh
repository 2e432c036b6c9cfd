cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
timeout 1500 python -m pytest tests/ -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r3/t1.log
cat gpurun_out/r3/t1.log
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-cube-only 2>&1 | tail -1 | cut -c1-2500
python bench.py --task fr3_pick --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-600

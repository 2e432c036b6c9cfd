#!/bin/bash
# GPU box, final build, part B: the plain bench lines again, now that profiles/r05_traffic.json comes from THIS build's PMC passes (the line's roofline.traffic is read from it),
# then what the driver runs at round end: pytest -m gpu, smoke, the default bench command
cd $GRAFT_REPO_ROOT
out=gpurun_out/prof_r05; mkdir -p $out
for t in leap_cube fr3_pick; do python bench.py --task $t > $out/bench_$t.json 2> $out/bench_$t.log; tail -n 1 $out/bench_$t.json | cut -c1-200; done
python bench.py --task leap_cube --steps 100 --warmup 10 --no-cpu-baseline > $out/bench_leap_cube_100steps.json 2>> $out/bench_leap_cube.log
python bench.py --task leap_cube --rollouts 8192 --no-cpu-baseline --no-cube-only > $out/bench_leap_cube_8192_one_gpu_share_of_8.json 2>> $out/bench_leap_cube.log
z=gpurun_out/r5z; rm -rf $z; mkdir -p $z
timeout 600 python -m pytest tests -x -q -m gpu -p no:cacheprovider > $z/pytest_plain.txt 2>&1; echo "plain pytest rc $?" >> $z/pytest_plain.txt; tail -n 3 $z/pytest_plain.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('__SMOKE_OK__')" 2>&1 | grep -v amdgpu.ids | tail -n 3
( time timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $z/bench_driver.json 2> $z/bench_driver.log ) 2>&1 | grep real
tail -n 1 $z/bench_driver.json | cut -c1-300

#!/bin/bash
cd $GRAFT_REPO_ROOT; out=gpurun_out/r4h; mkdir -p $out
TASK=fr3_pick REPS=2 bash tools/gpu/ab4.sh v6old product v6old > $out/ab_fr3.txt 2>&1; cat $out/ab_fr3.txt
timeout 900 python -m pytest tests/test_gpu_fr3.py tests/test_gpu_controller.py tests/test_gpu_dist.py -x -q -m gpu > $out/pytest.txt 2>&1; tail -3 $out/pytest.txt
python bench.py --task fr3_pick --no-cpu-baseline > $out/bench_fr3.json 2>/dev/null; python -c "
import json; d=json.loads(open('$out/bench_fr3.json').read().strip().splitlines()[-1]); print('fr3 bench', d['ms_per_step'], d['roofline']['kernel_ms'], d['steady_state'], d['recorded_inputs']['kernel_ms'], d['solver'])"

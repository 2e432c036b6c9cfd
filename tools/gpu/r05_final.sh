#!/bin/bash
# GPU box: what the driver runs at round end (pytest -m gpu, smoke, the default bench command), plus the margins of the final build
cd $GRAFT_REPO_ROOT
out=gpurun_out/r5z; rm -rf $out; mkdir -p $out
rm -f gpurun_out/test_margins.jsonl
JUDO_RECORD_MARGINS=1 timeout 900 python -m pytest tests -m gpu -q -W error::RuntimeWarning -p no:cacheprovider > $out/pytest.txt 2>&1; echo "pytest rc $?" >> $out/pytest.txt; tail -n 6 $out/pytest.txt
cp gpurun_out/test_margins.jsonl $out/
timeout 600 python -m pytest tests -x -q -m gpu -p no:cacheprovider > $out/pytest_plain.txt 2>&1; echo "plain pytest rc $?" >> $out/pytest_plain.txt; tail -n 3 $out/pytest_plain.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('__SMOKE_OK__')" 2>&1 | grep -v amdgpu.ids | tail -n 3
( time timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_driver.json 2> $out/bench_driver.log ) 2>&1 | grep real
python - <<PY
import json
d=json.loads([l for l in open("$out/bench_driver.json") if l.startswith("{")][-1])
print({k:d[k] for k in ("metric","value","ms_per_step","n_gpus","steps","warmup","dtype","scaling")}); print(d["roofline"]); print(d["cpu_baseline"]); print(d["benchmark_100"])
PY

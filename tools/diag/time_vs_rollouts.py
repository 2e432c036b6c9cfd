"""Plan-step time against the number of rollouts at the shipped horizons: flat while every rollout's wave finds a free SIMD."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from judo_amd.benchmark import plan_times
for task in ("leap_cube", "fr3_pick", "spot_navigate"):
    row = {}
    for n in (32, 256, 1024, 2048, 4096, 8192, 16384):
        row[n] = round(float(__import__("numpy").median(plan_times(task, "mppi", 12, 3, n))) * 1e3, 2)
    print(task, json.dumps(row), flush=True)

"""leap_cube plan-step time at small rollout counts (shipped horizon, seeded, 30 plan steps from the same start), for A/B of library variants (JUDO_AMD_LIB)."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from judo_amd.controller import make_controller
row = {}
for n in (32, 1024, 2048, 4096, 8192):
    c = make_controller("leap_cube", "mppi"); c.optimizer.config.num_rollouts = n
    c.reset(); c.current_state = c.task.default_state(); c.system_metadata = {"goal_quat": np.array([0.0, 1.0, 0.0, 0.0])}; c.optimizer.seed(5)
    ts = []; t = 0.0
    for i in range(33):
        torch.cuda.synchronize(); t0 = time.perf_counter(); c.time = t; c.update_action(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3); t += 0.05
    row[n] = (round(float(np.median(ts[3:])), 2), round(float(np.mean(ts[3:])), 2))
print(os.environ.get("JUDO_AMD_LIB", "default"), json.dumps(row))

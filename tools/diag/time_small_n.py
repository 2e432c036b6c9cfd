"""Plan-step time of every cooperative kernel at small rollout counts (the reference's shipped configurations and small shards), shipped horizons, seeded:
leap_cube MPPI, fr3_pick CEM, spot_navigate MPPI.  Run twice, with JUDO_AMD_LATENCY_SHIFT=0 and without, for the effect of the latency mode."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from judo_amd.controller import make_controller
out = {}
for task, opt, ns in (("leap_cube", "mppi", (32, 256, 1024, 2048, 4096)), ("fr3_pick", "cem", (32, 256, 1024, 2048, 4096)), ("spot_navigate", "mppi", (24, 256, 1024))):
    row = {}
    for n in ns:
        c = make_controller(task, opt); c.optimizer.config.num_rollouts = n
        c.reset(); c.current_state = c.task.default_state()
        if task == "leap_cube": c.system_metadata = {"goal_quat": np.array([0.0, 1.0, 0.0, 0.0])}
        c.optimizer.seed(5); ts = []; t = 0.0
        for i in range(23):
            torch.cuda.synchronize(); t0 = time.perf_counter(); c.time = t; c.update_action(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3); t += 0.05
        row[n] = round(float(np.median(ts[3:])), 2)
    out[task] = row
print("JUDO_AMD_LATENCY_SHIFT=" + os.environ.get("JUDO_AMD_LATENCY_SHIFT", "auto"), json.dumps(out))

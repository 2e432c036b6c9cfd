"""Where a plan step's time goes at one GPU's share of an 8-GPU run (8 192 rollouts of the headline workload): kernel events vs wall clock."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from judo_amd.controller import make_controller

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
c = make_controller("leap_cube", "mppi")
c.optimizer.config.num_rollouts = N
c.controller_cfg.horizon = 64 * c.task.dt
c.reset(); c.current_state = c.task.default_state(); c.system_metadata = {"goal_quat": np.array([0.0, 1.0, 0.0, 0.0])}
c.optimizer.seed(1)
t = 0.0
for _ in range(3):
    c.time = t; c.update_action(); t += 0.05
c.record_kernel_events = True; c.kernel_events.clear()
torch.cuda.synchronize()
w = []
for _ in range(20):
    t0 = time.perf_counter(); c.time = t; c.update_action(); t += 0.05; w.append(time.perf_counter() - t0)
torch.cuda.synchronize()
k = [a.elapsed_time(b) for a, b in c.kernel_events]
print(f"N={N}: plan step {np.mean(w)*1e3:.2f} ms (min {np.min(w)*1e3:.2f}), rollout kernel {np.mean(k):.2f} ms, everything else {np.mean(w)*1e3 - np.mean(k):.2f} ms")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(10):
    c.time = t; c.update_action(); t += 0.05
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)

"""Where the host time of a small plan step goes: cProfile over cartpole plan steps (kernel 0.06 ms) + a wall-clock split."""
import cProfile, pstats, sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from judo_amd.controller import make_controller

task = sys.argv[1] if len(sys.argv) > 1 else "cartpole"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
ctrl = make_controller(task, "mppi")
ctrl.optimizer.config.num_rollouts = N
ctrl.controller_cfg.horizon = 64 * ctrl.task.dt
ctrl.reset(); ctrl.current_state = ctrl.task.default_state()
if task == "leap_cube": ctrl.system_metadata = {"goal_quat": np.array([0.0, 1.0, 0.0, 0.0])}
t = 0.0
for _ in range(20):
    ctrl.time = t; ctrl.update_action(); t += 0.05
torch.cuda.synchronize()
ts = []
for _ in range(200):
    ctrl.time = t; t0 = time.perf_counter(); ctrl.update_action(); ts.append(time.perf_counter() - t0); t += 0.05
print(f"{task} N={N}: plan step median {np.median(ts)*1e3:.4f} ms, min {np.min(ts)*1e3:.4f} ms")
pr = cProfile.Profile(); pr.enable()
for _ in range(200):
    ctrl.time = t; ctrl.update_action(); t += 0.05
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)

import sys, time, numpy as np, torch
sys.path.insert(0, ".")
from judo_amd.controller import make_controller
for task, opt, N, H in (("cylinder_push", "mppi", 16384, 64), ("fr3_pick", "cem", 32768, 40)):
    c = make_controller(task, opt); c.optimizer.config.num_rollouts = N; c.controller_cfg.horizon = H * c.task.dt
    c.reset(); c.current_state = c.task.default_state(); c.optimizer.seed(1234); c.record_kernel_events = True
    t = 0.0; ts = []
    for i in range(30):
        t0 = time.perf_counter(); c.time = t; c.update_action(); t += 1.0 / c.controller_cfg.control_freq; ts.append((time.perf_counter() - t0) * 1e3)
    print(task, " ".join(f"{x:.2f}" for x in ts))

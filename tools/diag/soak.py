"""Soak: many closed-loop plan steps (every nominal finite, traces read, counters printed per block) at the headline size and at the shipped configurations."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from judo_amd.controller import make_controller
for task, opt, n, blocks in (("leap_cube", "mppi", 65536, 6), ("fr3_pick", "cem", 32768, 6), ("fr3_pick", "cem", None, 6), ("leap_cube_down", "mppi", None, 6), ("caltech_leap_cube", "ps", None, 6)):
    c = make_controller(task, opt); c.solver_warnings = False
    if n: c.optimizer.config.num_rollouts = n; c.controller_cfg.horizon = (64 if task == "leap_cube" else 40) * c.task.dt
    c.reset(); c.current_state = c.task.default_state(); c.system_metadata = c.task.get_sim_metadata(); c.optimizer.seed(11)
    t = 0.0; c.solver_stats()
    for b in range(blocks):
        t0 = time.perf_counter()
        for i in range(100):
            c.time = t; c.update_action(); tr = c.traces; t += 0.05
            assert np.isfinite(c.nominal_knots).all() and (tr is None or np.isfinite(tr).all()), (task, b, i)
        torch.cuda.synchronize(); st = c.solver_stats()
        print(f"{task} {opt} N={c.optimizer.num_rollouts} H={c.num_timesteps} steps {100 * b}-{100 * b + 99}: {(time.perf_counter() - t0) * 10:.2f} ms/step, Newton {st['newton_iters'] / st['steps']:.2f} it/step, cap hits {st['newton_cap_hits'] / st['steps']:.1e}, dropped {st['contact_overflow'] / st['steps']:.1e}", flush=True)

"""Solver counters at the SHIPPED configurations (the rollout counts and horizons a judo user gets by default): 100 closed-loop plan steps per task x optimizer."""
import sys, warnings
import numpy as np, torch
sys.path.insert(0, ".")
from judo_amd.controller import make_controller
for task in (sys.argv[1:] or ["leap_cube", "leap_cube_down", "caltech_leap_cube", "fr3_pick"]):
    for opt in ("mppi", "cem", "ps"):
        c = make_controller(task, opt); c.solver_warnings = False
        c.reset(); c.current_state = c.task.default_state(); c.system_metadata = c.task.get_sim_metadata(); c.optimizer.seed(3)
        t = 0.0; c.solver_stats()
        for i in range(100):
            c.time = t; c.update_action(); _ = c.traces; t += 0.05
        st = c.solver_stats()
        print(f"{task:18s} {opt:5s} N={c.optimizer.num_rollouts:3d} H={c.num_timesteps:3d}: Newton {st['newton_iters'] / st['steps']:.2f} it/step, cap hits {st['newton_cap_hits'] / st['steps']:.1e}/step, contacts dropped {st['contact_overflow'] / st['steps']:.1e}/step, finite {bool(np.isfinite(c.nominal_knots).all())}")

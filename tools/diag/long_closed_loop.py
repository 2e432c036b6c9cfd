"""Robustness run: 300 consecutive plan steps of the headline workload (fixed state, evolving plan), every nominal finite, solver counters per 100 steps."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from judo_amd.controller import make_controller
task = sys.argv[1] if len(sys.argv) > 1 else "leap_cube"
c = make_controller(task, "mppi"); c.optimizer.config.num_rollouts = 65536; c.controller_cfg.horizon = 0.64; c.solver_warnings = False
c.reset(); c.current_state = c.task.default_state(); c.system_metadata = {"goal_quat": np.array([0.0, 1.0, 0.0, 0.0])}; c.optimizer.seed(7)
t = 0.0; c.solver_stats()
for blk in range(3):
    t0 = time.perf_counter()
    for i in range(100):
        c.time = t; c.update_action(); t += 0.05
        assert np.isfinite(c.nominal_knots).all(), (blk, i)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 100
    st = c.solver_stats()
    print(f"{task} steps {100 * blk}-{100 * blk + 99}: {dt * 1e3:.1f} ms/step, Newton {st['newton_iters'] / st['steps']:.2f} it/step (wave {st['wave_newton_iters'] / max(st['wave_steps'], 1):.2f}), cap hits {st['newton_cap_hits'] / st['steps']:.2e}/step, "
          f"contacts dropped {st['contact_overflow'] / st['steps']:.2e}/step, |nominal| max {np.abs(c.nominal_knots).max():.2f}")

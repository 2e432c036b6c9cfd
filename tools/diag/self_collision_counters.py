"""Counters of the hand self-collision path of jh_engine_v5.hip (a -DJH_V5_COUNT build): replay of the recorded headline plan steps."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, ".")
from judo_amd.controller import make_controller
from judo_amd import _lib
d = np.load("tools/diag/ab_inputs_leap.npz")
c = make_controller("leap_cube", "mppi"); c.optimizer.config.num_rollouts = 65536; c.controller_cfg.horizon = 0.64
c.reset(); c.current_state = c.task.default_state(); c.system_metadata = {"goal_quat": np.array([0.0, 1.0, 0.0, 0.0])}
L = C.CDLL(_lib.LIB_PATH)
for i in (0, 10, 20, 39):
    c.model.stats()
    c.optimizer.seed(1000 + i); c.nominal_knots = d["knots"][i].copy(); c.times = d["times"][i].copy(); c.update_spline(c.times, c.nominal_knots); c.time = float(d["t"][i])
    c.update_action(); torch.cuda.synchronize()
    out = (C.c_int * 40)(); L.jh_model_hist(c.model.handle, out)
    dense, its, l2, bp, ws, hh = out[0], out[1], out[2], out[3], out[4], out[5]
    print(f"plan step {i}: wave-iterations {its}, dense {dense} ({dense / max(its, 1):.3%}); per rollout-step: body pairs hit {bp / (65536 * 64):.2f}, hand geom pairs hit {hh / (65536 * 64):.3f}; level-2 passes per wave-step {l2 / max(ws, 1):.1f}; "
          f"coupling classes per rollout-step [none, pairs, a chain with two neighbours, cycle] {[round(out[6 + k] / (65536 * 64), 4) for k in range(4)]}")

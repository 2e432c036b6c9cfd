"""Broad-phase counters of the leap kernel (a -DJH_V5_COUNT build selected with JUDO_AMD_LIB) on recorded plan steps of the headline workload: surviving hand body pairs per
rollout-step, level-2 passes per wave-step (the maximum over the wave's four rollouts), hand-hand candidate geom pairs per rollout-step."""
import ctypes as C, sys
import numpy as np, torch
sys.path.insert(0, ".")
from judo_amd.controller import make_controller
from judo_amd import _lib
d = np.load("tools/diag/ab_inputs_leap.npz")
L = _lib.lib(); L.jh_model_counters.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.c_int, C.c_int]
c = make_controller("leap_cube", "mppi"); c.optimizer.config.num_rollouts = 65536; c.controller_cfg.horizon = 0.64
c.reset(); c.current_state = c.task.default_state(); c.system_metadata = {"goal_quat": np.array([0.0, 1.0, 0.0, 0.0])}
for i in (2, 20, 35):
    c.model.stats()
    c.optimizer.seed(1000 + i); c.nominal_knots = d["knots"][i].copy(); c.times = d["times"][i].copy(); c.update_spline(c.times, c.nominal_knots); c.time = float(d["t"][i])
    c.update_action(); torch.cuda.synchronize()
    raw = (C.c_int * 10)(); assert L.jh_model_counters(c.model.handle, raw, 24, 10) == 0
    dense, its, l2, bp, hsteps, hh = raw[0], raw[1], raw[2], raw[3], raw[4], raw[5]
    nw = 65536 // 4 * 64; nr = 65536 * 64
    print(f"plan step {i:2d}: surviving hand body pairs {bp / nr:.2f} per rollout-step; level-2 passes {l2 / nw:.2f} per wave-step; hand-hand candidate geom pairs {hh / nr:.2f} per rollout-step; "
          f"wave iterations {its / nw:.2f}, with a dense row {dense / max(its, 1):.3f}; slot classes {[raw[6 + k] for k in range(4)]}")

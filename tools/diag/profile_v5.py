"""Shader-clock split of jh_engine_v5.hip per phase (a -DJH_V5_TICKS build), on recorded plan steps of the headline workload, with and without the hand's own contacts."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, ".")
from judo_amd.controller import make_controller
from judo_amd import _lib
d = np.load("tools/diag/ab_inputs_leap.npz")
L = C.CDLL(_lib.LIB_PATH); L.jh_model_profile.argtypes = [C.c_void_p, C.POINTER(C.c_longlong)]
names = ["kinematics+dynamics", "broad phase", "narrow phase", "rows+warm start", "gradient", "Newton matrix", "factorisation+direction", "line search+step, integration"]
for self_on in (False, True):
    c = make_controller("leap_cube", "mppi"); c.optimizer.config.num_rollouts = 65536; c.controller_cfg.horizon = 0.64
    c.reset(); c.current_state = c.task.default_state(); c.system_metadata = {"goal_quat": np.array([0.0, 1.0, 0.0, 0.0])}
    c.model.set_self_collision(self_on)
    for i in (2, 35):
        c.model.stats()
        c.optimizer.seed(1000 + i); c.nominal_knots = d["knots"][i].copy(); c.times = d["times"][i].copy(); c.update_spline(c.times, c.nominal_knots); c.time = float(d["t"][i])
        c.update_action(); torch.cuda.synchronize()
        out = (C.c_longlong * 10)(); L.jh_model_profile(c.model.handle, out)
        tot = sum(out); nw = 65536 // 4 * 64
        print(f"self-collision {'on ' if self_on else 'off'} plan step {i:2d}: {tot / nw / 1e3:7.1f} kcycles per wave-step | " + " | ".join(f"{n} {100 * v / tot:.0f}% ({v / nw / 1e3:.1f}k)" for n, v in zip(names, out)))

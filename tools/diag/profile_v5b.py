"""Shader-clock split of the leap kernel per phase, fine (a -DJH_V5_TICKS build selected with JUDO_AMD_LIB), on recorded plan steps of the headline workload."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, ".")
from judo_amd.controller import make_controller
from judo_amd import _lib
d = np.load("tools/diag/ab_inputs_leap.npz")
L = _lib.lib(); L.jh_model_counters.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.c_int, C.c_int]
names = ["kinematics+dynamics", "broad: geom level of the pairs", "narrow phase", "rows+warm start", "gradient", "Hessian assembly", "chain blocks", "integration+cost", "conv. test+Hessian init",
         "Schur+6x6+back-subst", "LS set-up (Mp, Jp)", "LS slope evaluations", "step", "broad: cube vs geoms", "broad: hand body pairs"]
c = make_controller("leap_cube", "mppi"); c.optimizer.config.num_rollouts = 65536; c.controller_cfg.horizon = 0.64
c.reset(); c.current_state = c.task.default_state(); c.system_metadata = {"goal_quat": np.array([0.0, 1.0, 0.0, 0.0])}
for i in (2, 20, 35):
    c.model.stats()
    c.optimizer.seed(1000 + i); c.nominal_knots = d["knots"][i].copy(); c.times = d["times"][i].copy(); c.update_spline(c.times, c.nominal_knots); c.time = float(d["t"][i])
    c.update_action(); torch.cuda.synchronize()
    raw = (C.c_int * 32)(); assert L.jh_model_counters(c.model.handle, raw, 384, 32) == 0
    out = np.frombuffer(bytes(raw), dtype=np.int64)[:15].astype(float)
    tot = out.sum(); nw = 65536 // 4 * 64
    print(f"plan step {i:2d}: {tot / nw / 1e3:7.2f} kticks per wave-step")
    for n, v in sorted(zip(names, out), key=lambda kv: -kv[1]): print(f"    {n:28s} {100 * v / tot:5.1f} %  ({v / nw / 1e3:.2f}k)")

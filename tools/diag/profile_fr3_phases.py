"""Per-phase shader-cycle split of the fr3 kernel (needs a -DJH_V6_PHASES -DJH_V6_EXITSTATS build of jh_engine_v6.hip, JUDO_AMD_LIB=build/libjudo_amd_v6ph.so)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from judo_amd import _lib
from judo_amd.controller import make_controller
N = 32768
c = make_controller("fr3_pick", "cem"); c.optimizer.config.num_rollouts = N; c.controller_cfg.horizon = 40 * c.task.dt
c.reset(); c.current_state = c.task.default_state()
t = 0.0
for _ in range(3):
    c.time = t; c.update_action(); t += 0.05
torch.cuda.synchronize(); c.model.stats()
c.time = t; c.update_action(); torch.cuda.synchronize()
L = _lib.lib(); L.jh_model_profile.argtypes = [C.c_void_p, C.POINTER(C.c_longlong)]
out = (C.c_longlong * 10)(); L.jh_model_profile(c.model.handle, out)
names = ["integrate+cost(prev) + controls + kinematics", "sensors (incl. geom distances)", "arm dynamics + a0", "collision", "constraint rows", "newton: gradient pass, convergence test, step", "tail", "newton: Hessian assembly", "newton: row Cholesky + backward solve", "newton: line search"]
tot = sum(out[:10])
for n, v in zip(names, out):
    print(f"  {n:46s} {v / (N // 4) / 40 / 1e3:8.1f} kcyc/step/wave {100 * v / tot:5.1f}%")
print(c.model.stats())

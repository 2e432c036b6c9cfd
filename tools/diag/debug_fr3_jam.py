"""Scratch diagnostic (GPU box): single steps from jammed-gripper states (up to 92 pad-against-pad contacts), solver exit statistics (needs a -DJH_V3_EXITSTATS build
for the exit counters) and the error against the oracle by contact count."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ctypes as C
import numpy as np
from judo_amd import _lib
from judo_amd.rollout_backend import GpuRolloutBackend
from judo_amd.tasks import FR3Pick
from oracle import oracle as O
from judo_amd import engine_model as EM
if os.environ.get("TOL"): EM.SOLVER_TOL = float(os.environ["TOL"])
if os.environ.get("LSTOL"): EM.SOLVER_LS_TOL = float(os.environ["LSTOL"])
om, task = O.Model("fr3_pick"), FR3Pick()
rng = np.random.default_rng(0)
N = 192
x0 = np.tile(task.default_state(), (N, 1))
x0[:, 14:16] = rng.uniform(-0.003, 0.0005, (N, 2)); x0[:, 16 + 13 : 16 + 15] = rng.uniform(-0.3, 0.1, (N, 2))
U = np.tile(task.reset_command, (N, 1, 1))
ncon = np.array([om.forward(x[:16], x[16:], task.reset_command)["ncon"] for x in x0])
rs, _ = om.rollout(x0, U)
be = GpuRolloutBackend("fr3_pick", N)
gs, _, _ = be.rollout(x0, U)
L = _lib.lib(); L.jh_model_hist.argtypes = [C.c_void_p, C.POINTER(C.c_int)]; hh = (C.c_int * 40)(); L.jh_model_hist(be.model.handle, hh)
print("stats", be.model.stats(), "exits grad/notdescent/decrease/cap/noise-floor", list(hh)[:5])
dv = np.abs(rs[:, 0, 29:31] - x0[:, 29:31]).max(axis=1) + 1e-3
ev = np.abs(gs[:, 0, 29:31] - rs[:, 0, 29:31]).max(axis=1) / dv
for lo, hi in ((0, 1), (1, 33), (33, 49), (49, 80), (80, 200)):
    m = (ncon >= lo) & (ncon < hi)
    if m.any(): print(f"ncon {lo}..{hi - 1}: {m.sum()} states, rel finger-velocity error median {np.median(ev[m]):.2e} max {ev[m].max():.2e}; position err max {np.abs(gs[m, 0, :16] - rs[m, 0, :16]).max():.2e}")
if os.environ.get("DUMP"):
    idx = np.argsort(-ncon)[:40]
    for i in idx:
        print(f"n{i:3d} ncon {ncon[i]:3d} q {x0[i,14]:+.5f} {x0[i,15]:+.5f} v {x0[i,29]:+.3f} {x0[i,30]:+.3f} | oracle acc {(rs[i,0,29]-x0[i,29])/0.004:+9.3f} {(rs[i,0,30]-x0[i,30])/0.004:+9.3f} kernel {(gs[i,0,29]-x0[i,29])/0.004:+9.3f} {(gs[i,0,30]-x0[i,30])/0.004:+9.3f}")

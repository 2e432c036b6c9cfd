"""Scratch diagnostic (GPU box): fr3_pick cooperative kernel vs oracle, per-step / per-column error report."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from tests.test_gpu_fr3 import _controls
from judo_amd.rollout_backend import GpuRolloutBackend
from tests import xcheck; xcheck.load()  # kernel generations 1 / 2 live in the test build
from judo_amd import engine_model
if os.environ.get('TOL'): engine_model.SOLVER_TOL = float(os.environ['TOL'])
if os.environ.get('LSTOL'): engine_model.SOLVER_LS_TOL = float(os.environ['LSTOL'])
N, H = 128, 40
import ctypes as C
from judo_amd import _lib
def hist(be):
    L = _lib.lib(); L.jh_model_hist.argtypes = [C.c_void_p, C.POINTER(C.c_int)]; hh = (C.c_int * 40)(); L.jh_model_hist(be.model.handle, hh); return 'exits grad/notdescent/decrease/cap', list(hh)[:4]
om, task, knots, U = _controls(N, H, seed=1)
for kind in ("home",):
    x0 = task.default_state()
    if kind == "grasping":
        x0 = x0.copy(); x0[7:14] = [0.0, 0.55, 0.0, -2.05, 0.0, 2.6, 0.785]; x0[14:16] = [0.03, 0.025]
    rs, rsens = om.rollout(x0, U)
    for gen in ((2, 1) if os.environ.get('BOTH') else (2,)):
        be = GpuRolloutBackend("fr3_pick", N); be.model.set_kernel(gen)
        gs, gsens, _ = be.rollout(x0, U)
        e = np.abs(gs - rs)
        print(kind, 'gen', gen, 'step0 max err per column:', np.round(e[:, 0].max(0), 5).tolist())
        print('   worst rollout/col step0', np.unravel_index(e[:, 0].argmax(), e[:, 0].shape), ' err by step (max over all):', np.round(e.max((0, 2))[[0, 1, 2, 5, 10, 20, 39]], 4).tolist())
        print('   sensors err by step:', np.round(np.abs(gsens - rsens).max((0, 2))[[0, 1, 2, 5, 10, 20, 39]], 4).tolist(), be.model.stats(reset=False), hist(be))

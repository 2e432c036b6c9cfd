"""Scratch diagnostic (CPU): Newton iterations per solve of the fp64 oracle on bench-like leap_cube rollouts, per tolerance."""
import ctypes as C, os, sys
os.environ["JUDO_ORACLE_EXPERIMENTS"] = "1"  # oracle/libjudo_oracle_exp.so: the solver experiments are not in the parity oracle
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import oracle as O
from judo_amd.tasks import LeapCube
from judo_amd import spline
task = LeapCube(); om = O.Model('leap_cube'); L = O.lib()
L.jo_set_solver.argtypes = [C.c_void_p, C.c_double, C.c_int]; L.jo_solver_histogram.argtypes = [C.c_void_p, C.c_int]
N, H, K = 64, 64, 4
rng = np.random.default_rng(0)
nom = np.tile(task.reset_command if hasattr(task, 'reset_command') else task.optimizer_warm_start(), (K, 1))
sig = 0.2 * np.arange(1, K + 1)[:, None]
knots = nom[None] + rng.standard_normal((N, K, 16)) * sig[None]
knots[0] = nom
lo, hi = task.actuator_ctrlrange[:, 0], task.actuator_ctrlrange[:, 1]
knots = np.clip(knots, lo, hi)
W = O.spline_weights('cubic', np.linspace(0, 0.64, K), np.arange(H) * 0.01)
ctrl = np.einsum('hk,nku->nhu', W, knots)
x0 = task.default_state() if hasattr(task, 'default_state') else None
for mode, tol in [(0, 1e-8), (1, 1e-8), (0, 1e-5), (1, 1e-5)]:
    L.jo_set_warmstart_mode(mode); L.jo_set_solver(om.ptr, tol, 100)
    h = (C.c_long * 32)(); L.jo_solver_histogram(h, 1)
    st, se = om.rollout(np.asarray(x0, float), ctrl)
    L.jo_solver_histogram(h, 1); h = np.array(list(h), float)
    print(f'warm-start mode {mode} tol {tol:g}: mean {np.sum(h * np.arange(32)) / h.sum():.2f}  hist%', ' '.join(f'{i}:{100 * v / h.sum():.0f}' for i, v in enumerate(h) if v), ' final cube z', st[:3, -1, 2])

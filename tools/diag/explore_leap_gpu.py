"""Scratch diagnostic (GPU box): leap_cube engine vs oracle, per-step and per-rollout errors."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from judo_amd import _lib
if os.environ.get("JH_LIB"): _lib.LIB_PATH = os.environ["JH_LIB"]
from oracle import oracle as O
from judo_amd.rollout_backend import GpuRolloutBackend
from judo_amd.tasks import LEAP_QPOS_HOME

om = O.Model('leap_cube')
rng = np.random.default_rng(0)
N, H = 128, 64
x0 = np.concatenate([LEAP_QPOS_HOME, np.zeros(22)])
ctrl = LEAP_QPOS_HOME[7:]
W = O.spline_weights('cubic', np.linspace(0, 0.64, 4), 0.01 * np.arange(H))
sig = O.mppi_sigma(0.2, True, 4.0, 4, 16)
knots = O.sample_knots(np.tile(ctrl, (4, 1)), rng.standard_normal((N - 1, 4, 16)), sig)
lo = np.array([a['ctrlrange'][0] for a in om.desc['actuators']]); hi = np.array([a['ctrlrange'][1] for a in om.desc['actuators']])
U = O.spline_eval(W, O.clip_knots(knots, lo, hi))
t = time.time(); rs, rsens = om.rollout(x0, U); print('oracle rollout s', time.time() - t)
be = GpuRolloutBackend('leap_cube', N)
t = time.time(); gs, gsens, _ = be.rollout(x0, U); torch.cuda.synchronize(); print('gpu rollout s (incl. first-launch)', time.time() - t)
t = time.time(); gs, gsens, _ = be.rollout(x0, U); torch.cuda.synchronize(); print('gpu rollout s', time.time() - t)
print('nan count', np.isnan(gs).sum())
# single-step parity from oracle states
xs = rs[:, :-1].reshape(-1, 45); us = U[:, 1:].reshape(-1, 1, 16)
nxt = rs[:, 1:].reshape(-1, 45)
g1, s1, _ = GpuRolloutBackend('leap_cube', len(xs)).rollout(xs, us)
e = np.abs(g1[:, 0] - nxt)
print('single-step abs err: qpos max', e[:, :23].max(), 'qvel max', e[:, 23:].max(), 'qvel 99.9pct', np.percentile(e[:, 23:], 99.9), 'median', np.median(e[:, 23:]))
worst = np.argsort(e[:, 23:].max(1))[-5:]
for wi in worst:
    o = om.forward(xs[wi, :23], xs[wi, 23:], us[wi, 0])
    print('  worst', wi, 'err', e[wi, 23:].max(), 'ncon', o['ncon'], 'iters', o['solver_iter'], 'dof', np.argmax(e[wi, 23:]))
print('sensor err (step sensors vs oracle)', np.abs(s1[:, 0] - rsens[:, 1:].reshape(-1, 31)).max())
# rollout-level
er = np.abs(gs - rs)
for h in (0, 1, 3, 7, 15, 31, 63):
    print(f'  h={h}: cube pos err max {er[:, h, :3].max():.2e} median {np.median(er[:, h, :3]):.2e}; finger q err max {er[:, h, 7:23].max():.2e}')
cr = -O.reward_leap(rs, (0, 1, 0, 0)); cg = -O.reward_leap(gs.astype(np.float64), (0, 1, 0, 0))
print('cost: oracle range', cr.min(), cr.max(), 'abs diff median', np.median(np.abs(cr - cg)), 'max', np.abs(cr - cg).max())
print('rank corr', np.corrcoef(np.argsort(np.argsort(cr)), np.argsort(np.argsort(cg)))[0, 1])
nom_o = O.mppi_update(O.clip_knots(knots, lo, hi), -cr, 0.0025); nom_g = O.mppi_update(O.clip_knots(knots, lo, hi), -cg, 0.0025)
print('MPPI nominal diff max', np.abs(nom_o - nom_g).max())
print('stats', be.model.stats())

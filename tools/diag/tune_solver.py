"""Scratch diagnostic (GPU box): effect of the Newton / line-search tolerances on time, iterations and parity."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from judo_amd import engine_model as EM
from oracle import oracle as O
from tests.harness import oracle_plan_step
from judo_amd.controller import make_controller

def run(tol, lstol, cap, N=8192, ref=None):
    EM.SOLVER_TOL, EM.SOLVER_LS_TOL, EM.SOLVER_MAX_ITER = tol, lstol, cap
    ctrl = make_controller('leap_cube', 'mppi'); ctrl.optimizer.config.num_rollouts = N; ctrl.controller_cfg.horizon = 0.64
    ctrl.reset(); ctrl.current_state = ctrl.task.default_state(); ctrl.system_metadata = {'goal_quat': np.array([0., 1, 0, 0])}
    rng = np.random.default_rng(0); noise = rng.standard_normal((N - 1, 4, 16)).astype(np.float32)
    ctrl.optimizer.injected_noise = noise
    nominal0 = ctrl.nominal_knots.copy()
    ctrl.update_action(); torch.cuda.synchronize(); ctrl.model.stats()
    ctrl.nominal_knots = nominal0.copy(); ctrl.update_spline(ctrl.times, nominal0)
    t = time.perf_counter(); ctrl.update_action(); torch.cuda.synchronize(); dt = time.perf_counter() - t
    st = ctrl.model.stats()
    costs = -ctrl.rewards_local
    out = f'tol={tol:g} lstol={lstol:g} cap={cap}: {dt*1e3:7.1f} ms  iters/step={st["newton_iters"]/st["steps"]:.2f} caphits={st["newton_cap_hits"]}'
    if ref is not None:
        d = np.abs(costs[:len(ref["c"])] - ref["c"])
        out += f'  cost err median {np.median(d):.2e} p95 {np.percentile(d,95):.2e} max {d.max():.2e}; nominal err vs oracle {np.abs(ctrl.nominal_knots - ref["nom"]).max():.2e}'
    print(out)
    return ctrl, noise, nominal0

N = 16384
ctrl, noise, nominal0 = run(1e-4, 1e-6, 20, N)
om = O.Model('leap_cube')
r = oracle_plan_step(om, ctrl, nominal0, noise, 'mppi')
ref = {"c": -r["rewards"], "nom": r["nominal"]}
for tol, lstol, cap in [(1e-4, 1e-3, 20), (1e-4, 1e-2, 20), (1e-4, 5e-2, 20), (1e-4, 0.2, 20), (1e-4, 0.5, 20), (3e-4, 0.1, 20), (1e-3, 0.1, 20)]:
    run(tol, lstol, cap, N, ref)

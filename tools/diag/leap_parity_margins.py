"""Observed errors behind the stated tolerances of tests/test_gpu_leap.py (single step, rollouts + costs, plan step), printed so that the tolerances can be kept honest."""
import sys
import numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import test_gpu_leap as T
from judo_amd.rollout_backend import GpuRolloutBackend
from judo_amd.tasks import LEAP_QPOS_HOME, LeapCube
from judo_amd.controller import make_controller
from oracle import oracle as O
from tests.harness import oracle_plan_step

om, knots, U, _ = T._mppi_controls(96)
x0 = np.concatenate([LEAP_QPOS_HOME, np.zeros(22)])
rs, rsens = om.rollout(x0, U)
xs, us, nxt = rs[:, :-1].reshape(-1, 45), U[:, 1:].reshape(-1, 1, 16), rs[:, 1:].reshape(-1, 45)
be = GpuRolloutBackend("leap_cube", len(xs)); g1, s1, _ = be.rollout(xs, us)
e = np.abs(g1[:, 0] - nxt)
print("single step: vel median %.2e p99 %.2e p99.9 %.2e max %.2e | pos max %.2e p99.9 %.2e | sensors max %.2e" % (np.median(e[:, 23:]), np.percentile(e[:, 23:], 99), np.percentile(e[:, 23:], 99.9), e[:, 23:].max(), e[:, :23].max(), np.percentile(e[:, :23], 99.9), np.abs(s1[:, 0] - rsens[:, 1:].reshape(-1, 31)).max()))
N = 192
om, knots, U, _ = T._mppi_controls(N, seed=4)
rs, rsens = om.rollout(x0, U)
gs, gsens, _ = GpuRolloutBackend("leap_cube", N).rollout(x0, U)
err = np.abs(gs - rs)
cr = -O.reward_leap(rs, T.GOAL["goal_quat"]); cg = -LeapCube().reward(gs, gsens, U, T.GOAL)
print("rollouts: first 5 steps max %.2e | cube pos at horizon median %.2e p95 %.2e | cost median %.2e p95 %.2e max %.2e" % (np.abs(gs[:, :5] - rs[:, :5]).max(), np.median(err[:, -1, :3]), np.percentile(err[:, -1, :3], 95), np.median(np.abs(cr - cg)), np.percentile(np.abs(cr - cg), 95), np.abs(cr - cg).max()))
N = 256
rng = np.random.default_rng(2)
ctrl = make_controller("leap_cube", "mppi"); ctrl.optimizer.config.num_rollouts = N; ctrl.controller_cfg.horizon = 0.64
ctrl.reset(); ctrl.current_state = ctrl.task.default_state(); ctrl.system_metadata = dict(T.GOAL)
noise = rng.standard_normal((N - 1, 4, 16)).astype(np.float32); ctrl.optimizer.injected_noise = noise; ctrl.keep_candidates = True
nominal0 = ctrl.nominal_knots.copy(); ctrl.update_action(); torch.cuda.synchronize()
ref = oracle_plan_step(O.Model("leap_cube"), ctrl, nominal0, noise, "mppi")
costs = -ctrl.rewards_local; d = np.abs(costs + ref["rewards"])
exp = O.mppi_update(ref["knots"], -costs.astype(np.float64), 0.0025)
print("plan step: cost median %.2e p95 %.2e max %.2e | nominal vs exact update on own costs %.2e | nominal vs oracle %.2e" % (np.median(d), np.percentile(d, 95), d.max(), np.abs(ctrl.nominal_knots - exp).max(), np.abs(ctrl.nominal_knots - ref["nominal"]).max()))

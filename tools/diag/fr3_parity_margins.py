"""Observed errors behind the stated tolerances of tests/test_gpu_fr3.py."""
import sys
import numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import test_gpu_fr3 as T
from judo_amd.rollout_backend import GpuRolloutBackend
from judo_amd.controller import make_controller
from oracle import oracle as O
from tests.harness import oracle_plan_step
for kind in ("home", "grasping"):
    N, H = 128, 40
    om, task, knots, U = T._controls(N, H, seed=1)
    x0 = task.default_state()
    if kind == "grasping":
        x0 = x0.copy(); x0[7:14] = [0.0, 0.55, 0.0, -2.05, 0.0, 2.6, 0.785]; x0[14:16] = [0.03, 0.025]
    rs, rsens = om.rollout(x0, U)
    gs, gsens, _ = GpuRolloutBackend("fr3_pick", N).rollout(x0, U)
    e = np.abs(gs - rs); es = np.abs(gsens - rsens)
    print(f"{kind}: first step max {np.abs(gs[:, 0] - rs[:, 0]).max():.2e} | all median {np.median(e):.2e} | qpos at horizon p95 {np.percentile(e[:, -1, :16], 95):.2e} max {e[:, -1, :16].max():.2e} | sensors median {np.median(es):.2e} p99 {np.percentile(es, 99):.2e}")
for phase in range(4):
    N = 256
    rng = np.random.default_rng(10 + phase)
    ctrl = make_controller("fr3_pick", "cem"); ctrl.optimizer.config.num_rollouts = N; ctrl.controller_cfg.horizon = 40 * ctrl.task.dt; ctrl.reset()
    x0 = ctrl.task.default_state()
    if phase == 1: x0[2] = 0.05
    elif phase == 2: x0[0:3] = [0.6, 0.4, 0.05]
    elif phase == 3: x0[0:3] = [0.6, 0.4, 0.02]
    ctrl.current_state = x0
    noise = rng.standard_normal((N - 1, 4, 8)).astype(np.float32); ctrl.optimizer.injected_noise = noise; ctrl.keep_candidates = True
    nominal0 = ctrl.nominal_knots.copy(); sigma0 = ctrl.optimizer.sigma.copy()
    ctrl.update_action(); torch.cuda.synchronize()
    ref = oracle_plan_step(O.Model("fr3_pick"), ctrl, nominal0, noise, "cem", sigma0)
    d = np.abs(-ctrl.rewards_local + ref["rewards"])
    print(f"phase {phase}: cost median {np.median(d):.2e} p95 {np.percentile(d, 95):.2e} p99 {np.percentile(d, 99):.2e} max {d.max():.2e} | nominal vs oracle {np.abs(ctrl.nominal_knots - ref['nominal']).max():.2e}")

"""Seeds for tests/test_gpu_fr3.py::test_fr3_plan_step_cem_matches_oracle whose third / fourth best oracle rewards are further apart than the fp32 rollout error, so that the
elite set -- and with it the CEM nominal -- can be compared with the oracle's unconditionally.  Oracle only for the gaps (CPU); prints per phase and seed: gap, and by class
of rollout (finger stacks slammed together or not) the cost error when a GPU is present."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def main():
    import torch

    from judo_amd.controller import make_controller
    from oracle import oracle as O
    from tests.harness import oracle_plan_step

    om = O.Model("fr3_pick")
    for phase in range(4):
        for seed in range(10 + phase, 10 + phase + 24, 4):
            N = 256
            rng = np.random.default_rng(seed)
            ctrl = make_controller("fr3_pick", "cem")
            ctrl.optimizer.config.num_rollouts = N
            ctrl.controller_cfg.horizon = 40 * ctrl.task.dt
            ctrl.reset()
            x0 = ctrl.task.default_state()
            if phase == 1:
                x0[2] = 0.05
            elif phase == 2:
                x0[0:3] = [0.6, 0.4, 0.05]
            elif phase == 3:
                x0[0:3] = [0.6, 0.4, 0.02]
            ctrl.current_state = x0
            noise = rng.standard_normal((N - 1, 4, 8)).astype(np.float32)
            ctrl.optimizer.injected_noise = noise
            nominal0, sigma0 = ctrl.nominal_knots.copy(), ctrl.optimizer.sigma.copy()
            ctrl.update_action()
            torch.cuda.synchronize()
            costs = -ctrl.rewards_local
            ref = oracle_plan_step(om, ctrl, nominal0, noise, "cem", sigma0)
            d = np.abs(costs + ref["rewards"])
            pen = ref["states"][:, :, 14] + ref["states"][:, :, 15]
            slam = pen.min(axis=1) < -5e-4
            order = np.argsort(-ref["rewards"])
            g = np.sort(ref["rewards"])[::-1]
            same = set(np.argsort(costs)[:3]) == set(order[:3])
            print(f"phase {phase} seed {seed}: gap3-4 {g[2] - g[3]:.3e}  d[elite 4] {d[order[:4]].max():.2e}  elites agree {same};  slam {slam.sum():3d}: d p95 {np.percentile(d[slam], 95) if slam.any() else 0:.2e} max {d[slam].max() if slam.any() else 0:.2e} | rest: median {np.median(d[~slam]):.2e} p95 {np.percentile(d[~slam], 95):.2e} max {d[~slam].max():.2e}")


if __name__ == "__main__":
    main()

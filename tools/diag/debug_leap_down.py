"""Scratch diagnostic (GPU box): leap_cube_down rollouts of test_leap_cube_down_variant_runs_on_the_leap_kernels, first non-finite state per kernel generation."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from judo_amd.rollout_backend import GpuRolloutBackend
from tests import xcheck; xcheck.load()  # kernel generations 1 / 2 live in the test build
from judo_amd.tasks import LeapCubeDown
from oracle import oracle as O
t = LeapCubeDown(); om = O.Model("leap_cube_down", scope="cube")
rng = np.random.default_rng(3); N, H = 64, 48
U = t.reset_command[None, None] + 0.3 * np.repeat(rng.standard_normal((N, 4, 16)), H // 4, axis=1)
x0 = t.default_state(); rs, _ = om.rollout(x0, U)
for gen in (3, 2):
    be = GpuRolloutBackend(t.gpu_model(), N); be.model.set_kernel(gen); be.model.set_self_collision(False)
    gs, _, _ = be.rollout(x0, U)
    bad = ~np.isfinite(gs).all(axis=2)
    print("gen", gen, be.model.stats(), "rollouts with non-finite states:", np.nonzero(bad.any(axis=1))[0].tolist())
    for n in np.nonzero(bad.any(axis=1))[0][:3]:
        h = bad[n].argmax()
        print("  rollout", n, "first bad step", h, "oracle cube pos/quat before", np.round(rs[n, h - 1, :7], 4), "kernel before", np.round(gs[n, h - 1, :7], 4))
        x = rs[n, h - 1]; o = om.forward(x[:23], x[23:], U[n, h]); print("  oracle ncon", o["ncon"], "iters", o["solver_iter"])
        np.savez("gpurun_out/r3/leap_down_bad.npz", x=gs[n, h - 1], xo=rs[n, h - 1], u=U[n, h])

"""Scratch diagnostic (GPU box): the closing-gripper rollouts of test_fr3_closed_empty_gripper_keeps_every_pad_contact, error by step."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from judo_amd.rollout_backend import GpuRolloutBackend
from judo_amd.tasks import FR3Pick
from oracle import oracle as O
om, task = O.Model("fr3_pick"), FR3Pick()
rng = np.random.default_rng(0)
rng.uniform(-0.003, 0.0005, (192, 2)); rng.uniform(-0.3, 0.1, (192, 2))
H, M = 40, 32
u = task.reset_command.copy(); u[7] = 0.0
U2 = np.tile(u, (M, H, 1)); U2[:, :, 7] = rng.uniform(-0.02, 0.01, (M, 1))
rs2, _ = om.rollout(task.default_state(), U2)
be2 = GpuRolloutBackend("fr3_pick", M)
gs2, _, _ = be2.rollout(task.default_state(), U2)
print(be2.model.stats())
e = np.abs(gs2 - rs2)
print("finger pos err by step (max over rollouts):", np.array2string(e[:, :, 14:16].max(axis=(0, 2)), precision=2))
print("finger vel err by step:", np.array2string(e[:, :, 29:31].max(axis=(0, 2)), precision=2))
w = np.unravel_index(e[:, :, 14:16].argmax(), e[:, :, 14:16].shape)
print("worst", w, "ctrl", U2[w[0], 0, 7])
i = w[0]
for h in range(10, 30):
    x = rs2[i, h - 1]
    o = om.forward(x[:16], x[16:], U2[i, h])
    print(h, "oracle q", np.round(rs2[i, h, 14:16], 6), "v", np.round(rs2[i, h, 29:31], 4), "| kernel q", np.round(gs2[i, h, 14:16], 6), "v", np.round(gs2[i, h, 29:31], 4), "ncon(before)", o["ncon"], "iters", o["solver_iter"])

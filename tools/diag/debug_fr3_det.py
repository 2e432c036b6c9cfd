"""Scratch diagnostic (GPU box): is the first step of the cooperative fr3 kernel independent of N, H and of the other rollouts?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from tests.test_gpu_fr3 import _controls
from judo_amd.rollout_backend import GpuRolloutBackend
np.set_printoptions(precision=5, suppress=True, linewidth=200)
om, task, knots, U = _controls(128, 40, seed=1)
x0 = task.default_state()
rs, _ = om.rollout(x0, U[33:34, :1])
print('ref            ', rs[0, 0, 16 + 6:])
for N, H, same in [(4, 8, True), (16, 1, True), (4, 1, True), (128, 40, True), (128, 40, False), (128, 1, False), (1, 1, True), (5, 3, True)]:
    Ux = np.repeat(U[33:34, :H], N, axis=0) if same else np.concatenate([U[33:34, :H], U[:N - 1, :H]])
    for rep in range(2):
        gs, _, _ = GpuRolloutBackend("fr3_pick", N).rollout(x0, Ux)
        print(f'N={N:4d} H={H:3d} same={same!s:5s}', gs[0, 0, 16 + 6:], ' max|err|', np.abs(gs[0, 0] - rs[0, 0]).max())

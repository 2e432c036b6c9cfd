"""A plugin task + plugin rollout backend with closed-form numpy arithmetic, used on BOTH sides of the controller golden:
`tools/gen_golden_controller.py` drives the reference's own `Controller.update_action` / `update_traces`
(judo/controller/controller.py:210-363) with them, `tests/test_gpu_controller.py` drives `judo_amd.Controller` with them.
Nothing here is reference code: it is the stand-in for "some third-party plugin" whose only job is to be identical on both sides.

Dimensions are those of the cartpole model (nq = nv = 2, nu = 1, nsensordata = 6, two `trace*` framepos sensors at 0 and 3,
ctrlrange +-1.8, dt = 0.04) so that the build's side can subclass its `Cartpole` task (plugins register on a shipped model).
"""

from __future__ import annotations

import numpy as np

NQ, NV, NU, NS, DT = 2, 2, 1, 6, 0.04
CTRLRANGE = np.array([[-1.8, 1.8]])
SENSOR_ADR = [0, 3]
A = np.array([[1.0, 0.0, DT, 0.0], [0.0, 1.0, 0.0, DT], [-0.08, 0.02, 0.97, 0.0], [0.03, -0.12, 0.0, 0.95]])
B = np.array([0.0, 0.0, 0.09, -0.05])
GOAL = np.array([0.4, -0.2])


def rollout_numpy(x0: np.ndarray, controls: np.ndarray):
    """x_{h+1} = A x_h + B u_h; states[n, h] is the state after control h (the RolloutBackend contract); sensors[n, h] describes the
    state BEFORE that step (MuJoCo's sensordata lag), two 3-d points."""
    controls = np.asarray(controls, dtype=np.float64)
    N, H, _ = controls.shape
    x = np.tile(np.asarray(x0, dtype=np.float64), (N, 1)) if np.ndim(x0) == 1 else np.array(x0, dtype=np.float64)
    states = np.zeros((N, H, NQ + NV))
    sensors = np.zeros((N, H, NS))
    for h in range(H):
        sensors[:, h, 0] = x[:, 0]
        sensors[:, h, 1] = 0.1 * x[:, 2]
        sensors[:, h, 2] = x[:, 1]
        sensors[:, h, 3] = x[:, 0] + np.sin(x[:, 1])
        sensors[:, h, 4] = -0.2 * x[:, 3]
        sensors[:, h, 5] = np.cos(x[:, 1])
        x = x @ A.T + controls[:, h, 0:1] * B[None, :]
        states[:, h] = x
    return states, sensors


def reward_numpy(states, sensors, controls) -> np.ndarray:
    states, controls = np.asarray(states, dtype=np.float64), np.asarray(controls, dtype=np.float64)
    d = states[..., :2] - GOAL
    return -(np.sum(d * d, axis=(-1, -2)) + 0.05 * np.sum(states[..., 2:] ** 2, axis=(-1, -2)) + 0.01 * np.sum(controls**2, axis=(-1, -2)))


class NumpyBackend:
    """A RolloutBackend plugin with the reference's numpy signature (judo/utils/rollout_backend.py:10-46)."""

    def __init__(self, num_threads: int) -> None:
        self.num_threads = num_threads
        self.calls = 0

    def rollout(self, x0, controls, last_policy_output=None):
        assert controls.shape[0] == self.num_threads, (controls.shape, self.num_threads)
        self.calls += 1
        s, y = rollout_numpy(x0, controls)
        return s, y, None

    def update(self, num_threads: int) -> None:
        self.num_threads = num_threads

"""RolloutBackend on the GPU -- the drop-in seam (judo/utils/rollout_backend.py:10-46).

`GpuRolloutBackend.rollout(x0, controls, last_policy_output=None) -> (states, sensors, None)` has the contract of
`MJRolloutBackend.rollout` (judo/utils/mj_rollout_backend.py:45-88): x0 (nq+nv,) is tiled over the batch (or given
per rollout), controls are (num_threads, T, nu); states[n,t] is the state after applying controls[n,t]; sensors[n,t] is
the sensordata of the step that produced it.  One lane per rollout replaces one OS thread + one MjModel copy per rollout;
`update(num_threads)` therefore only records the new batch size.
"""

from __future__ import annotations

from abc import ABC, abstractmethod

import numpy as np
import torch

from judo_amd import _lib
from judo_amd.device import GpuModel, current_stream_ptr, f32


class RolloutBackend(ABC):
    num_threads: int

    @abstractmethod
    def rollout(self, x0: np.ndarray, controls: np.ndarray, last_policy_output: np.ndarray | None = None):
        ...

    @abstractmethod
    def update(self, num_threads: int) -> None:
        ...


class GpuRolloutBackend(RolloutBackend):
    def __init__(self, model: GpuModel | str, num_threads: int) -> None:
        self.model = model if isinstance(model, GpuModel) else GpuModel(model)
        self.num_threads = int(num_threads)

    def rollout_device(self, x0: torch.Tensor, controls: torch.Tensor, want_states: bool = True, want_sensors: bool = True):
        """Device tensors in, device tensors out (fp32)."""
        gm = self.model
        if controls.ndim != 3:
            raise ValueError(f"controls must be (num_threads, T, nu), got {tuple(controls.shape)}")
        N, H, nu = (int(s) for s in controls.shape)
        if nu != gm.nu:
            raise ValueError(f"controls.shape[-1] = {nu} != nu = {gm.nu}")
        batched = int(x0.ndim == 2)
        if x0.shape[-1] != gm.nx:
            raise ValueError(f"x0 must have {gm.nx} = nq+nv entries, got {tuple(x0.shape)}")
        if batched and x0.shape[0] != N:
            raise ValueError(f"batched x0 has {x0.shape[0]} rows but controls has {N}")
        if not want_states and not want_sensors:
            raise ValueError("nothing to compute: both outputs disabled")
        x0 = x0.to(torch.float32).contiguous()
        controls = controls.to(torch.float32).contiguous()
        states = torch.empty((N, H, gm.nx), dtype=torch.float32, device=gm.device) if want_states else None
        sensors = torch.empty((N, H, gm.ns), dtype=torch.float32, device=gm.device) if want_sensors else None
        st = _lib.lib().jh_rollout_materialize(gm.handle, _lib.ptr(x0), batched, _lib.ptr(controls), N, H, _lib.ptr(states), _lib.ptr(sensors), current_stream_ptr())
        _lib.check(st, "jh_rollout_materialize")
        return states, sensors

    def rollout(self, x0: np.ndarray, controls: np.ndarray, last_policy_output: np.ndarray | None = None):
        x0 = np.asarray(x0)
        controls = np.asarray(controls)
        if controls.ndim != 3 or controls.shape[0] != (x0.shape[0] if x0.ndim == 2 else controls.shape[0]):
            raise ValueError("controls must be (num_threads, T, nu) with one row per rollout")
        states, sensors = self.rollout_device(f32(x0, self.model.device), f32(controls, self.model.device))
        return states.cpu().numpy().astype(np.float64), sensors.cpu().numpy().astype(np.float64), None

    def update(self, num_threads: int) -> None:
        self.num_threads = int(num_threads)

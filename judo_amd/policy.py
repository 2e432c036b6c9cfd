"""Batched policy step of the Spot policy rollout on the GPU (the policy half of `mujoco_extensions/policy_rollout`:
System::setObservation + System::policyInference, mujoco_extensions/system/system_class.cpp:125-238).

    policy = SpotLocomotionPolicy()                       # actor weights extracted from the reference's ONNX file
    control, policy_out = policy.step(states, commands, last_policy_output, layout)

`states` (N, nq+nv), `commands` (N, 25) in the layout `SpotBase.task_to_sim_ctrl` produces (judo/tasks/spot/spot_base.py:325-391),
`last_policy_output` (N, 12).  Returns the 19 joint position targets the plant applies for the next `physics_substeps` steps and
the new policy output.  numpy in -> numpy out; torch device tensors in -> torch device tensors out.  The physics between two
policy steps needs the Spot model, which the engine kernels do not cover yet.
"""

from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass

import numpy as np
import torch

from judo_amd import _lib
from judo_amd.device import current_stream_ptr, f32, require_gpu

POLICY_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "models", "spot_locomotion_policy.npz")


@dataclass(frozen=True)
class SpotStateLayout:
    """Where the free base and the 19 joints (12 leg + 7 arm, MuJoCo order) sit in a state row (system_class.cpp:137-143)."""

    nq: int
    nv: int
    base_qpos: int = 0
    base_qvel: int = 0
    leg_qpos: int = 7
    leg_qvel: int = 6


class SpotLocomotionPolicy:
    OBS, ACT, NJ, NCMD = 84, 12, 19, 25

    def __init__(self, path: str = POLICY_PATH, device: torch.device | None = None) -> None:
        self.device = device or require_gpu()
        w = np.load(path)
        self._host = [np.ascontiguousarray(w[f"W{i}"], dtype=np.float32) for i in range(4)] + [np.ascontiguousarray(w[f"b{i}"], dtype=np.float32) for i in range(4)]
        if [a.shape for a in self._host[:4]] != [(512, 84), (256, 512), (128, 256), (12, 128)]:
            raise ValueError("spot locomotion actor must be 84-512-256-128-12")
        wp = (C.c_void_p * 4)(*[a.ctypes.data for a in self._host[:4]])
        bp = (C.c_void_p * 4)(*[a.ctypes.data for a in self._host[4:]])
        handle = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().jh_policy_create(wp, bp, C.byref(handle)), "jh_policy_create")
        self.handle = handle

    def step(self, states, commands, last_policy_output, layout: SpotStateLayout):
        as_numpy = not isinstance(states, torch.Tensor)
        st = f32(states, self.device) if as_numpy else states.to(torch.float32).contiguous()
        cmd = f32(commands, self.device) if not isinstance(commands, torch.Tensor) else commands.to(torch.float32).contiguous()
        out = (f32(last_policy_output, self.device) if not isinstance(last_policy_output, torch.Tensor) else last_policy_output.to(torch.float32)).clone().contiguous()
        if st.ndim != 2 or st.shape[1] != layout.nq + layout.nv:
            raise ValueError(f"states must be (N, {layout.nq + layout.nv}), got {tuple(st.shape)}")
        N = int(st.shape[0])
        if tuple(cmd.shape) != (N, self.NCMD) or tuple(out.shape) != (N, self.ACT):
            raise ValueError(f"commands must be ({N}, 25) and last_policy_output ({N}, 12), got {tuple(cmd.shape)} and {tuple(out.shape)}")
        L = _lib.lib()
        control = torch.empty((N, self.NJ), dtype=torch.float32, device=self.device)
        scratch = torch.empty(int(L.jh_policy_scratch_floats(N)), dtype=torch.float32, device=self.device)
        s = L.jh_policy_step(self.handle, _lib.ptr(st), int(st.shape[1]), layout.nq, layout.base_qpos, layout.base_qvel, layout.leg_qpos, layout.leg_qvel,
                             _lib.ptr(cmd), _lib.ptr(out), _lib.ptr(control), _lib.ptr(scratch), N, current_stream_ptr())
        _lib.check(s, "jh_policy_step")
        self.last_observation = scratch[: N * self.OBS].view(N, self.OBS)
        if as_numpy:
            return control.cpu().numpy().astype(np.float64), out.cpu().numpy().astype(np.float64)
        return control, out

    def __del__(self) -> None:
        try:
            if getattr(self, "handle", None):
                _lib.lib().jh_policy_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

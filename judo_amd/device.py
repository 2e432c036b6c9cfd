"""Device plumbing: PyTorch-ROCm is used only as the container for device memory, streams and RCCL.

`GpuModel` owns one `jh_model` handle (model constants resident in HBM) per (task, device).
"""

from __future__ import annotations

import ctypes as C
import os

import numpy as np
import torch

from judo_amd import _lib
from judo_amd.models import layout, load_description, pack_model


def require_gpu() -> torch.device:
    if not torch.cuda.is_available():
        raise RuntimeError("judo_amd needs an MI355X (HIP device): torch.cuda.is_available() is False and there is no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def current_stream_ptr() -> int:
    """The current HIP stream of the current device as an integer (what the C ABI takes).  `torch.cuda.current_stream()` builds a Stream object through several
    layers of Python (10 us per call, a tenth of a small plan step); the raw accessor behind it returns the same handle in well under a microsecond."""
    if _raw_stream is not None:
        return int(_raw_stream(torch.cuda.current_device()))
    return torch.cuda.current_stream().cuda_stream


class HipEvent:
    """A timing event on the launch stream (jh_event_*): what `Controller.record_kernel_events` brackets the kernels with.  Same two methods as torch.cuda.Event, but the
    handle exists from construction on, so the library can record it inside a multi-kernel call (jh_plan_step)."""

    __slots__ = ("handle",)

    def __init__(self) -> None:
        h = C.c_void_p()
        _lib.check(_lib.lib().jh_event_create(C.byref(h)), "jh_event_create")
        self.handle = h.value

    def record(self, stream: int | None = None) -> None:
        _lib.check(_lib.lib().jh_event_record(self.handle, current_stream_ptr() if stream is None else stream), "jh_event_record")

    def elapsed_time(self, other: "HipEvent") -> float:
        ms = C.c_float()
        _lib.check(_lib.lib().jh_event_elapsed_ms(self.handle, other.handle, C.byref(ms)), "jh_event_elapsed_ms")
        return float(ms.value)

    def __del__(self) -> None:
        try:
            if self.handle:
                _lib.lib().jh_event_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


def f32(x, device) -> torch.Tensor:
    """Host array -> contiguous fp32 device tensor."""
    return torch.as_tensor(np.ascontiguousarray(x, dtype=np.float32)).to(device, non_blocking=False)


class GpuModel:
    """Model constants of one task on one GPU (the counterpart of the per-thread MjModel copies,
    judo/utils/mj_rollout_backend.py:38-43 -- here a single read-only image shared by every lane)."""

    def __init__(self, task: str | dict, device: torch.device | None = None) -> None:
        self.desc = load_description(task) if isinstance(task, str) else task
        self.task = self.desc["task"]
        lay = layout(self.desc)
        self.nq, self.nv, self.nu, self.ns = lay.nq, lay.nv, lay.nu, lay.ns
        self.nx = self.nq + self.nv
        self.dt = float(self.desc["option"]["timestep"])
        self.device = device if device is not None else require_gpu()
        blob = pack_model(self.desc)
        self._blob = blob
        handle = C.c_void_p()
        buf = C.create_string_buffer(blob, len(blob))
        st = _lib.lib().jh_model_create(buf, len(blob), self.device.index or 0, C.byref(handle))
        _lib.check(st, "jh_model_create")
        self.handle = handle
        self._max_knots_cache: dict[int, int] = {}
        self.kernel_generation = 3 if self.desc.get("family", self.task) in ("leap_cube", "fr3_pick") else 2  # library defaults (jh_model_create)
        self._self_collision_requested = True
        gen = os.environ.get("JUDO_AMD_FR3_KERNEL")  # diagnostic: run the fr3_pick tests / benches on another kernel generation of the library
        if gen and self.task == "fr3_pick":
            self.set_kernel(int(gen))
        self.self_collision = self.desc.get("family", self.task) == "leap_cube"  # library default: on where the kernel models it
        # contacts a rollout can hold.  The leap kernel exists in two builds (48, all in LDS; 64 with 16 in global memory, 2.8 % slower): the headline model drops 2e-6
        # contacts per rollout-step at 48, the SHIPPED workloads of the caged / primitive-hand variants 2e-4 .. 4e-4 (tools/diag/shipped_config_stats.py): those take 64
        self.closed_form = self.task in ("cartpole", "cylinder_push")  # jh_simple.hip: rollout kernels of ~50 us -- the plan step runs as one launch that reads its host block in place
        self.contact_capacity = 0
        if self.desc.get("family", self.task) == "leap_cube":
            self.set_contact_capacity(48 if self.task == "leap_cube" else 64)

    def set_kernel(self, generation: int) -> None:
        """Select the articulated-engine kernel generation (3 = cooperative, two waves per SIMD: the default; 2 = cooperative, one wave per SIMD; 1 = one lane per rollout)."""
        _lib.check(_lib.lib().jh_model_set_kernel(self.handle, int(generation)), "jh_model_set_kernel")
        self._max_knots_cache.clear()
        self.kernel_generation = int(generation)
        # only generation 3 of the leap family models the hand's own contacts: what bench.py / tests report must follow the kernel actually selected
        self.self_collision = self._self_collision_requested and self.kernel_generation == 3 and self.desc.get("family", self.task) == "leap_cube"

    def set_contact_capacity(self, contacts: int) -> None:
        """leap_cube family: 48 or 64 contacts per rollout (`jh_model_set_contact_capacity`)."""
        _lib.check(_lib.lib().jh_model_set_contact_capacity(self.handle, int(contacts)), "jh_model_set_contact_capacity")
        self._max_knots_cache.clear()
        self.contact_capacity = int(contacts)

    def set_self_collision(self, on: bool) -> None:
        """leap_cube, kernel generation 3: model the hand's own contacts too (default) or the cube's alone."""
        _lib.check(_lib.lib().jh_model_set_self_collision(self.handle, int(bool(on))), "jh_model_set_self_collision")
        self._self_collision_requested = bool(on)
        self.self_collision = bool(on) and self.kernel_generation == 3 and self.desc.get("family", self.task) == "leap_cube"

    def trace_layout(self) -> tuple[int, int, bool]:
        """(first sensor address, number of floats, buffer is column-major) of the trace sensors the fused kernel can write for every rollout
        (`jh_model_trace_layout`); 0 floats: none."""
        out = (C.c_int * 3)()
        _lib.check(_lib.lib().jh_model_trace_layout(self.handle, out), "jh_model_trace_layout")
        return int(out[0]), int(out[1]), bool(out[2])

    def limits(self) -> tuple[int, int, int, int]:
        """`jh_model_limits`: (largest fused knot count, JH_MAX_KNOT_DIM, JH_MAX_ELITES, contact capacity per rollout)."""
        out = (C.c_int * 4)()
        _lib.check(_lib.lib().jh_model_limits(self.handle, out), "jh_model_limits")
        return tuple(int(v) for v in out)

    @property
    def max_fused_knots(self) -> int:
        """Largest knot count the fused rollout kernel of this model accepts (`jh_model_limits`)."""
        out = (C.c_int * 4)()
        _lib.check(_lib.lib().jh_model_limits(self.handle, out), "jh_model_limits")
        return int(out[0])

    def max_fused_knots_at(self, H: int) -> int:
        """Largest K `jh_rollout_cost` accepts for a launch of H steps (the one-lane kernels stage W and the knots in LDS: it depends on H)."""
        H = int(H)
        k = self._max_knots_cache.get(H)  # (asked once per plan step; the answer changes with the kernel generation / contact capacity only: those setters clear the cache)
        if k is None:
            k = int(_lib.lib().jh_model_max_fused_knots(self.handle, H))
            if k < 0:
                _lib.check(k, "jh_model_max_fused_knots")
            self._max_knots_cache[H] = k
        return k

    def stats(self, reset: bool = True) -> dict:
        """Diagnostic counters of the articulated-body kernels (synchronises)."""
        out = (C.c_int * 8)()
        _lib.check(_lib.lib().jh_model_stats(self.handle, out, int(reset)), "jh_model_stats")
        u = [v & 0xFFFFFFFF for v in out]  # 32-bit counters: unsigned
        return {"contact_overflow": u[0], "newton_cap_hits": u[1], "newton_iters": u[2], "steps": u[3], "wave_newton_iters": u[4], "wave_steps": u[5], "overflow_pool_fallbacks": u[6]}

    def __del__(self) -> None:
        try:
            if getattr(self, "handle", None):
                _lib.lib().jh_model_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

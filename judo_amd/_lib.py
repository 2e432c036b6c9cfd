"""ctypes binding of `libjudo_amd.so` (the C ABI declared in `include/judo_amd.h`).

There is no fallback: if the HIP library is missing the import of any compute entry point raises.
Build it with `python -c "import __graft_entry__ as g; g.build()"` (hipcc --offload-arch=gfx950).
"""

from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("JUDO_AMD_LIB") or os.path.join(_HERE, "libjudo_amd.so")  # JUDO_AMD_LIB: alternative build of the same C ABI

_lib: C.CDLL | None = None

f32p = C.c_void_p  # device pointers travel as integers (tensor.data_ptr())

_SIGNATURES = {
    "jh_last_error": (C.c_char_p, []),
    "jh_version": (C.c_int, []),
    "jh_model_create": (C.c_int, [C.c_void_p, C.c_size_t, C.c_int, C.POINTER(C.c_void_p)]),
    "jh_model_destroy": (None, [C.c_void_p]),
    "jh_model_dims": (C.c_int, [C.c_void_p, C.POINTER(C.c_int)]),
    "jh_model_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.c_int]),
    "jh_model_set_kernel": (C.c_int, [C.c_void_p, C.c_int]),
    "jh_model_limits": (C.c_int, [C.c_void_p, C.POINTER(C.c_int)]),
    "jh_register_xcheck": (C.c_int, [C.c_void_p]),
    "jh_model_max_fused_knots": (C.c_int, [C.c_void_p, C.c_int]),
    "jh_model_set_self_collision": (C.c_int, [C.c_void_p, C.c_int]),
    "jh_model_set_contact_capacity": (C.c_int, [C.c_void_p, C.c_int]),
    "jh_model_trace_layout": (C.c_int, [C.c_void_p, C.POINTER(C.c_int)]),
    "jh_trace_gather": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "jh_upload_async": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "jh_download_wait": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "jh_download_begin": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "jh_download_end": (C.c_int, []),
    "jh_rollout_cost": (C.c_int, [C.c_void_p, f32p, f32p, f32p, C.c_int, f32p, f32p, f32p, f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, f32p, f32p, C.c_void_p]),
    "jh_rollout_cost_traced": (C.c_int, [C.c_void_p, f32p, f32p, f32p, C.c_int, f32p, f32p, f32p, f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, f32p, f32p, f32p, C.c_void_p]),
    "jh_rollout_materialize": (C.c_int, [C.c_void_p, f32p, C.c_int, f32p, C.c_int, C.c_int, f32p, f32p, C.c_void_p]),
    "jh_task_reward": (C.c_int, [C.c_void_p, f32p, f32p, f32p, f32p, C.c_int, C.c_int, C.c_int, f32p, C.c_void_p]),
    "jh_noise_normal": (C.c_int, [C.c_ulonglong, C.c_uint, C.c_int, C.c_int, C.c_int, f32p, C.c_int, C.c_void_p]),
    "jh_sample_knots": (C.c_int, [f32p, f32p, C.c_int, f32p, f32p, C.c_int, C.c_int, C.c_int, C.c_int, f32p, C.c_void_p]),
    "jh_spline_controls": (C.c_int, [f32p, f32p, f32p, f32p, C.c_int, f32p, f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, f32p, C.c_void_p]),
    "jh_knot_moments": (C.c_int, [f32p, f32p, f32p, C.c_int, f32p, f32p, f32p, C.c_int, C.c_int, C.c_int, C.c_int, f32p, C.c_void_p]),
    "jh_policy_create": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    "jh_policy_destroy": (None, [C.c_void_p]),
    "jh_policy_scratch_floats": (C.c_size_t, [C.c_int]),
    "jh_policy_step": (C.c_int, [C.c_void_p, f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, f32p, f32p, f32p, f32p, C.c_int, C.c_void_p]),
    "jh_tree_create": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]),
    "jh_tree_destroy": (None, [C.c_void_p]),
    "jh_tree_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.c_int]),
    "jh_tree_dims": (C.c_int, [C.c_void_p, C.POINTER(C.c_int)]),
    "jh_tree_set_self_collision": (C.c_int, [C.c_void_p, C.c_int]),
    "jh_tree_substeps": (C.c_int, [C.c_void_p, f32p, f32p, f32p, C.c_int, C.c_int, f32p, f32p, C.c_void_p]),
    "jh_policy_rollout_scratch_floats": (C.c_size_t, [C.c_int]),
    "jh_policy_rollout": (C.c_int, [C.c_void_p, C.c_void_p, f32p, C.c_int, f32p, f32p, f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, f32p, f32p, f32p, C.POINTER(C.c_int), C.c_void_p]),
    "jh_update_scratch_floats": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "jh_mppi_partial": (C.c_int, [f32p, f32p, f32p, f32p, C.c_int, f32p, f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, f32p, f32p, C.c_void_p]),
    "jh_mppi_merge": (C.c_int, [f32p, C.c_int, C.c_int, C.c_int, C.c_float, f32p, C.c_void_p]),
    "jh_topk_partial": (C.c_int, [f32p, f32p, f32p, f32p, C.c_int, f32p, f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, f32p, f32p, C.c_void_p]),
    "jh_update_fused_scratch_floats": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "jh_update_fused": (C.c_int, [f32p, f32p, f32p, f32p, C.c_int, f32p, f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, C.c_int, f32p, C.c_int, C.c_int,
                                  f32p, f32p, f32p, f32p, C.c_void_p]),
    "jh_plan_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, f32p, C.c_int, f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, f32p, f32p, f32p,
                               C.c_int, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, f32p, f32p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "jh_shard_record_floats": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "jh_update_shard": (C.c_int, [f32p, f32p, f32p, f32p, C.c_int, f32p, f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, C.c_int, f32p, C.c_int, C.c_int,
                                  f32p, f32p, C.c_void_p]),
    "jh_shard_merge": (C.c_int, [f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int, f32p, f32p, f32p, C.c_void_p]),
    "jh_plan_step_shard": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, f32p, C.c_int, f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, f32p, f32p, f32p,
                                     C.c_int, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, f32p, f32p, C.c_void_p, C.c_void_p]),
    "jh_plan_merge": (C.c_int, [f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int, f32p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "jh_event_create": (C.c_int, [C.POINTER(C.c_void_p)]),
    "jh_event_destroy": (None, [C.c_void_p]),
    "jh_event_record": (C.c_int, [C.c_void_p, C.c_void_p]),
    "jh_stream_wait_event": (C.c_int, [C.c_void_p, C.c_void_p]),
    "jh_event_elapsed_ms": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_float)]),
    "jh_elite_merge": (C.c_int, [f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, f32p, f32p, C.c_void_p]),
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)

# compile-time limits of the library (include/judo_amd.h; tests/test_host.py checks that the two agree)
MAX_KNOT_DIM = 512  # JH_MAX_KNOT_DIM: K * nu
MAX_ELITES = 32  # JH_MAX_ELITES: CEM elites / trace elites
MAX_TASK_PARAMS = 32  # JH_MAX_TASK_PARAMS


class JudoAmdError(RuntimeError):
    """A HIP/runtime failure reported by libjudo_amd.so."""


def lib() -> C.CDLL:
    """Load the shared library (once).  Raises ImportError when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: the MI355X engine has no CPU fallback. "
                "Build it with `python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc)."
            )
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError if the .so does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(status: int, what: str) -> None:
    """Map a jh_status to the reference's error behaviour: shape/argument errors -> ValueError
    (the conditions judo/utils/mj_rollout_backend.py:78-82 asserts), runtime errors -> RuntimeError."""
    if status == 0:
        return
    msg = lib().jh_last_error().decode("utf-8", "replace")
    if status == -1:
        raise ValueError(f"{what}: {msg}")
    raise JudoAmdError(f"{what} failed (status {status}): {msg}")


def ptr(t) -> int | None:
    """data_ptr of a torch tensor (None passes a NULL pointer; an int is a raw device address and passes through)."""
    if t is None or isinstance(t, int):
        return t
    return t.data_ptr()

"""MuJoCo probe -- TEST INFRASTRUCTURE (only tests/, tools/ and bench.py's cpu_baseline leg import this).

The reference's physics is the third-party wheel `mujoco` 3.5.0 (pyproject.toml:33), absent from this image and from the GPU box.  The day
it is importable -- and the reference's MJCF files are reachable (JUDO_REFERENCE_ROOT, default /root/reference) -- this module
  * times the reference's own CPU rollout path, `mujoco.rollout.Rollout(nthread=cores)` exactly as judo/utils/mj_rollout_backend.py:36-88 drives it
    (one model copy + one MjData per rollout, time prepended to the state), for `bench.py`'s `cpu_baseline` (kind "reference");
  * produces `mj_step` trajectories for tools/gen_golden_mujoco.py, which pins oracle/jo_engine.c.
Until then `available()` is False and every caller says so instead of substituting anything.
"""

from __future__ import annotations

import importlib.util
import os
import time

import numpy as np

MESH_FREE_TASKS = ("cartpole", "cylinder_push")  # their MJCF needs no mesh assets (the leap / fr3 / spot meshes are not in the repository)


def find_mujoco():
    """The mujoco module, or None when the wheel is not installed."""
    if importlib.util.find_spec("mujoco") is None:
        return None
    try:
        import mujoco
        import mujoco.rollout  # noqa: F401

        return mujoco
    except Exception:
        return None


def reference_xml(task: str) -> str | None:
    root = os.environ.get("JUDO_REFERENCE_ROOT", "/root/reference")
    path = os.path.join(root, "judo", "models", "xml", f"{task}.xml")
    return path if os.path.exists(path) else None


def available(task: str = "cartpole") -> bool:
    return find_mujoco() is not None and reference_xml(task) is not None


def rollout(task: str, x0: np.ndarray, controls: np.ndarray, nthread: int | None = None, xml_path: str | None = None):
    """`MJRolloutBackend.rollout` (judo/utils/mj_rollout_backend.py:45-88): states (N, H, nq+nv) after each control, sensors (N, H, nsensordata)."""
    mujoco = find_mujoco()
    if mujoco is None:
        raise RuntimeError("the mujoco wheel is not installed")
    from copy import deepcopy

    from mujoco.rollout import Rollout

    path = xml_path or reference_xml(task)
    if path is None:
        raise RuntimeError(f"no MJCF for {task}: set JUDO_REFERENCE_ROOT")
    model = mujoco.MjModel.from_xml_path(path)
    controls = np.ascontiguousarray(controls, dtype=np.float64)
    N = controls.shape[0]
    x0 = np.asarray(x0, dtype=np.float64)
    if x0.ndim == 1:
        x0 = np.tile(x0, (N, 1))
    models = [deepcopy(model) for _ in range(N)]
    datas = [mujoco.MjData(m) for m in models]
    full = np.concatenate([np.zeros((N, 1)), x0], axis=-1)
    with Rollout(nthread=nthread or os.cpu_count() or 1) as ro:
        states, sensors = ro.rollout(models, datas, full, controls)
    return np.array(states)[..., 1:], np.array(sensors)


def time_reference_rollouts(task: str, x0: np.ndarray, controls: np.ndarray, nthread: int) -> dict:
    """Wall time of the reference's CPU rollout of `controls` (N, H, nu) on `nthread` threads (model copies and MjData built outside the timed region,
    as `MJRolloutBackend.__init__` does)."""
    mujoco = find_mujoco()
    from copy import deepcopy

    from mujoco.rollout import Rollout

    model = mujoco.MjModel.from_xml_path(reference_xml(task))
    N = controls.shape[0]
    models = [deepcopy(model) for _ in range(N)]
    datas = [mujoco.MjData(m) for m in models]
    full = np.concatenate([np.zeros((N, 1)), np.tile(np.asarray(x0, dtype=np.float64), (N, 1))], axis=-1)
    with Rollout(nthread=nthread) as ro:
        ro.rollout(models, datas, full, controls)  # warm-up (thread pool start)
        t0 = time.perf_counter()
        ro.rollout(models, datas, full, controls)
        dt = time.perf_counter() - t0
    return {"rollouts": N, "seconds": dt, "mujoco": mujoco.__version__}

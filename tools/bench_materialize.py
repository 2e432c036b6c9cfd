#!/usr/bin/env python3
"""Materialise-mode throughput (drop-in RolloutBackend.rollout on device tensors): the HBM-bound exhibit of SURVEY.md 8(d).
Algorithmic bytes per rollout = 4*H*(nx + ns + nu) read/written once (+ x0)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from judo_amd import _lib
if os.environ.get("JH_LIB"): _lib.LIB_PATH = os.environ["JH_LIB"]
from judo_amd.rollout_backend import GpuRolloutBackend

out = []
for task, N, H in (("cartpole", 1 << 20, 64), ("cylinder_push", 1 << 20, 64), ("leap_cube", 65536, 64)):
    be = GpuRolloutBackend(task, N)
    gm = be.model
    x0 = torch.zeros(gm.nx, device=gm.device)
    if task == "leap_cube":
        from judo_amd.tasks import LEAP_QPOS_HOME
        x0[:23] = torch.tensor(LEAP_QPOS_HOME, dtype=torch.float32)
    elif task == "cylinder_push":
        x0[:4] = torch.tensor([1.0, 0.0, 1.08, 1.68])
    else:
        x0[:2] = torch.tensor([1.0, 3.14159])
    U = 0.5 * torch.randn((N, H, gm.nu), device=gm.device)
    if task == "leap_cube":
        U += x0[7:23]
    for _ in range(2):
        be.rollout_device(x0, U)
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 5
    ev0.record()
    for _ in range(reps):
        s, y = be.rollout_device(x0, U)
    ev1.record(); torch.cuda.synchronize()
    ms = ev0.elapsed_time(ev1) / reps
    nbytes = 4 * N * H * (gm.nx + gm.ns + gm.nu)
    out.append({"task": task, "rollouts": N, "H": H, "ms": ms, "rollouts_per_s": N / ms * 1e3, "algorithmic_bytes": nbytes, "achieved_GBps": nbytes / ms / 1e6,
                "frac_of_8TBps": nbytes / ms / 1e6 / 8000})
    del s, y, U
print(json.dumps(out, indent=1))

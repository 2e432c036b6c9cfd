"""Scratch diagnostic (GPU box): are two identical materialise launches bit-identical?  (per task, both kernel generations)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from judo_amd.rollout_backend import GpuRolloutBackend
from tests import xcheck; xcheck.load()  # kernel generations 1 / 2 live in the test build
from judo_amd.tasks import get_registered_tasks
for task, N, H in (("leap_cube", 512, 64), ("fr3_pick", 512, 40), ("cartpole", 512, 64)):
    t = get_registered_tasks()[task][0]()
    x0 = torch.as_tensor(np.asarray(t.default_state(), dtype=np.float32)).cuda()
    g = torch.Generator(device="cuda").manual_seed(1)
    U = (0.3 * torch.randn((N, H, t.nu), device="cuda", generator=g) + torch.as_tensor(np.asarray(t.optimizer_warm_start(), dtype=np.float32)).cuda()).contiguous()
    for gen in ((2, 1) if task != "cartpole" else (2,)):
        be = GpuRolloutBackend(task, N)
        if task != "cartpole": be.model.set_kernel(gen)
        outs = []
        for rep in range(3):
            s, y = be.rollout_device(x0, U); torch.cuda.synchronize(); outs.append((s.clone(), y.clone()))
        same = [bool(torch.equal(outs[0][0], outs[i][0]) and torch.equal(outs[0][1], outs[i][1])) for i in (1, 2)]
        d = (outs[0][0] - outs[1][0]).abs()
        first = int((d.amax(dim=(0, 2)) > 0).nonzero()[0]) if d.max() > 0 else -1
        print(f"{task} gen {gen}: bit-identical repeats {same}; max |diff| {float(d.max()):.3e}; first differing step {first}; rollouts differing {int((d.amax(dim=(1, 2)) > 0).sum())}/{N}")
print("placement test: the same controls, rollouts shifted by 1 / 2 / 3 positions (different lane row and wave-mates)")
for task, N, H in (("leap_cube", 256, 64), ("fr3_pick", 256, 40)):
    t = get_registered_tasks()[task][0]()
    x0 = torch.as_tensor(np.asarray(t.default_state(), dtype=np.float32)).cuda()
    g = torch.Generator(device="cuda").manual_seed(2)
    U = (0.3 * torch.randn((N, H, t.nu), device="cuda", generator=g) + torch.as_tensor(np.asarray(t.optimizer_warm_start(), dtype=np.float32)).cuda()).contiguous()
    be = GpuRolloutBackend(task, N)
    s0, _ = be.rollout_device(x0, U)
    for sh in (1, 2, 3, 4):
        Us = torch.cat([U[:sh] * 0 + U[:1], U[:-sh]]).contiguous()  # rollout i of U sits at position i+sh
        s1, _ = GpuRolloutBackend(task, N).rollout_device(x0, Us)
        d = (s1[sh:] - s0[:-sh]).abs()
        print(f"  {task} shift {sh}: identical rollouts {int((d.amax(dim=(1, 2)) == 0).sum())}/{N - sh}, max |diff| {float(d.max()):.3e}, first differing step {int((d.amax(dim=(0, 2)) > 0).nonzero()[0]) if d.max() > 0 else -1}")

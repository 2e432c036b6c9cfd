"""What re-rolling the E <= 5 trace elites in materialise mode costs on the headline model (the alternative to writing every rollout's trace rows in the fused kernel,
SURVEY A16): jh_rollout_materialize of E rollouts x H = 64 on leap_cube, latency mode.  usage: python tools/diag/time_elite_reroll.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from judo_amd.rollout_backend import GpuRolloutBackend
from judo_amd.tasks import LeapCube
task = LeapCube(); H = 64
rec = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "ab_inputs_leap.npz"))
for E in (1, 5):
    be = GpuRolloutBackend("leap_cube", E)
    x0 = torch.as_tensor(np.asarray(task.default_state(), dtype=np.float32), device=be.model.device)
    # controls of a mid-run recorded plan (nominal knots of plan step 30 held over the horizon)
    U = torch.as_tensor(np.repeat(rec["knots"][30][:, None, :], H // rec["knots"].shape[1], axis=1).reshape(1, H, -1).repeat(E, axis=0).astype(np.float32), device=be.model.device)
    for _ in range(3): be.rollout_device(x0, U)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    torch.cuda.synchronize(); ev[0].record()
    for _ in range(20): be.rollout_device(x0, U)
    ev[1].record(); torch.cuda.synchronize()
    print(f"elite re-roll, leap_cube, E={E} rollouts x H={H} (materialise, states + sensors): {ev[0].elapsed_time(ev[1]) / 20:.3f} ms per launch")

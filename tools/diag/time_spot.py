"""Spot policy rollout throughput at the headline batch: physics kernel alone and with the policy step.  JUDO_AMD_LIB selects a variant build."""
import sys, time
import numpy as np, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from judo_amd import spot_tasks as ST
from judo_amd.policy import PolicyRolloutBackend

N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
T = int(sys.argv[2]) if len(sys.argv) > 2 else 10
be = PolicyRolloutBackend(N)
if os.environ.get("SELF") == "0":  # the ground contacts only (rounds 1-4), for A/B
    from judo_amd import _lib
    _lib.check(_lib.lib().jh_tree_set_self_collision(be.engine.handle, 0), "jh_tree_set_self_collision")
x0 = np.concatenate([[0, 0, ST.STANDING_HEIGHT, 1, 0, 0, 0], ST.LEGS_STANDING_POS_RL, ST.ARM_STOWED_POS, np.zeros(25)])
DEFAULT_POLICY_COMMAND = np.concatenate([[0, 0, 0], ST.ARM_STOWED_POS, np.zeros(12), [0, 0, ST.STANDING_HEIGHT]])
DEFAULT_JOINT_POS = np.array([0.12, 0.5, -1, -0.12, 0.5, -1, 0.12, 0.5, -1, -0.12, 0.5, -1, 0, -0.9, 1.8, 0, -0.9, 0, -1.54])
cm = torch.as_tensor(np.tile(DEFAULT_POLICY_COMMAND, (N, T, 1)), dtype=torch.float32, device="cuda")
cm[:, :, :3] = (torch.rand((N, 1, 3), device="cuda") - 0.5) * 1.0
xx = torch.as_tensor(np.tile(x0, (N, 1)), dtype=torch.float32, device="cuda"); xx[:, 7:19] += torch.randn((N, 12), device="cuda") * 0.05
lo = torch.zeros((N, 12), device="cuda")
s, _, _ = be.rollout(xx, cm, lo); torch.cuda.synchronize(); be.engine.stats()
t0 = time.perf_counter(); s, _, _ = be.rollout(xx, cm, lo); torch.cuda.synchronize(); dt = time.perf_counter() - t0
st = be.engine.stats()
print(f"{N} x {T} control steps: {dt*1e3:.1f} ms -> {N*T*2/dt/1e6:.2f} M physics steps/s, {N*T/dt/1e6:.2f} M control steps/s; newton/step {st['newton_iterations']/st['steps']:.2f}", st, "finite", bool(torch.isfinite(s).all()))
# physics alone, from the rolled-out states (walking gaits)
x = s[:, -1].contiguous(); ctrl = torch.as_tensor(np.tile(DEFAULT_JOINT_POS, (N, 1)), dtype=torch.float32, device="cuda"); warm = torch.zeros((N, 25), device="cuda")
be.engine.substeps(x, ctrl, warm, 2); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); y = be.engine.substeps(x, ctrl, warm, 20); e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1)
print(f"physics only: 20 substeps {ms:.2f} ms -> {N*20/ms/1e3:.2f} M steps/s", be.engine.stats())

"""A/B of Spot policy-rollout builds (JUDO_AMD_LIB selects the library) on seeded inputs: the shipped batch (24 rollouts x 100 control steps: latency) and the headline batch
(65 536 x 10: throughput); the rolled-out states are saved (OUT=path.npy) so that two builds can be compared bit for bit.  usage: python tools/diag/ab_spot.py"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from judo_amd import spot_tasks as ST
from judo_amd.policy import PolicyRolloutBackend

torch.manual_seed(7)
x0 = np.concatenate([[0, 0, ST.STANDING_HEIGHT, 1, 0, 0, 0], ST.LEGS_STANDING_POS_RL, ST.ARM_STOWED_POS, np.zeros(25)])
CMD = np.concatenate([[0, 0, 0], ST.ARM_STOWED_POS, np.zeros(12), [0, 0, ST.STANDING_HEIGHT]])
keep = []
for N, T, reps in ((24, 100, 5), (65536, 10, 3)):
    be = PolicyRolloutBackend(N)
    cm = torch.as_tensor(np.tile(CMD, (N, T, 1)), dtype=torch.float32, device="cuda")
    cm[:, :, :3] = (torch.rand((N, 1, 3), device="cuda") - 0.5) * 1.0
    xx = torch.as_tensor(np.tile(x0, (N, 1)), dtype=torch.float32, device="cuda"); xx[:, 7:19] += torch.randn((N, 12), device="cuda") * 0.05
    lo = torch.zeros((N, 12), device="cuda")
    s, _, _ = be.rollout(xx, cm, lo); torch.cuda.synchronize(); be.engine.stats()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); s, _, _ = be.rollout(xx, cm, lo); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    st = be.engine.stats()
    print(f"{N} x {T}: {min(ts) * 1e3:.2f} ms (min of {reps}; {min(ts) / T * 1e6:.1f} us per control step)  newton/step {st['newton_iterations'] / max(st['steps'], 1):.2f} dropped {st.get('contact_overflow', 0)}", flush=True)
    keep.append(s[:: max(1, N // 256)].cpu().numpy().ravel())
if os.environ.get("OUT"): np.save(os.environ["OUT"], np.concatenate(keep))

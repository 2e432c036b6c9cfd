"""How much of the Newton work of the leap kernel is lock-step waste (a wave runs the maximum iteration count of its four rollouts), and how much of it a
regrouping of rollouts by their recent iteration counts could recover.  Needs the -DJH_V2_ITERDUMP build (JUDO_AMD_LIB=build/libjudo_amd_iterdump.so)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from judo_amd.controller import make_controller
N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
c = make_controller("leap_cube", "mppi"); c.optimizer.config.num_rollouts = N; c.controller_cfg.horizon = 0.64
c.reset(); c.current_state = c.task.default_state(); c.system_metadata = {"goal_quat": np.array([0.0, 1.0, 0.0, 0.0])}; c.optimizer.seed(1234)
c.keep_candidates = True
t = 0.0
for _ in range(4):
    c.time = t; c.update_action(); t += 0.05
it = c.candidate_knots_device.reshape(64, N).cpu().numpy()          # (step, rollout)
print("mean iterations per rollout-step", it.mean(), " fraction of steps without constraint rows", (it == 0).mean())
def wave_cost(order): return it[:, order].reshape(64, N // 4, 4).max(2).sum()
ideal = it.sum() / 4
base = wave_cost(np.arange(N))
print(f"lock-step cost / ideal: as launched {base/ideal:.3f}")
for chunk in (64, 16, 8, 4, 2, 1):
    tot = 0.0
    order = np.arange(N)
    for s0 in range(0, 64, chunk):
        blk = it[s0:s0 + chunk][:, order]
        tot += blk.reshape(chunk, N // 4, 4).max(2).sum()
        order = np.argsort(it[s0:s0 + chunk].sum(0), kind="stable")   # regroup by the iterations of the chunk just finished
    print(f"  regroup every {chunk:2d} steps by the previous chunk's count: {tot/ideal:.3f}")
# oracle grouping (knows the future): upper bound of what any predictor can reach
tot = sum(np.sort(it[s])[::-1].reshape(N // 4, 4).max(1).sum() for s in range(64))
print(f"  perfect per-step grouping: {tot/ideal:.3f}")
ac = np.corrcoef(it[:-1].reshape(-1), it[1:].reshape(-1))[0, 1]
print("step-to-step autocorrelation of the iteration count", ac)

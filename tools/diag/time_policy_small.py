"""Latency of one policy step / one tree-kernel launch at a few dozen rollouts (the reference's shipped batch): where a control step's 230 us go."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from judo_amd import spot_tasks as ST
from judo_amd.policy import SpotLocomotionPolicy, SpotStateLayout, SpotTreeEngine

N = int(sys.argv[1]) if len(sys.argv) > 1 else 24
pol, eng, lay = SpotLocomotionPolicy(), SpotTreeEngine(), SpotStateLayout(26, 25)
x0 = np.concatenate([[0, 0, ST.STANDING_HEIGHT, 1, 0, 0, 0], ST.LEGS_STANDING_POS_RL, ST.ARM_STOWED_POS, np.zeros(25)])
x = torch.as_tensor(np.tile(x0, (N, 1)), dtype=torch.float32, device="cuda")
cmd = torch.as_tensor(np.tile(np.concatenate([[0, 0, 0], ST.ARM_STOWED_POS, np.zeros(12), [0, 0, 0.52]]), (N, 1)), dtype=torch.float32, device="cuda")
out = torch.zeros((N, 12), device="cuda"); warm = torch.zeros((N, 25), device="cuda")
ctrl, _ = pol.step(x, cmd, out, lay)
def timeit(f, reps=200):
    for _ in range(10): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e6
print(f"N={N}: policy step {timeit(lambda: pol.step(x, cmd, out, lay)):.1f} us, tree kernel (2 substeps) {timeit(lambda: eng.substeps(x, ctrl, warm, 2)):.1f} us, "
      f"empty torch kernel {timeit(lambda: out.add_(0.0)):.1f} us")

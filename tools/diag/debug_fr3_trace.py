"""Scratch diagnostic (GPU box, -DJH_V3_TRACE build): re-run the captured cap-hit rollout alone; the kernel prints its late Newton iterations."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from judo_amd.rollout_backend import GpuRolloutBackend
d = np.load("tools/diag/cap_state.npz")
be = GpuRolloutBackend("fr3_pick", 1)
gs, _, _ = be.rollout(d["x0"], d["U"])
print(be.model.stats())

import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from judo_amd.rollout_backend import GpuRolloutBackend
from judo_amd.tasks import FR3Pick
from oracle import oracle as O
t = FR3Pick(); x0 = t.default_state(); om = O.Model('fr3_pick')
for N, H in ((1, 18), (1, 20), (1, 22), (1, 40), (2, 40)):
    be = GpuRolloutBackend('fr3_pick', N)
    U = np.tile(t.reset_command, (N, H, 1))
    gs, gy, _ = be.rollout(x0, U); torch.cuda.synchronize()
    rs, ry = om.rollout(x0, U)
    print(N, H, be.model.stats(), 'max state err', np.abs(gs - rs).max(), 'fingers', gs[0, -1, 14:16], rs[0, -1, 14:16])

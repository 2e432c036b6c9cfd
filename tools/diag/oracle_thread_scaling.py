"""Scratch diagnostic (any box): throughput of the oracle's threaded rollout against the thread count (the bench's cpu_baseline uses all hardware threads)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import oracle as O
from judo_amd.tasks import LeapCube
task = LeapCube(); om = O.Model('leap_cube')
H, K = 64, 4
rng = np.random.default_rng(0)
print('cpu_count', os.cpu_count())
for nt in [int(a) for a in sys.argv[1:]] or [1, 16, 64, 128, 256]:
    N = max(nt * 4, 64)
    nom = np.tile(task.optimizer_warm_start(), (K, 1)); sig = 0.2 * np.arange(1, K + 1)[:, None]
    knots = np.clip(nom[None] + rng.standard_normal((N, K, 16)) * sig[None], task.actuator_ctrlrange[:, 0], task.actuator_ctrlrange[:, 1])
    W = O.spline_weights('cubic', np.linspace(0, 0.64, K), np.arange(H) * 0.01)
    ctrl = np.einsum('hk,nku->nhu', W, knots)
    t = time.perf_counter(); om.rollout(np.asarray(task.default_state(), float), ctrl, nthread=nt); dt = time.perf_counter() - t
    print(f'threads {nt:4d}: {N / dt:9.1f} rollouts/s  {N * H / dt / nt:9.1f} steps/s/thread', flush=True)

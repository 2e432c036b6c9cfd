"""GPU box: per-state errors of the tree kernel against the oracle on robot self-collision states (the groups of tests/test_gpu_spot.py), without asserting."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import policy as P
from judo_amd.policy import SpotTreeEngine
import importlib.util
spec = importlib.util.spec_from_file_location("t", os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests", "test_gpu_spot.py")); T = importlib.util.module_from_spec(spec); spec.loader.exec_module(T)
om = P.spot_model(self_collision=True); eng = SpotTreeEngine(); eng0 = SpotTreeEngine(self_collision=False)
groups = T._self_collision_states(P, om, 6, seed=5)
np.set_printoptions(precision=3, linewidth=200, suppress=False)
for kind, xs in groups.items():
    X = np.stack(xs); U = X[:, 7:26].copy()
    xt = torch.as_tensor(X, dtype=torch.float32, device="cuda"); ut = torch.as_tensor(U, dtype=torch.float32, device="cuda")
    for k in (1, 3):
        eng.stats()
        got = eng.substeps(xt, ut, torch.zeros((len(X), 25), dtype=torch.float32, device="cuda"), k).cpu().numpy()
        got0 = eng0.substeps(xt, ut, torch.zeros((len(X), 25), dtype=torch.float32, device="cuda"), k).cpu().numpy()
        ref = T._oracle_steps(om, X, U, k)
        e = np.abs(got - ref); e0 = np.abs(got0 - ref)
        print(f"{kind:7s} k={k}: qd err max per state {e[:, 32:].max(1)}  (without self-collision {e0[:, 32:].max(1)})  base v {e[:, 26:32].max():.2e} pos {e[:, :26].max():.2e}  stats {eng.stats()}")
    for i, x in enumerate(X[:2]):
        f = om.forward(x[:26], x[26:], x[7:26])
        print("   state", i, "ncon", f["ncon"], "iters", f["solver_iter"], "pairs", [(int(c[13]), int(c[14]), round(c[0], 4)) for c in f["contacts"]])

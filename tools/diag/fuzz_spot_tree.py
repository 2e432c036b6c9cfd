"""Random-state sweep of the Spot tree kernel against the oracle (one and three steps each): worst errors and where they occur."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import policy as P
from judo_amd.policy import SpotTreeEngine

N = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
om, eng = P.spot_model(self_collision=True), SpotTreeEngine()  # both with the robot's own contact pairs (round 5)
lo = np.array([j["range"][0] if j["range"] else -3 for j in eng.desc["joints"] if j["type"] != "free"])
hi = np.array([j["range"][1] if j["range"] else 3 for j in eng.desc["joints"] if j["type"] != "free"])
X = np.tile(P.spot_reset_state(), (N, 1))
X[:, 2] = rng.uniform(float(os.environ.get("ZMIN", "0.15")), 0.9, N)
q = rng.standard_normal((N, 4)); q[:, 0] += rng.uniform(0, 6, N); X[:, 3:7] = q / np.linalg.norm(q, axis=1, keepdims=True)
X[:, 7:26] = rng.uniform(lo - 0.15, hi + 0.15, (N, 19))          # some beyond their limits
X[:, 26:29] = rng.standard_normal((N, 3)) * 1.5; X[:, 29:32] = rng.standard_normal((N, 3)) * 3; X[:, 32:] = rng.standard_normal((N, 19)) * 4
U = rng.uniform(lo - 0.5, hi + 0.5, (N, 19))
# classify the states by what the oracle finds in them: inside the kernel's capacities (32 contacts, 8 of them between two chains) and not tangled (no robot-robot penetration beyond 5 cm)
from judo_amd.tree_model import tree_structure
_st = tree_structure(eng.desc); _gs = eng.desc["geoms"]
def _chain(g):
    b = _gs[g]["body"]
    if b not in _st["body_of"]: return 0
    c0 = _st["info"][_st["body_of"][b]]["start"]; return 1 + (c0 // 3 if c0 < 12 else 4)
ok = np.zeros(N, dtype=bool)
for i, x in enumerate(X):
    f = om.forward(x[:26], x[26:], U[i])
    cross = sum(1 for c in f["contacts"] if 27 not in (int(c[13]), int(c[14])) and _chain(int(c[13])) > 0 and _chain(int(c[14])) > 0 and _chain(int(c[13])) != _chain(int(c[14])))
    deep = min([c[0] for c in f["contacts"] if 27 not in (int(c[13]), int(c[14]))] + [0.0])
    ok[i] = f["ncon"] <= 28 and cross <= 6 and deep > -0.05
print(f"{ok.sum()} of {N} states inside the capacities with margin (<= 28 contacts, <= 6 between two chains) and not tangled (robot-robot penetration < 5 cm)")
xs, us = torch.as_tensor(X, dtype=torch.float32, device="cuda"), torch.as_tensor(U, dtype=torch.float32, device="cuda")
for k in (1, 3):
    sens = torch.zeros((N, 48), device="cuda")
    got = eng.substeps(xs, us, torch.zeros((N, 25), device="cuda"), k, sensors=sens).cpu().numpy()
    ref, sref = np.zeros_like(X), np.zeros((N, 48))
    st, se = om.rollout(X, np.repeat(U[:, None], k, axis=1), nthread=os.cpu_count())
    ref, sref = st[:, -1], se[:, -1]
    err = np.abs(got - ref); scale = 1 + np.abs(ref)
    rel = (err / scale)
    worst = np.argsort(rel.max(1))[::-1][:5]
    for name, m in (("inside", ok), ("outside", ~ok)):
        print(f"  steps {k}, {name} ({m.sum()}): vel err median {np.median(err[m][:, 26:].max(1)):.2e} p90 {np.percentile(err[m][:, 26:].max(1), 90):.2e} p99 {np.percentile(err[m][:, 26:].max(1), 99):.2e} max {err[m][:, 26:].max():.2e}; "
              f"q err p99 {np.percentile(err[m][:, 7:26].max(1), 99):.2e} max {err[m][:, 7:26].max():.2e}; non-finite {int((~np.isfinite(got[m])).any(1).sum())}", flush=True)
    print(f"steps {k}: max abs err pos {err[:, :7].max():.2e} q {err[:, 7:26].max():.2e} vel {err[:, 26:].max():.2e}; sensors {np.abs(sens.cpu().numpy() - sref).max():.2e}; "
          f"p99 rel {np.percentile(rel.max(1), 99):.2e} max rel {rel.max():.2e}", eng.stats(), flush=True)
    for w in worst:
        j = int(rel[w].argmax()); print(f"   rollout {w}: component {j} got {got[w, j]:.6f} ref {ref[w, j]:.6f}  base z {X[w, 2]:.2f}")

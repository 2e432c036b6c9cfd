import sys
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from judo_amd.rollout_backend import GpuRolloutBackend
import test_gpu_leap_self as T
N = 1200
rng = np.random.default_rng(123)
om, xs, q = T._tangled_states(N, seed=99, frac=0.5)
home = xs[0, :3].copy(); home[2] -= 0.3
xs[:, :3] = home + rng.uniform(-0.03, 0.03, (N, 3))
quat = rng.standard_normal((N, 4)); xs[:, 3:7] = quat / np.linalg.norm(quat, axis=1, keepdims=True)
xs[:, 23:29] = rng.standard_normal((N, 6)) * np.array([0.2, 0.2, 0.2, 2, 2, 2])
U = q[:, None, :]
ref, _ = om.rollout(xs, U)
g, _, _ = GpuRolloutBackend("leap_cube", N).rollout(xs, U)
scale = np.maximum(1.0, np.abs(ref[:, 0, 23:]).max(axis=1, keepdims=True))
ev = (np.abs(g[:, 0] - ref[:, 0])[:, 23:] / scale).max(1)
d = om.desc; body = [gg["body"] for gg in d["geoms"]]; names = [b["name"] for b in d["bodies"]]
chain = {b: ({"if": 0, "mf": 1, "rf": 2, "th": 3}.get(n[:2], -1) if n != "cube" else -2) for b, n in enumerate(names)}
worst = [i for i in np.argsort(-ev) if T._contact_kinds(om, xs[i], q[i])[:3] and sum(T._contact_kinds(om, xs[i], q[i])[:3]) <= 32][:4]
for i in worst:
    k = T._contact_kinds(om, xs[i], q[i]); f = om.forward(xs[i, :23], xs[i, 23:], q[i])
    pairs = set()
    for row in f["contacts"]:
        ca, cb = chain[body[int(row[13])]], chain[body[int(row[14])]]
        if ca >= 0 and cb >= 0 and ca != cb: pairs.add((min(ca, cb), max(ca, cb)))
    be1 = GpuRolloutBackend("leap_cube", 1); be1.model.stats()
    g1, _, _ = be1.rollout(xs[i:i + 1], U[i:i + 1]); st = be1.model.stats()
    e1 = (np.abs(g1[0, 0] - ref[i, 0])[23:] / scale[i]).max()
    print(f"state {i}: error {ev[i]:.2e} (alone {e1:.2e}); oracle contacts {f['ncon']} kinds {k}, coupled pairs {sorted(pairs)}, oracle iterations {f['solver_iter']}; kernel alone: overflow {st['contact_overflow']}, iterations {st['newton_iters']}, cap hits {st['newton_cap_hits']}")
    idx = np.argsort(-np.abs(g1[0, 0] - ref[i, 0])[23:])[:4]
    print("    worst dofs", idx, "gpu", g1[0, 0, 23 + idx], "oracle", ref[i, 0, 23 + idx])

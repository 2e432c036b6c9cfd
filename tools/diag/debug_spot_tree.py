"""Spot tree kernel against the oracle, step by step: prints per-block errors (base pos / quat / joints / velocities)."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from oracle import policy as P
from judo_amd.policy import SpotTreeEngine, PolicyRolloutBackend

om = P.spot_model()
eng = SpotTreeEngine()
rng = np.random.default_rng(0)
x0 = P.spot_reset_state()

def blocks(a, b):
    d = np.abs(a - b)
    return dict(pos=d[..., 0:3].max(), quat=d[..., 3:7].max(), q=d[..., 7:26].max(), vlin=d[..., 26:29].max(), vang=d[..., 29:32].max(), qd=d[..., 32:].max())

def run_case(name, X, U, nsteps):
    N = X.shape[0]
    xs = torch.as_tensor(X, dtype=torch.float32, device="cuda").contiguous()
    us = torch.as_tensor(U, dtype=torch.float32, device="cuda").contiguous()
    for k in range(1, nsteps + 1):
        warm = torch.zeros((N, 25), dtype=torch.float32, device="cuda")
        got = eng.substeps(xs, us, warm, k).cpu().numpy()
        ref = np.stack([om.rollout(X[i], np.repeat(U[i][None], k, axis=0)[None], nthread=1)[0][0, -1] for i in range(N)])
        print(name, "steps", k, {a: f"{b:.2e}" for a, b in blocks(got, ref).items()}, eng.stats(), flush=True)
        if not np.isfinite(got).all(): print("  non-finite output"); break

# 1. in the air (no contacts): base lifted
X = np.tile(x0, (4, 1)); X[:, 2] = 1.0
X[:, 7:26] += rng.standard_normal((4, 19)) * 0.1; X[:, 26:] = rng.standard_normal((4, 25)) * 0.3
U = np.tile(P.DEFAULT_JOINT_POS, (4, 1))
run_case("air", X, U, 3)
# 2. standing
X = np.tile(x0, (4, 1)); X[:, 7:19] += rng.standard_normal((4, 12)) * 0.02
run_case("stand", X, U, 4)
# 3. dropped / tilted
X = np.tile(x0, (8, 1)); X[:, 2] = 0.45; q = rng.standard_normal((8, 4)) * 0.15; q[:, 0] = 1; X[:, 3:7] = q / np.linalg.norm(q, axis=1, keepdims=True)
X[:, 26:] = rng.standard_normal((8, 25)) * 0.5
U = np.tile(P.DEFAULT_JOINT_POS, (8, 1)) + rng.standard_normal((8, 19)) * 0.2
run_case("tilt", X, U, 6)
# 4. full policy rollout
Ws, bs = P.load_actor()
N, T = 4, 50
cmds = np.tile(P.DEFAULT_POLICY_COMMAND, (N, T, 1)); cmds[1, :, 0] = 0.5; cmds[2, :, 1] = 0.3; cmds[3, :, 2] = 0.5
be = PolicyRolloutBackend(N, carry_warmstart=False)
st, _, po = be.rollout(x0, cmds, np.zeros((N, 12)))
for i in range(N):
    ref, o = P.policy_rollout(om, Ws, bs, x0, cmds[i])
    for t in (0, 4, 9, 24, 49):
        print("rollout", i, "t", t, {a: f"{b:.2e}" for a, b in blocks(st[i, t], ref[t]).items()})
    print("  policy out err", np.abs(po[i] - o).max(), "base", st[i, -1, :3], ref[-1, :3])
print(be.engine.stats())
# 5. timing
N = 65536
be = PolicyRolloutBackend(N)
cm = torch.as_tensor(np.tile(P.DEFAULT_POLICY_COMMAND, (N, 8, 1)), dtype=torch.float32, device="cuda")
xx = torch.as_tensor(np.tile(x0, (N, 1)), dtype=torch.float32, device="cuda"); xx[:, 7:19] += torch.randn((N, 12), device="cuda") * 0.05
lo = torch.zeros((N, 12), device="cuda")
be.rollout(xx, cm, lo); torch.cuda.synchronize()
t0 = time.perf_counter(); s, _, _ = be.rollout(xx, cm, lo); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"65536 x 8 control steps: {dt*1e3:.1f} ms -> {N*8*2/dt/1e6:.2f} M physics steps/s", be.engine.stats(), "finite", bool(torch.isfinite(s).all()))

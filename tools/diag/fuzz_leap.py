"""Random-state sweep of the leap kernels against the fp64 oracle: tangled hand configurations WITH the cube inside the hand (cube contacts, hand self-contacts and both at once),
random velocities; one and three physics steps.  Reports the error distribution by solver path and the worst cases."""
import sys
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from judo_amd.rollout_backend import GpuRolloutBackend
from oracle import oracle as O
import test_gpu_leap_self as T

task = sys.argv[1] if len(sys.argv) > 1 else "leap_cube"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
rng = np.random.default_rng(123)
om, xs, q = T._tangled_states(N, seed=99, frac=0.5, task=task)
home = xs[0, :3].copy(); home[2] -= 0.3
xs[:, :3] = home + rng.uniform(-0.03, 0.03, (N, 3))          # cube back into the hand, +-3 cm
quat = rng.standard_normal((N, 4)); xs[:, 3:7] = quat / np.linalg.norm(quat, axis=1, keepdims=True)
xs[:, 23:29] = rng.standard_normal((N, 6)) * np.array([0.2, 0.2, 0.2, 2, 2, 2])
kinds = np.array([T._contact_kinds(om, xs[i], q[i]) for i in range(N)])
ncon = kinds[:, :3].sum(1)
ok = ncon <= 48  # the leap kernel's contact pool (jh_model_limits out[3])
for H in (1, 3):
    U = np.repeat(q[:, None, :], H, axis=1)
    ref, _ = om.rollout(xs, U)
    be = GpuRolloutBackend(task, N); g, _, _ = be.rollout(xs, U)
    assert np.isfinite(g).all()
    scale = np.maximum(1.0, np.abs(ref[:, -1, 23:]).max(axis=1, keepdims=True))
    e = np.abs(g[:, -1] - ref[:, -1]); ev = (e[:, 23:] / scale).max(1); ep = e[:, :23].max(1)
    print(f"{task} H={H}: {N} states, {int(ok.sum())} within the 48-contact pool; contacts per state mean {ncon.mean():.1f} max {ncon.max()}")
    for name, sel in (("no contact", ok & (ncon == 0)), ("cube only", ok & (kinds[:, 0] > 0) & (kinds[:, 1] + kinds[:, 2] == 0)), ("hand only", ok & (kinds[:, 0] == 0) & (kinds[:, 1] + kinds[:, 2] > 0)),
                      ("cube + hand, arrow", ok & (kinds[:, 0] > 0) & (kinds[:, 1] > 0) & (kinds[:, 2] == 0)), ("cube + coupled chains (staged)", ok & (kinds[:, 0] > 0) & (kinds[:, 2] > 0) & (kinds[:, 3] < 2)),
                      ("cycle (dense)", ok & (kinds[:, 3] == 2)), ("pool overflow", ~ok)):
        if sel.sum():
            print(f"   {name:32s} n={int(sel.sum()):5d}  velocity error / scale: median {np.median(ev[sel]):.1e} p95 {np.percentile(ev[sel], 95):.1e} p99 {np.percentile(ev[sel], 99):.1e} max {ev[sel].max():.1e} | position max {ep[sel].max():.1e}")
st = be.model.stats()
print("kernel counters of the last batch:", st)
# worst single-step cases of the pool-respecting states: contact count, deepest penetration, oracle Newton iterations
if "--worst" in sys.argv:
    U = q[:, None, :]
    ref, _ = om.rollout(xs, U); g, _, _ = GpuRolloutBackend(task, N).rollout(xs, U)
    scale = np.maximum(1.0, np.abs(ref[:, -1, 23:]).max(axis=1, keepdims=True))
    ev = (np.abs(g[:, -1] - ref[:, -1])[:, 23:] / scale).max(1); ev[~ok] = 0
    for i in np.argsort(-ev)[:12]:
        f = om.forward(xs[i, :23], xs[i, 23:], q[i])
        depth = min((row[6] for row in f["contacts"]), default=0.0)
        print(f"state {i}: error {ev[i]:.2e}, contacts {f['ncon']} (cube {kinds[i, 0]}, hand {kinds[i, 1]}, cross {kinds[i, 2]}), deepest penetration {depth * 1e3:.1f} mm, oracle iterations {f['solver_iter']}, velocity scale {scale[i, 0]:.1f}")
    depth_all = np.array([min((row[6] for row in om.forward(xs[i, :23], xs[i, 23:], q[i])["contacts"]), default=0.0) for i in range(0, N, 5)])
    evs = ev[::5]; oks = ok[::5]
    for lo_, hi_ in ((0, 1), (1, 3), (3, 6), (6, 12), (12, 100)):
        sel = oks & (-depth_all * 1e3 >= lo_) & (-depth_all * 1e3 < hi_)
        if sel.sum():
            print(f"deepest penetration {lo_}-{hi_} mm: n={int(sel.sum())} median {np.median(evs[sel]):.1e} p95 {np.percentile(evs[sel], 95):.1e} max {evs[sel].max():.1e}")

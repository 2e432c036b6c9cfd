"""Random-state sweep of the fr3_pick kernel against the fp64 oracle: arm configurations around the grasp pose (fingers around / inside / on the cube, pads on the table),
random gripper openings, cube poses and velocities; one and three physics steps."""
import sys
import numpy as np
sys.path.insert(0, ".")
from judo_amd.rollout_backend import GpuRolloutBackend
from judo_amd.tasks import FR3Pick
from oracle import oracle as O
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
rng = np.random.default_rng(5)
om = O.Model("fr3_pick"); t = FR3Pick()
x0 = t.default_state()
xs = np.tile(x0, (N, 1))
grasp = np.array([0.0, 0.55, 0.0, -2.05, 0.0, 2.6, 0.785])
xs[:, 7:14] = grasp + rng.uniform(-0.2, 0.2, (N, 7)) * rng.uniform(0, 1, (N, 1))
xs[:, 14:16] = rng.uniform(0.0, 0.04, (N, 2))
xs[:, 0:2] = x0[0:2] + rng.uniform(-0.03, 0.03, (N, 2))
xs[:, 2] = x0[2] + rng.uniform(-0.003, 0.03, N)
yaw = rng.uniform(-np.pi, np.pi, N); tilt = rng.uniform(0, 0.3, N) * (rng.uniform(0, 1, N) < 0.3)
xs[:, 3] = np.cos(yaw / 2) * np.cos(tilt / 2); xs[:, 4] = np.sin(tilt / 2) * np.cos(yaw / 2); xs[:, 5] = np.sin(tilt / 2) * np.sin(yaw / 2); xs[:, 6] = np.sin(yaw / 2) * np.cos(tilt / 2)
xs[:, 16:] = 0.3 * rng.standard_normal((N, 15))
lo, hi = t.actuator_ctrlrange[:, 0], t.actuator_ctrlrange[:, 1]
u = np.clip(np.concatenate([xs[:, 7:14] + rng.uniform(-0.1, 0.1, (N, 7)), rng.uniform(lo[-1], hi[-1], (N, 1))], axis=1), lo, hi)
d = om.desc; body = [g["body"] for g in d["geoms"]]; names = [b["name"] for b in d["bodies"]]
lf, rf = names.index("left_finger"), names.index("right_finger")
ncon = np.zeros(N, int); nff = np.zeros(N, int)
for i in range(N):
    f = om.forward(xs[i, :16], xs[i, 16:], u[i]); ncon[i] = f["ncon"]
    nff[i] = sum(1 for row in f["contacts"] if {body[int(row[13])], body[int(row[14])]} == {lf, rf})
ok = (ncon - nff <= 32) & (nff <= 48)
print(f"fr3_pick: {N} states, contacts mean {ncon.mean():.1f} max {ncon.max()} (finger-finger mean {nff.mean():.1f} max {nff.max()}), {int(ok.sum())} within the kernel's capacities")
for H in (1, 3):
    U = np.repeat(u[:, None, :], H, axis=1)
    ref, rsens = om.rollout(xs, U)
    be = GpuRolloutBackend("fr3_pick", N); be.model.stats(); g, gsens, _ = be.rollout(xs, U)
    assert np.isfinite(g).all()
    scale = np.maximum(1.0, np.abs(ref[:, -1, 16:]).max(axis=1, keepdims=True))
    ev = (np.abs(g[:, -1] - ref[:, -1])[:, 16:] / scale).max(1); ep = np.abs(g[:, -1] - ref[:, -1])[:, :16].max(1); es = np.abs(gsens[:, 0] - rsens[:, 0]).max(1)
    for name, sel in (("no contact", ok & (ncon == 0)), ("1-8 contacts", ok & (ncon > 0) & (ncon <= 8)), ("9-32 contacts", ok & (ncon > 8) & (ncon <= 32)), ("> 32 contacts", ok & (ncon > 32)), ("over capacity", ~ok)):
        if sel.sum():
            print(f"  H={H} {name:16s} n={int(sel.sum()):5d} velocity error / scale: median {np.median(ev[sel]):.1e} p95 {np.percentile(ev[sel], 95):.1e} p99 {np.percentile(ev[sel], 99):.1e} max {ev[sel].max():.1e} | position max {ep[sel].max():.1e} | sensors (first step) max {es[sel].max():.1e}")
    print("  kernel counters:", be.model.stats())

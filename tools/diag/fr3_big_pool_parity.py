"""States of the shipped fr3_pick workload with more general contacts than the LDS pool holds (33..64): single physics steps of the product kernel against the oracle."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from judo_amd.controller import make_controller
from judo_amd.rollout_backend import GpuRolloutBackend
from oracle import oracle as O

def workload_states(plan_steps=60, first=30):
    c = make_controller("fr3_pick", "cem"); c.solver_warnings = False
    c.reset(); c.current_state = c.task.default_state(); c.system_metadata = {}; c.optimizer.seed(3); c.force_materialize = True
    out, t = [], 0.0
    for step in range(plan_steps):
        c.time = t; c.update_action(); t += 0.05
        if step >= first and step % 5 == 0:
            torch.cuda.synchronize()
            st = c.last_rollout[0].cpu().numpy().astype(np.float64)
            out.append(st[::4, ::5].reshape(-1, st.shape[-1]))
    return np.concatenate(out)

def contact_split(om, xs):
    gbody = [g["body"] for g in om.desc["geoms"]]; fing = {i for i, b in enumerate(om.desc["bodies"]) if "finger" in b["name"]}
    gen, ff = np.zeros(len(xs), int), np.zeros(len(xs), int)
    for i, x in enumerate(xs):
        for r in om.forward(x[:16], x[16:], x[7:15])["contacts"]:
            if gbody[int(r[13])] in fing and gbody[int(r[14])] in fing: ff[i] += 1
            else: gen[i] += 1
    return gen, ff

if __name__ == "__main__":
    om = O.Model("fr3_pick")
    xs = workload_states()
    gen, ff = contact_split(om, xs)
    print(f"{len(xs)} states; general contacts: {np.bincount(np.minimum(gen // 8, 12))} (bins of 8), pad-pad above 96: {(ff > 96).sum()}")
    sel = (gen > 32) & (gen <= 64) & (ff <= 96)
    x = xs[sel]; U = x[:, None, 7:15]
    ref, _ = om.rollout(x, U)
    be = GpuRolloutBackend("fr3_pick", len(x)); be.model.stats()
    g, _, _ = be.rollout(x, U)
    sc = np.maximum(1.0, np.abs(ref[:, 0, 16:]).max(axis=1, keepdims=True))
    e = (np.abs(g[:, 0, 16:] - ref[:, 0, 16:]) / sc).max(axis=1)
    print(f"{sel.sum()} states with 33..64 general contacts: velocity error / scale median {np.median(e):.1e} p90 {np.percentile(e, 90):.1e} p99 {np.percentile(e, 99):.1e} max {e.max():.1e}; stats {be.model.stats()}")
    sel2 = (gen <= 32) & (gen > 8) & (ff <= 96)
    x = xs[sel2][:400]; U = x[:, None, 7:15]
    ref, _ = om.rollout(x, U); g, _, _ = GpuRolloutBackend("fr3_pick", len(x)).rollout(x, U)
    sc = np.maximum(1.0, np.abs(ref[:, 0, 16:]).max(axis=1, keepdims=True)); e = (np.abs(g[:, 0, 16:] - ref[:, 0, 16:]) / sc).max(axis=1)
    print(f"{len(x)} states with 9..32 general contacts (LDS pool only): median {np.median(e):.1e} p90 {np.percentile(e, 90):.1e} p99 {np.percentile(e, 99):.1e} max {e.max():.1e}")

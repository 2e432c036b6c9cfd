"""Which contacts fill the leap kernel's 32-entry pool on the REAL workload?  Replays recorded headline plan steps with 2048 rollouts in materialise mode (GPU), then
counts the oracle's contacts (CPU) in every visited state: histogram of contacts per state, and for the states above 32 the composition by geom-pair type and body pair."""
import collections, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from judo_amd.controller import make_controller
from oracle import oracle as O
d = np.load("tools/diag/ab_inputs_leap.npz")
N = int(os.environ.get("N", "2048"))
c = make_controller("leap_cube", "mppi"); c.optimizer.config.num_rollouts = N; c.controller_cfg.horizon = 0.64
c.reset(); c.current_state = c.task.default_state(); c.system_metadata = {"goal_quat": np.array([0.0, 1.0, 0.0, 0.0])}
c.force_materialize = True
om = O.Model("leap_cube"); desc = om.desc
gtype = [g["type"] for g in desc["geoms"]]; gbody = [g["body"] for g in desc["geoms"]]; bname = [b["name"] for b in desc["bodies"]]
hist = collections.Counter(); comp = collections.Counter(); bodyp = collections.Counter(); over = 0; total = 0; perpair_max = collections.Counter()
for i in (int(a) for a in os.environ.get("STEPS", "5,20,35").split(",")):
    c.optimizer.seed(1000 + i); c.nominal_knots = d["knots"][i].copy(); c.times = d["times"][i].copy(); c.update_spline(c.times, c.nominal_knots); c.time = float(d["t"][i])
    c.update_action(); torch.cuda.synchronize()
    states, sensors, controls = (t.cpu().numpy().astype(np.float64) for t in c.last_rollout)
    x0 = np.asarray(c.current_state, dtype=np.float64)
    for n in range(states.shape[0]):
        for h in range(states.shape[1]):
            x = x0 if h == 0 else states[n, h - 1]
            f = om.forward(x[:23], x[23:], controls[n, h]); k = int(f["ncon"]); hist[min(k, 64)] += 1; total += 1
            if k > 32:
                over += 1; pp = collections.Counter()
                for row in f["contacts"]:
                    ga, gb = int(row[13]), int(row[14]); comp[tuple(sorted((gtype[ga], gtype[gb])))] += 1
                    key = tuple(sorted((bname[gbody[ga]], bname[gbody[gb]]))); bodyp[key] += 1; pp[(ga, gb)] += 1
                for key, v in pp.items(): perpair_max[v] += 1
print(f"{total} states, {over} above 32 contacts ({over / total:.2e})")
cum = 0
for k in sorted(hist): cum += hist[k]; print(f"  ncon {k:2d}{'+' if k == 64 else ' '}: {hist[k]:7d}  cumulative {cum / total:.6f}")
print("above 32: contacts by geom-type pair:", dict(comp))
print("above 32: contacts per geom pair (how many pairs produced k points):", dict(sorted(perpair_max.items())))
print("above 32: top body pairs:", bodyp.most_common(12))

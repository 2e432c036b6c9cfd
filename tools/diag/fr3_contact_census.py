"""Contacts of fr3_pick at the SHIPPED configuration (64 rollouts, 1 s horizon = 250 steps): the oracle's contact count per visited state, split into pad-against-pad contacts
(the kernel's 96 dedicated slots) and all others (its general pool), and the composition of the states above the general pool."""
import collections, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from judo_amd.controller import make_controller
from oracle import oracle as O
opt = sys.argv[1] if len(sys.argv) > 1 else "cem"
c = make_controller("fr3_pick", opt); c.solver_warnings = False
c.reset(); c.current_state = c.task.default_state(); c.system_metadata = {}; c.optimizer.seed(3)
c.force_materialize = True
om = O.Model("fr3_pick"); desc = om.desc
gbody = [g["body"] for g in desc["geoms"]]; bname = [b["name"] for b in desc["bodies"]]; gtype = [g["type"] for g in desc["geoms"]]
fingers = {i for i, n in enumerate(bname) if "finger" in n}
hist_gen, hist_ff = collections.Counter(), collections.Counter(); comp = collections.Counter(); total = 0; t = 0.0
for step in range(int(os.environ.get("STEPS", "40"))):
    c.time = t; c.update_action(); t += 0.05
    if step % 8 != 7: continue
    torch.cuda.synchronize()
    states, sensors, controls = (x.cpu().numpy().astype(np.float64) for x in c.last_rollout)
    x0 = np.asarray(c.current_state, dtype=np.float64)
    for n in range(0, states.shape[0], 4):
        for h in range(0, states.shape[1], 5):
            x = x0 if h == 0 else states[n, h - 1]
            f = om.forward(x[:om.nq], x[om.nq:], controls[n, h]); total += 1
            ff = gen = 0; pp = collections.Counter()
            for row in f["contacts"]:
                ba, bb = gbody[int(row[13])], gbody[int(row[14])]
                if ba in fingers and bb in fingers: ff += 1
                else: gen += 1; pp[tuple(sorted((bname[ba], bname[bb])))] += 1
            hist_gen[min(gen // 8 * 8, 96)] += 1; hist_ff[min(ff // 16 * 16, 112)] += 1
            if gen > 32: comp.update(pp)
print(f"{opt}: {total} states sampled")
print("general contacts per state (bins of 8):", dict(sorted(hist_gen.items())))
print("pad-against-pad contacts per state (bins of 16):", dict(sorted(hist_ff.items())))
print("states above 32 general contacts: contacts by body pair:", comp.most_common(10))

"""Where the update's tail spends its time inside the one-launch plan step of a closed-form model (a -DJH_TAIL_TICKS build of jh_simple.hip: 100 MHz wall-clock stamps of the
last workgroup).  usage: JUDO_AMD_LIB=variants/libjudo_amd_tticks.so python tools/diag/tail_ticks.py [task] [N]"""
import ctypes as C, sys
import numpy as np, torch
sys.path.insert(0, ".")
from judo_amd import _lib
from judo_amd.controller import make_controller
task = sys.argv[1] if len(sys.argv) > 1 else "cartpole"; N = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
c = make_controller(task, "mppi"); c.optimizer.config.num_rollouts = N; c.controller_cfg.horizon = 64 * c.task.dt
c.reset(); c.current_state = c.task.default_state()
f = _lib.lib().jh_debug_tail_ticks; f.restype = C.c_int
t = 0.0; acc = []
for i in range(300):
    c.time = t; c.update_action(); _ = c.traces; t += 0.05
    if i >= 100:
        torch.cuda.synchronize(); out = (C.c_longlong * 16)(); f(out); v = np.array(out[:8], dtype=np.float64); acc.append(np.diff(v) * 10.0)  # ns
a = np.median(np.array(acc), axis=0) / 1e3
names = ["block stage: update", "block stage: trace elites", "fence + ticket", "merge: nominal", "merge: trace elites choose", "trace rows", "system fence + flag"]
print(f"{task} N={N}: tail of the LAST workgroup, median over 200 plan steps, us: " + "; ".join(f"{n} {x:.2f}" for n, x in zip(names, a)) + f"; total {a.sum():.2f}")

"""Census of the leap kernel on the recorded headline inputs (a -DJH_V5_CENSUS build selected with JUDO_AMD_LIB): contacts per rollout-step and per wave-step (maximum over the
wave's four rollouts), Newton iterations per rollout-step / wave-step, line-search evaluations per Newton iteration of a rollout / of the wave, active rollouts per wave iteration.
usage: JUDO_AMD_LIB=variants/libjudo_amd_census.so python tools/diag/census_v5.py [plan steps, default 2,12,22,32,38]"""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from judo_amd.controller import make_controller
from judo_amd import _lib
d = np.load("tools/diag/ab_inputs_leap.npz")
c = make_controller("leap_cube", "mppi"); c.optimizer.config.num_rollouts = 65536; c.controller_cfg.horizon = 0.64
c.reset(); c.current_state = c.task.default_state(); c.system_metadata = {"goal_quat": np.array([0.0, 1.0, 0.0, 0.0])}
L = _lib.lib(); L.jh_model_counters.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.c_int, C.c_int]
steps = [int(a) for a in (sys.argv[1] if len(sys.argv) > 1 else "2,12,22,32,38").split(",")]
tot = np.zeros(512, np.int64)
for i in steps:
    c.model.stats(reset=True)
    c.optimizer.seed(1000 + i); c.nominal_knots = d["knots"][i].copy(); c.times = d["times"][i].copy(); c.update_spline(c.times, c.nominal_knots); c.time = float(d["t"][i])
    c.update_action(); torch.cuda.synchronize()
    out = (C.c_int * 512)(); assert L.jh_model_counters(c.model.handle, out, 0, 512) == 0
    a = np.array(list(out), np.int64); tot += a
    def line(name, h):
        n = h.sum(); cum = np.cumsum(h) / max(n, 1); mean = (h * np.arange(len(h))).sum() / max(n, 1)
        return f"  {name:42s} mean {mean:6.2f}  " + " ".join(f"{k}:{100 * v / max(n, 1):.1f}" for k, v in enumerate(h) if v > 0.002 * n) + f"   | <=16: {cum[min(16, len(h) - 1)]:.4f} <=32: {cum[min(32, len(h) - 1)]:.4f}"
    print(f"plan step {i}")
    print(line("contacts / rollout-step", a[64:128])); print(line("max contacts of the wave / wave-step", a[128:192]))
    print(line("Newton iterations / rollout-step", a[192:224])); print(line("Newton iterations / wave-step", a[224:256]))
    print(line("line-search evals / rollout iteration", a[256:288])); print(line("line-search evals / wave iteration", a[288:320]))
    print(line("active rollouts / wave iteration", a[320:328]))
print("ALL", steps)
a = tot
for name, lo, hi in (("contacts / rollout-step", 64, 128), ("max contacts of the wave / wave-step", 128, 192), ("Newton iterations / rollout-step", 192, 224), ("Newton iterations / wave-step", 224, 256),
                     ("line-search evals / rollout iteration", 256, 288), ("line-search evals / wave iteration", 288, 320), ("active rollouts / wave iteration", 320, 328),
                     ("quad-per-contact passes / rollout-step", 328, 344), ("quad-per-contact passes / wave-step", 344, 360), ("contacts on the busiest chain / rollout-step", 360, 372), ("... / wave-step", 372, 384)):
    h = a[lo:hi]; n = h.sum(); print(f"  {name:42s} mean {(h * np.arange(len(h))).sum() / max(n, 1):6.2f}  cumulative " + " ".join(f"{k}:{v:.3f}" for k, v in enumerate(np.cumsum(h) / max(n, 1)) if k in (0, 1, 2, 3, 4, 6, 8, 10, 12, 14, 16, 20, 24, 32)))

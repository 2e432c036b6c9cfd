"""Closed-loop (bench.py's loop: fixed state, evolving plan) per-plan-step counters of jh_engine_v5.hip with the hand's own contacts on:
kernel time, Newton iterations, dense-direction share, broad-phase survivors, phase split.  Needs a -DJH_V5_COUNT -DJH_V5_TICKS build (JUDO_AMD_LIB)."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, ".")
from judo_amd.controller import make_controller
from judo_amd import _lib
L = C.CDLL(_lib.LIB_PATH); L.jh_model_profile.argtypes = [C.c_void_p, C.POINTER(C.c_longlong)]
names = ["kin", "broad", "narrow", "rows", "grad", "matrix", "factor", "ls+int"]
c = make_controller("leap_cube", "mppi"); c.optimizer.config.num_rollouts = 65536; c.controller_cfg.horizon = 0.64
c.reset(); c.current_state = c.task.default_state(); c.system_metadata = {"goal_quat": np.array([0.0, 1.0, 0.0, 0.0])}
c.model.set_self_collision(os.environ.get("SELF", "1") == "1")
c.optimizer.seed(1234); c.record_kernel_events = True
t = 0.0
for i in range(int(os.environ.get("STEPS", "12"))):
    c.model.stats(); c.kernel_events.clear()
    c.time = t; c.update_action(); t += 1.0 / c.controller_cfg.control_freq; torch.cuda.synchronize()
    ms = c.kernel_events[-1][0].elapsed_time(c.kernel_events[-1][1])
    out = (C.c_int * 40)(); L.jh_model_hist(c.model.handle, out)
    prof = (C.c_longlong * 10)(); L.jh_model_profile(c.model.handle, prof)
    st = c.model.stats(reset=False)
    dense, its, l2, bp, ws, hh = out[0], out[1], out[2], out[3], out[4], out[5]
    nw = 65536 // 4 * 64; tot = sum(prof)
    print(f"step {i:2d}: {ms:6.1f} ms  iters/step {st['newton_iters'] / max(st['steps'], 1):5.2f} cap {st['newton_cap_hits']:6d} ovf {st['contact_overflow']:6d}  wave-its {its / max(ws, 1):5.2f} dense {dense / max(its, 1):6.2%}  bp {bp / (65536 * 64):.2f} hh {hh / (65536 * 64):.3f} cls none/match/forest/cycle {out[6] / (65536 * 64):.3f}/{out[7] / (65536 * 64):.3f}/{out[8] / (65536 * 64):.3f}/{out[9] / (65536 * 64):.4f} | "
          + " ".join(f"{n} {v / nw / 1e3:.0f}k" for n, v in zip(names, prof)) + f" | tot {tot / nw / 1e3:.0f}k")

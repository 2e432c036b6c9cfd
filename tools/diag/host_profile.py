"""cProfile of the host side of a small plan step (cartpole MPPI 4096 x 64, traces read in every step, as bench.py does).  usage: python tools/diag/host_profile.py [task] [N]"""
import cProfile, pstats, sys, time, io
import numpy as np, torch
sys.path.insert(0, ".")
from judo_amd.controller import make_controller
task = sys.argv[1] if len(sys.argv) > 1 else "cartpole"; N = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
c = make_controller(task, "mppi"); c.optimizer.config.num_rollouts = N; c.controller_cfg.horizon = 64 * c.task.dt
c.reset(); c.current_state = c.task.default_state()
t = 0.0
def time_path(fused, zero_copy, reps=3):
    global t
    c.fused_update, c.zero_copy_out = fused, zero_copy
    best = 1e9
    for _ in range(reps):
        for _ in range(50): c.time = t; c.update_action(); _ = c.traces; t += 0.05
        torch.cuda.synchronize(); T0 = time.perf_counter()
        for _ in range(500): c.time = t; c.update_action(); _ = c.traces; t += 0.05
        best = min(best, (time.perf_counter() - T0) / 500 * 1e6)
    return best
for name, f, z in (("jh_plan_step (one call, results written into the pinned block)", True, True), ("jh_update_fused + download", True, False), ("separate update kernels", False, False),
                   ("jh_plan_step again", True, True)):
    print(f"{task} N={N}: {time_path(f, z):7.1f} us per plan step incl. traces, best of 3 x 500  [{name}]")
c.fused_update = c.zero_copy_out = True
def step():
    global t
    c.time = t; c.update_action(); _ = c.traces; t += 0.05
for _ in range(100): step()
torch.cuda.synchronize()
T0 = time.perf_counter()
for _ in range(1000): step()
print(f"{task} N={N}: {(time.perf_counter() - T0) / 1000 * 1e6:.1f} us per plan step incl. traces (no profiler)")
pr = cProfile.Profile(); pr.enable()
for _ in range(1000): step()
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(45); print(s.getvalue()[:9000])
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(30); print(s.getvalue()[:6000])

"""Wall-clock split of a small plan step's host side: the controller's helper calls wrapped with timers (cartpole, 4096 rollouts)."""
import sys, time, collections
import numpy as np, torch
sys.path.insert(0, ".")
from judo_amd.controller import make_controller
c = make_controller("cartpole", "mppi"); c.optimizer.config.num_rollouts = 4096; c.controller_cfg.horizon = 64 * c.task.dt
c.reset(); c.current_state = c.task.default_state()
acc = collections.defaultdict(float); cnt = collections.Counter()
def wrap(obj, name):
    f = getattr(obj, name)
    def g(*a, **k):
        t0 = time.perf_counter(); r = f(*a, **k); acc[name] += time.perf_counter() - t0; cnt[name] += 1; return r
    setattr(obj, name, g)
for n in ("_pack_block", "_draw_noise", "_fetch", "_stage_traces", "_shifted_nominal", "_weights", "_buffers", "_current_normalizer", "_raw_bounds", "update_spline", "_prefetch_noise", "_num_trace_elites"):
    wrap(c, n)
for n in ("knot_sigma", "device_partial", "device_merge", "pre_optimization"):
    wrap(c.optimizer, n)
wrap(c.task, "task_params")
t = 0.0
for _ in range(50):
    c.time = t; c.update_action(); t += 0.05
acc.clear(); cnt.clear()
T0 = time.perf_counter()
for _ in range(500):
    c.time = t; c.update_action(); t += 0.05
tot = (time.perf_counter() - T0) / 500 * 1e6
print(f"update_action {tot:.1f} us per call (with timers)")
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print(f"  {k:22s} {v / 500 * 1e6:7.1f} us  ({cnt[k] / 500:.1f} calls)")
print(f"  unaccounted            {tot - sum(acc.values()) / 500 * 1e6 + acc['_stage_traces'] / 500 * 1e6 + acc['_prefetch_noise'] / 500 * 1e6:7.1f} us (stage_traces and prefetch run inside _fetch)")

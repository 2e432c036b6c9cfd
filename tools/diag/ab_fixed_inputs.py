"""Fair A/B of two library builds on the headline workload: the plan inputs (nominal knots, knot times) of 40 consecutive plan steps are recorded once
(`record`), then every build replays exactly those inputs with the same noise (`replay`), so the plans cannot drift apart between the variants."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from judo_amd.controller import make_controller
from tests import xcheck; xcheck.load()  # kernel generations 1 / 2 live in the test build
from judo_amd import engine_model as EM
if os.environ.get("LSTOL"): EM.SOLVER_LS_TOL = float(os.environ["LSTOL"])
if os.environ.get("TOL"): EM.SOLVER_TOL = float(os.environ["TOL"])
mode, path = sys.argv[1], sys.argv[2]
task = sys.argv[3] if len(sys.argv) > 3 else "leap_cube"
S = 40
OPT, NR, HS = {"leap_cube": ("mppi", 65536, 64), "fr3_pick": ("cem", 32768, 40)}[task]
c = make_controller(task, OPT); c.optimizer.config.num_rollouts = NR; c.controller_cfg.horizon = HS * c.task.dt
c.reset(); c.current_state = c.task.default_state(); c.system_metadata = {"goal_quat": np.array([0.0, 1.0, 0.0, 0.0])} if task == "leap_cube" else {}
if mode == "record":
    c.optimizer.seed(1234); rec = []; t = 0.0
    for i in range(S):
        sig = np.array(getattr(c.optimizer, "sigma", 0.0)).copy()
        rec.append((c.nominal_knots.copy(), c.times.copy(), t, sig)); c.time = t; c.update_action(); t += 0.05
    np.savez(path, knots=np.stack([r[0] for r in rec]), times=np.stack([r[1] for r in rec]), t=np.array([r[2] for r in rec]), sigma=np.stack([r[3] for r in rec]))
else:
    d = np.load(path)
    c.record_kernel_events = True
    if os.environ.get("KERNEL_GEN"): c.model.set_kernel(int(os.environ["KERNEL_GEN"]))
    if os.environ.get("FUSED_TRACES") is not None: c.fused_traces = os.environ["FUSED_TRACES"] == "1"
    if os.environ.get("SELF") is not None: c.model.set_self_collision(os.environ["SELF"] == "1")
    noms = []
    for rep in range(int(os.environ.get("REPS", "1"))):
        c.kernel_events.clear()
        for i in range(S):
            c.optimizer.seed(1000 + i)
            c.nominal_knots = d["knots"][i].copy(); c.times = d["times"][i].copy(); c.update_spline(c.times, c.nominal_knots); c.time = float(d["t"][i])
            if OPT == "cem": c.optimizer.sigma = d["sigma"][i].copy()
            c.update_action()
            if rep == 0: noms.append(c.nominal_knots.copy())
        torch.cuda.synchronize()
        k = np.array([a.elapsed_time(b) for a, b in c.kernel_events])
        if os.environ.get("OUT") and rep == 0: np.save(os.environ["OUT"], np.stack(noms))
        if os.environ.get("REF") and rep == 0:
            dn = np.abs(np.stack(noms) - np.load(os.environ["REF"]))
            per_step = dn.reshape(dn.shape[0], -1).max(axis=1)
            print(f"  returned nominal vs {os.environ['REF']}: median {np.median(dn):.2e} rad, 99th percentile {np.percentile(dn, 99):.2e}, max {dn.max():.2e}; plan steps (of {len(per_step)}) whose nominal "
                  f"moved by more than 2e-3 rad anywhere: {int((per_step > 2e-3).sum())} (MPPI at temperature 0.0025 is close to an argmin: such a step picked another winner)")
        if os.environ.get("HIST"):
            import ctypes as C
            from judo_amd import _lib
            L = _lib.lib(); L.jh_model_hist.argtypes = [C.c_void_p, C.POINTER(C.c_int)]; hh = (C.c_int * 40)(); L.jh_model_hist(c.model.handle, hh)
            print("  solver exits (EXITSTATS builds) gradient / not-descent / expected-decrease / cap / rounding-floor:", list(hh)[:5])
        st = c.model.stats()
        print(f"  contacts dropped above the pool {st['contact_overflow']} ({st['contact_overflow'] / max(st['steps'], 1):.2e} per step), Newton cap hits {st['newton_cap_hits']}")
        print(f"{os.environ.get('JUDO_AMD_LIB', 'default')} lstol={EM.SOLVER_LS_TOL:g} tol={EM.SOLVER_TOL:g}: kernel mean {k.mean():.2f} ms  (first 10: {k[:10].mean():.2f}, last 10: {k[-10:].mean():.2f})  iters/step {st['newton_iters'] / (NR * HS * S):.3f}")

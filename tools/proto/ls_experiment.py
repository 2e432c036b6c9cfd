"""CPU prototype of line-search variants for the engine kernels (fp64 oracle with the kernels' search, oracle/jo_engine.c::jo_set_ls_experiment) on rollouts of the recorded
headline plan steps: slope evaluations per Newton iteration of a rollout and of a "wave" (maximum over 4 consecutive rollouts at the same iteration number), Newton iterations.
usage: python tools/proto/ls_experiment.py [plan step] [rollouts]"""
import ctypes as C, os, sys
os.environ["JUDO_ORACLE_EXPERIMENTS"] = "1"  # oracle/libjudo_oracle_exp.so: the solver experiments are not in the parity oracle
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import oracle as O
from judo_amd.tasks import LeapCube
step = int(sys.argv[1]) if len(sys.argv) > 1 else 32
N = int(sys.argv[2]) if len(sys.argv) > 2 else 32
task = LeapCube(); om = O.Model("leap_cube"); L = O.lib()
L.jo_set_solver.argtypes = [C.c_void_p, C.c_double, C.c_int]
L.jo_set_ls_experiment.argtypes = [C.c_int, C.c_double, C.c_int, C.c_void_p, C.c_long]; L.jo_ls_log_size.restype = C.c_long
d = np.load("tools/diag/ab_inputs_leap.npz")
H, K = 64, 4
rng = np.random.default_rng(step)
nom = d["knots"][step]
sig = O.mppi_sigma(0.2, True, 4.0, K, 16)
knots = O.sample_knots(nom, rng.standard_normal((N - 1, K, 16)), sig)
r = task.actuator_ctrlrange; knots = O.clip_knots(knots, r[:, 0], r[:, 1])
W = O.spline_weights("cubic", d["times"][step], d["t"][step] + 0.01 * np.arange(H))
U = O.spline_eval(W, knots)
x0 = np.asarray(task.default_state(), float)
L.jo_set_solver(om.ptr, 1e-6, 50)
log = np.zeros(4_000_000, np.int32)
for mode, lsmax in ((1, 16), (5, 16), (6, 16), (7, 16)):
    L.jo_set_ls_experiment(mode, 1e-2, lsmax, log.ctypes.data, len(log))
    st, se = om.rollout(x0, U, nthread=1)
    n = L.jo_ls_log_size(); a = log[:n].copy()
    # split into solves: marker -1, nrows, then evals per iteration
    starts = np.nonzero(a == -1)[0]
    solves = [a[s + 2 : e] for s, e in zip(starts, list(starts[1:]) + [n])]
    # solves appear rollout by rollout, step by step, but steps without rows (free flight of everything) log nothing: group by rollout via the total count = N * H only if all steps have rows
    its = np.array([len(s) for s in solves]); ev = np.concatenate(solves) if solves else np.zeros(0)
    per_solve_mean = its.mean()
    # "waves": 4 consecutive rollouts; needs a fixed number of solves per rollout -> every step of the leap hand has its 16 friction-loss rows, so N * H solves
    assert len(solves) == N * H, (len(solves), N * H)
    grid = [[solves[rr * H + h] for h in range(H)] for rr in range(N)]
    wave_ev, wave_it = [], []
    for w in range(N // 4):
        for h in range(H):
            ss = [grid[4 * w + k][h] for k in range(4)]; m = max(len(s) for s in ss); wave_it.append(m)
            for i in range(m): wave_ev.append(max(int(s[i]) for s in ss if len(s) > i))
    wave_ev = np.array(wave_ev)
    print(f"mode {mode} lsmax {lsmax}: Newton iterations / rollout-step {per_solve_mean:.2f}, / wave-step {np.mean(wave_it):.2f};  slope evaluations / rollout iteration {ev.mean():.2f} "
          f"(>= 9: {100 * (ev >= 9).mean():.1f} %, at the cap: {100 * (ev >= lsmax).mean():.1f} %), / wave iteration {wave_ev.mean():.2f} (>= 9: {100 * (wave_ev >= 9).mean():.1f} %); "
          f"evaluations / wave-step {wave_ev.sum() / len(wave_it):.1f};  cube z at the horizon {st[:3, -1, 2]}")
L.jo_set_ls_experiment(0, 1e-2, 16, None, 0)

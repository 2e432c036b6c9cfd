"""Random sizes through the update kernels' drop-in form (`Optimizer.update_nominal_knots(sampled_knots, rewards)`): MPPI / CEM / PS against the oracle (pinned to the
reference's update_nominal_knots by tests/test_oracle.py), incl. ties, K * nu up to the limit of 512 and up to 32 elites."""
import sys
import numpy as np
sys.path.insert(0, ".")
from judo_amd.optimizers import get_registered_optimizers
from oracle import oracle as O
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
R = get_registered_optimizers()
worst = []
for case in range(150):
    name = ["mppi", "cem", "ps"][rng.integers(3)]
    N = int(rng.choice([1, 2, 3, 31, 64, 65, 255, 256, 257, 1000, 4097, 20000]))
    nu = int(rng.choice([1, 2, 8, 16, 19])); K = int(rng.integers(1, min(32, 512 // nu) + 1))
    cls, cfg_cls = R[name]; cfg = cfg_cls(); cfg.num_rollouts, cfg.num_nodes = N, K
    if name == "mppi": cfg.temperature = float(10 ** rng.uniform(-3, 0.5))
    if name == "cem": cfg.num_elites = int(rng.integers(1, min(N, 32) + 1))
    opt = cls(cfg, nu)
    knots = rng.standard_normal((N, K, nu)); rewards = rng.standard_normal(N) * 10 ** rng.uniform(-2, 2)
    if rng.integers(3) == 0 and N > 3: rewards[rng.integers(N, size=max(2, N // 4))] = rewards.max()  # ties for the best
    desc = f"{name} N={N} K={K} nu={nu}" + (f" E={cfg.num_elites}" if name == "cem" else "") + (f" T={cfg.temperature:.1e}" if name == "mppi" else "")
    try:
        out = opt.update_nominal_knots(knots.copy(), rewards.copy())
        k32, r32 = knots.astype(np.float32).astype(np.float64), rewards.astype(np.float32).astype(np.float64)  # what the device sees
        if name == "mppi": ref = O.mppi_update(k32, r32, cfg.temperature)
        elif name == "ps": ref = O.ps_update(k32, r32)
        else:
            ref, sig, _ = O.cem_update(k32, r32, cfg.num_elites, cfg.sigma_min, cfg.sigma_max)
            worst.append((float(np.abs(np.asarray(opt.sigma) - sig).max()), desc + " (sigma)"))
        worst.append((float(np.abs(out - ref).max()), desc))
    except Exception as ex:
        worst.append((float("inf"), desc + " EXC " + repr(ex)[:160]))
worst.sort(key=lambda t: -t[0])
for e, d in worst[:10]: print(f"  {e:.2e}  {d}")
print("median %.2e over %d checks" % (np.median([w[0] for w in worst]), len(worst)))

#!/usr/bin/env python3
"""Generate golden vectors for the sampling-MPC hot path from the reference's own numpy code.

Runs ONLY in the build container (needs `/root/reference`).  It imports the reference
(judo v0.0.7) through `tools/_ref_import.py` and records inputs/outputs of the functions
on the hot path as small `.npz` / `.json` fixtures in `tests/golden/`.
The fixtures are data; no reference source is copied.

Covered reference functions (file:line relative to /root/reference):
  G1  MPPI/PS/CEM.sample_control_knots      judo/optimizers/mppi.py:38-59, ps.py:29-50, cem.py:55-74
  G2  MPPI/CEM/PS.update_nominal_knots      judo/optimizers/mppi.py:61-82, cem.py:76-92, ps.py:52-65
      CEM.pre_optimization (K change)       judo/optimizers/cem.py:44-53
  G3  make_spline + evaluation              judo/controller/controller.py:382-401, :220-221, :261-262
  G4  Cartpole/CylinderPush/LeapCube/FR3Pick.reward
                                            judo/tasks/cartpole.py:42-78, cylinder_push.py:50-93,
                                            leap_cube.py:63-88, fr3_pick.py:225-311,
                                            judo/utils/math_utils.py:95-104
  G5  per-task override-resolved configs    judo/optimizers/overrides.py, judo/controller/overrides.py
  G6  MinMaxNormalizer round trip           judo/utils/normalization.py:94-138

`np.random.randn` draws are reproduced by re-seeding and drawing the same shape, so every
fixture also stores the exact noise tensor the reference consumed ("same noise injected on
both sides", SURVEY.md section 7 hard part 5).
"""

from __future__ import annotations

import json
import os
import sys
from types import SimpleNamespace

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _ref_import  # noqa: E402

_ref_import.install()

import scipy  # noqa: E402
from judo.controller.controller import ControllerConfig, make_spline  # noqa: E402
from judo.optimizers.cem import CrossEntropyMethod, CrossEntropyMethodConfig  # noqa: E402
from judo.optimizers.mppi import MPPI, MPPIConfig  # noqa: E402
from judo.optimizers.ps import PredictiveSampling, PredictiveSamplingConfig  # noqa: E402
from judo.tasks.cartpole import Cartpole, CartpoleConfig  # noqa: E402
from judo.tasks.cylinder_push import CylinderPush, CylinderPushConfig  # noqa: E402
from judo.tasks.fr3_pick import QPOS_HOME as FR3_QPOS_HOME  # noqa: E402
from judo.tasks.fr3_pick import FR3Pick, FR3PickConfig, Phase  # noqa: E402
from judo.tasks.leap_cube import LeapCube, LeapCubeConfig  # noqa: E402
from judo.utils.normalization import MinMaxNormalizer, RunningMeanStdNormalizer  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
META = {
    "reference": "bdaiinstitute/judo v0.0.7 (/root/reference)",
    "numpy": np.__version__,
    "scipy": scipy.__version__,
}


def _draw(seed: int, shape: tuple[int, ...]) -> np.ndarray:
    np.random.seed(seed)
    return np.random.randn(*shape)


# --------------------------------------------------------------------------- G1 / G2
def gen_optimizers() -> dict[str, np.ndarray]:
    out: dict[str, np.ndarray] = {}
    rng = np.random.default_rng(7)
    case = 0
    # ---- MPPI + PS sampling
    for name, cls, cfg_cls in (("mppi", MPPI, MPPIConfig), ("ps", PredictiveSampling, PredictiveSamplingConfig)):
        for nu in (1, 2, 16):
            for ramp in (False, True):
                for K in (4, 8):
                    N = 8
                    cfg = cfg_cls(num_rollouts=N, num_nodes=K, use_noise_ramp=ramp, noise_ramp=4.0 if nu == 16 else 2.5)
                    cfg.sigma = 0.2 if nu == 16 else cfg.sigma
                    opt = cls(cfg, nu)
                    nominal = rng.standard_normal((K, nu))
                    seed = 100 + case
                    noise = _draw(seed, (N - 1, K, nu))
                    np.random.seed(seed)
                    sampled = opt.sample_control_knots(nominal)
                    key = f"sample_{name}_{case}"
                    out[key + "_nominal"] = nominal
                    out[key + "_noise"] = noise
                    out[key + "_params"] = np.array([N, K, nu, float(ramp), cfg.noise_ramp, cfg.sigma])
                    out[key + "_out"] = sampled
                    case += 1
    # ---- CEM sampling: cumulative ramp (called twice) + K change through pre_optimization
    for nu in (1, 8):
        for ramp in (False, True):
            K, N = 4, 8
            cfg = CrossEntropyMethodConfig(
                num_rollouts=N, num_nodes=K, use_noise_ramp=ramp, noise_ramp=4.0, sigma_min=0.01, sigma_max=0.3, num_elites=3
            )
            opt = CrossEntropyMethod(cfg, nu)
            key = f"sample_cem_{case}"
            out[key + "_params"] = np.array([N, K, nu, float(ramp), cfg.noise_ramp, cfg.sigma_min, cfg.sigma_max])
            out[key + "_sigma0"] = opt.sigma.copy()
            for call in range(2):
                nominal = rng.standard_normal((K, nu))
                seed = 200 + case * 4 + call
                noise = _draw(seed, (N - 1, K, nu))
                np.random.seed(seed)
                sampled = opt.sample_control_knots(nominal)
                out[f"{key}_call{call}_nominal"] = nominal
                out[f"{key}_call{call}_noise"] = noise
                out[f"{key}_call{call}_out"] = sampled
                out[f"{key}_call{call}_sigma_after"] = opt.sigma.copy()
            # K change 4 -> 6: pre_optimization re-interpolates sigma with linear extrapolation
            opt.sigma = np.abs(rng.standard_normal((K, nu))) * 0.1 + 0.02
            out[key + "_prek_sigma_in"] = opt.sigma.copy()
            old_times = 1.5 + np.linspace(0, 1.0, K)
            cfg.num_nodes = 6
            new_times = 1.55 + np.linspace(0, 1.0, 6)
            opt.pre_optimization(old_times, new_times)
            out[key + "_prek_old_times"] = old_times
            out[key + "_prek_new_times"] = new_times
            out[key + "_prek_sigma_out"] = opt.sigma.copy()
            case += 1

    # ---- updates
    ucase = 0
    for nu, K, N in ((1, 4, 8), (2, 4, 33), (16, 4, 64), (8, 5, 17)):
        knots = rng.standard_normal((N, K, nu))
        rew_sets = {
            "rand": -np.abs(rng.standard_normal(N)) * 3.0,
            "ties": np.round(-np.abs(rng.standard_normal(N)) * 2.0),  # many exact ties
            "dominant": np.concatenate([[-0.001], -50.0 - np.abs(rng.standard_normal(N - 1))]),
            "equal": -np.ones(N) * 2.5,
        }
        for rname, rewards in rew_sets.items():
            key = f"update_{ucase}"
            out[key + "_knots"] = knots
            out[key + "_rewards"] = rewards
            out[key + "_tag"] = np.array([N, K, nu])
            for lam in (0.05, 0.0025):
                opt = MPPI(MPPIConfig(num_rollouts=N, num_nodes=K, temperature=lam), nu)
                out[f"{key}_mppi_{lam}"] = opt.update_nominal_knots(knots, rewards)
            opt = PredictiveSampling(PredictiveSamplingConfig(num_rollouts=N, num_nodes=K), nu)
            out[key + "_ps"] = opt.update_nominal_knots(knots, rewards)
            for k in (2, 3):
                cfg = CrossEntropyMethodConfig(num_rollouts=N, num_nodes=K, num_elites=k, sigma_min=0.01, sigma_max=0.3)
                opt = CrossEntropyMethod(cfg, nu)
                out[f"{key}_cem{k}_nominal"] = opt.update_nominal_knots(knots, rewards)
                out[f"{key}_cem{k}_sigma"] = opt.sigma.copy()
                # the elite index set numpy's argsort picked (ties: implementation-defined; recorded so
                # a tie-tolerant check is possible)
                out[f"{key}_cem{k}_elite_idx"] = np.flip(np.argsort(rewards))[:k].copy()
            out[key + "_name"] = np.array(rname)
            ucase += 1
    return out


# --------------------------------------------------------------------------- G3
def gen_spline() -> dict[str, np.ndarray]:
    out: dict[str, np.ndarray] = {}
    rng = np.random.default_rng(11)
    cfgs = {
        # name: (kind, K, H, dt, horizon)
        "cartpole_ps": ("zero", 4, 50, 0.04, 2.0),
        "cartpole_mppi": ("zero", 4, 64, 0.04, 2.56),
        "cylinder_mppi": ("zero", 4, 64, 0.02, 1.28),
        "fr3_cem": ("linear", 4, 40, 0.004, 0.16),
        "leap_mppi": ("cubic", 4, 64, 0.01, 0.64),
        "leap_k8": ("cubic", 8, 64, 0.01, 0.64),
        "lin_k8": ("linear", 8, 50, 0.04, 2.0),
        "zero_k8": ("zero", 8, 64, 0.02, 1.28),
        "cubic_k6": ("cubic", 6, 30, 0.01, 1.0),
    }
    for name, (kind, K, H, dt, horizon) in cfgs.items():
        for t0 in (0.0, 3.7):
            new_times = t0 + np.linspace(0, horizon, K, endpoint=True)  # controller.py:220 / :160
            q = t0 + dt * np.arange(H)  # controller.py:262 / :155
            # W[h, k]: response to unit knot k (the spline is linear in the knots)
            eye = np.eye(K)[:, :, None]  # (K "batch", K, 1)
            W = make_spline(new_times, eye, kind)(q)[:, :, 0].T  # (H, K)
            nu = 3
            knots = rng.standard_normal((5, K, nu))
            U = make_spline(new_times, knots, kind)(q)
            # time-shift re-sampling of the previous plan (controller.py:220-221) incl. hold-ends
            shift = 0.05
            shifted_times = (t0 + shift) + np.linspace(0, horizon, K, endpoint=True)
            prev = make_spline(new_times, knots[0], kind)
            shifted = prev(shifted_times)
            far = prev(np.array([t0 - 1.0, t0 + horizon + 2.0]))
            key = f"{name}_t{int(t0 * 10)}"
            out[key + "_cfg"] = np.array([{"zero": 0, "linear": 1, "cubic": 3}[kind], K, H, dt, horizon, t0])
            out[key + "_W"] = W
            out[key + "_knots"] = knots
            out[key + "_U"] = U
            out[key + "_shift_times"] = shifted_times
            out[key + "_shift_knots"] = shifted
            out[key + "_far"] = far
    return out


# --------------------------------------------------------------------------- G4
def _unit(q: np.ndarray) -> np.ndarray:
    return q / np.linalg.norm(q, axis=-1, keepdims=True)


def gen_rewards() -> dict[str, np.ndarray]:
    out: dict[str, np.ndarray] = {}
    rng = np.random.default_rng(23)
    N, H = 5, 7
    # cartpole
    self = SimpleNamespace(config=CartpoleConfig())
    states = rng.standard_normal((N, H, 4)) * 2
    controls = rng.standard_normal((N, H, 1))
    out["cartpole_states"] = states
    out["cartpole_controls"] = controls
    out["cartpole_reward"] = Cartpole.reward(self, states, None, controls)
    # cylinder push (default goal (0,0) and a moved goal)
    for gi, goal in enumerate((np.array([0.0, 0.0]), np.array([0.3, -0.2]))):
        cfg = CylinderPushConfig()
        cfg.goal_pos = goal
        self = SimpleNamespace(config=cfg)
        states = rng.standard_normal((N, H, 8))
        out[f"cylinder{gi}_states"] = states
        out[f"cylinder{gi}_goal"] = goal
        out[f"cylinder{gi}_reward"] = CylinderPush.reward(self, states, None, rng.standard_normal((N, H, 2)))
    # leap cube: random quats + edge cases (== goal, antipodal, tiny vector part, angle > pi branch)
    self = SimpleNamespace(config=LeapCubeConfig(), goal_pos=np.array([0.0, 0.03, 0.1]))
    nx = 23 + 22
    for gi, goal_quat in enumerate((np.array([1.0, 0, 0, 0]), np.array([0.0, 1.0, 0, 0]), _unit(rng.standard_normal(4)))):
        states = rng.standard_normal((N + 4, H, nx)) * 0.1
        states[..., 3:7] = _unit(rng.standard_normal((N + 4, H, 4)))
        states[N + 0, :, 3:7] = goal_quat  # exactly at goal
        states[N + 1, :, 3:7] = -goal_quat  # antipodal representation of the goal
        tiny = goal_quat + np.array([0, 1e-8, -2e-8, 1e-8])
        states[N + 2, :, 3:7] = tiny / np.linalg.norm(tiny)  # |v| < 1e-6 branch
        states[N + 3, :, 3:7] = _unit(np.array([-0.3, 0.5, 0.7, -0.1]))  # w<0 => speed>pi wrap
        out[f"leap{gi}_states"] = states
        out[f"leap{gi}_goal_quat"] = goal_quat
        out[f"leap{gi}_reward"] = LeapCube.reward(self, states, None, None, {"goal_quat": goal_quat})
    # default metadata path (system_metadata None)
    out["leap_default_reward"] = LeapCube.reward(self, out["leap0_states"], None, None, None)

    # fr3 pick: all four phases, finger-touch boundary dist == 0
    nq, nv, ns = 16, 15, 14
    # sensor addresses in MJCF order (fr3_components/params_and_default.xml:58-76):
    # 5 distance (1 each), framezaxis (3), framepos trace_object (3), framepos trace_grasp_site (3)
    adr = dict(left_finger_obj=0, right_finger_obj=1, left_finger_table=2, right_finger_table=3, obj_table=4, ee_z=5, trace_object=8, grasp=11)
    for phase in Phase:
        self = SimpleNamespace(
            config=FR3PickConfig(),
            phase=phase,
            model=SimpleNamespace(nq=nq, nv=nv),
            obj_pos_adr=0,
            obj_pos_slice=slice(0, 3),
            arm_pos_slice=slice(7, 16),
            left_finger_table_adr=adr["left_finger_table"],
            right_finger_table_adr=adr["right_finger_table"],
            left_finger_obj_adr=adr["left_finger_obj"],
            right_finger_obj_adr=adr["right_finger_obj"],
            obj_table_adr=adr["obj_table"],
            grasp_site_adr=adr["grasp"],
            ee_z_adr=adr["ee_z"],
            ee_z_slice=slice(adr["ee_z"], adr["ee_z"] + 3),
        )
        self.check_sensor_dists = lambda sensors, pair, _s=self: FR3Pick.check_sensor_dists(_s, sensors, pair)
        states = rng.standard_normal((N, H, nq + nv)) * 0.3
        sensors = rng.standard_normal((N, H, ns)) * 0.2
        sensors[0, 0, adr["left_finger_table"]] = 0.0  # boundary: dist == 0 counts as touching
        sensors[1, 1, adr["right_finger_table"]] = 0.0
        sensors[2, :, adr["left_finger_table"]] = np.abs(sensors[2, :, adr["left_finger_table"]]) + 0.01
        sensors[2, :, adr["right_finger_table"]] = np.abs(sensors[2, :, adr["right_finger_table"]]) + 0.01
        out[f"fr3_{phase.name}_states"] = states
        out[f"fr3_{phase.name}_sensors"] = sensors
        out[f"fr3_{phase.name}_reward"] = FR3Pick.reward(self, states, sensors, None)
    out["fr3_qpos_home"] = FR3_QPOS_HOME
    out["fr3_sensor_adr"] = np.array([adr[k] for k in ("left_finger_obj", "right_finger_obj", "left_finger_table", "right_finger_table", "obj_table", "ee_z", "trace_object", "grasp")])
    return out


# --------------------------------------------------------------------------- G5
def gen_configs() -> dict:
    res: dict = {"meta": META, "optimizer": {}, "controller": {}}
    for task in ("cartpole", "cylinder_push", "fr3_pick", "leap_cube", "leap_cube_down", "caltech_leap_cube"):
        res["optimizer"][task] = {}
        for name, cfg_cls in (("mppi", MPPIConfig), ("cem", CrossEntropyMethodConfig), ("ps", PredictiveSamplingConfig)):
            cfg = cfg_cls()
            cfg.set_override(task)
            res["optimizer"][task][name] = {k: v for k, v in vars(cfg).items()}
        c = ControllerConfig()
        c.set_override(task)
        res["controller"][task] = {k: v for k, v in vars(c).items()}
    res["optimizer"]["default"] = {
        "mppi": vars(MPPIConfig()),
        "cem": vars(CrossEntropyMethodConfig()),
        "ps": vars(PredictiveSamplingConfig()),
    }
    res["controller"]["default"] = vars(ControllerConfig())
    res["task_defaults"] = {
        "cartpole": vars(CartpoleConfig()),
        "leap_cube": vars(LeapCubeConfig()),
    }
    return res


# --------------------------------------------------------------------------- G6
def gen_normalizer() -> dict[str, np.ndarray]:
    rng = np.random.default_rng(5)
    lo = np.array([-1.8, -10.0, -np.inf, 0.0])
    hi = np.array([1.8, 10.0, np.inf, 0.04])
    import warnings

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        nrm = MinMaxNormalizer(4, lo, hi)
    x = rng.standard_normal((6, 3, 4))
    out = {"minmax_lo": lo, "minmax_hi": hi, "minmax_x": x, "minmax_norm": nrm.normalize(x), "minmax_denorm": nrm.denormalize(x)}
    # "running" action normaliser inside the optimiser loop of Controller.update_action (judo/controller/controller.py:222-296):
    # normalise the nominal, sample + clip in normalised units, denormalise the candidates, update on the normalised candidates,
    # update the running statistics with the raw candidates, denormalise the result with the UPDATED statistics.
    # The rollout + Task.reward in the middle is replaced by a deterministic stand-in reward so that the sequence needs no simulator.
    N, K, nu, iters = 24, 4, 3, 3
    rlo, rhi = np.array([-1.0, -np.inf, 0.0]), np.array([1.0, np.inf, 0.5])
    run = RunningMeanStdNormalizer(nu)
    out["running_defaults"] = np.array([1.0, run.min_std, run.max_std, run.eps])
    opt = MPPI(MPPIConfig(num_rollouts=N, num_nodes=K, sigma=0.3, temperature=0.05, use_noise_ramp=True, noise_ramp=2.0), nu)
    nominal = rng.standard_normal((K, nu)) * 0.3 + np.array([0.2, 1.5, 0.25])
    target = np.array([0.5, 1.0, 0.1])
    out["running_nominal_in"] = nominal.copy()
    out["running_lo"], out["running_hi"], out["running_target"] = rlo, rhi, target
    out["running_cfg"] = np.array([N, K, nu, iters, 0.3, 0.05, 2.0])
    nominal_n = run.normalize(nominal)
    for it in range(iters):
        seed = 900 + it
        out[f"running_it{it}_noise"] = _draw(seed, (N - 1, K, nu))
        np.random.seed(seed)
        cand_n = opt.sample_control_knots(nominal_n)
        cand_n = np.clip(cand_n, run.normalize(rlo), run.normalize(rhi))
        cand = run.denormalize(cand_n)
        rewards = -np.sum((cand - target) ** 2, axis=(1, 2))
        nominal_n = opt.update_nominal_knots(cand_n, rewards)
        run.update(cand)
        out[f"running_it{it}_candidates"] = cand
        out[f"running_it{it}_rewards"] = rewards
        out[f"running_it{it}_nominal_normalized"] = nominal_n.copy()
        out[f"running_it{it}_state"] = np.concatenate([[run.count], run.mean, run.std, run.M2])
    out["running_nominal_out"] = run.denormalize(nominal_n)
    return out


def main() -> None:
    os.makedirs(OUT, exist_ok=True)
    np.savez_compressed(os.path.join(OUT, "optimizers.npz"), **gen_optimizers())
    np.savez_compressed(os.path.join(OUT, "spline.npz"), **gen_spline())
    np.savez_compressed(os.path.join(OUT, "rewards.npz"), **gen_rewards())
    np.savez_compressed(os.path.join(OUT, "normalizer.npz"), **gen_normalizer())
    with open(os.path.join(OUT, "configs.json"), "w") as f:
        json.dump(gen_configs(), f, indent=1, default=lambda o: o.tolist() if hasattr(o, "tolist") else str(o))
    with open(os.path.join(OUT, "META.json"), "w") as f:
        json.dump(META, f, indent=1)
    for fn in sorted(os.listdir(OUT)):
        print(fn, os.path.getsize(os.path.join(OUT, fn)))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Golden vectors of the reference's own plan-step harness: `Controller.update_action` and `Controller.update_traces`
(judo/controller/controller.py:210-299, :323-363), executed here from the imported reference with a plugin task and a plugin
rollout backend (`tests/plugin_fixture.py`: closed-form numpy on both sides) and the reference's MPPI / CEM / PS optimizers.

Runs ONLY in the build container (needs /root/reference).  Writes tests/golden/controller.npz (data only):
  traces_*        update_traces called unbound on hand-made sensors / rewards (ties, E > N, interleaved sensor columns)
  plan_<case>_*   three consecutive plan steps (time advancing 0.05 s) per case: every np.random.randn draw the reference consumed,
                  the nominal knots after each plan step, the last iteration's rewards and the traces

The Controller object is built without its __init__ (which needs a real MjModel): every attribute __init__ would set is set
here by hand, then the reference's reset() / update_action() run unmodified.
"""

from __future__ import annotations

import os
import sys
from types import SimpleNamespace

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _ref_import  # noqa: E402

_ref_import.install()

from judo.controller.controller import Controller, ControllerConfig  # noqa: E402
from judo.optimizers.cem import CrossEntropyMethod, CrossEntropyMethodConfig  # noqa: E402
from judo.optimizers.mppi import MPPI, MPPIConfig  # noqa: E402
from judo.optimizers.ps import PredictiveSampling, PredictiveSamplingConfig  # noqa: E402

from tests import plugin_fixture as PF  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


class RefPluginTask:
    """Duck-typed judo Task (judo/tasks/base.py:24-203) around the plugin arithmetic."""

    uses_locomotion_policy = False
    nu = PF.NU
    dt = PF.DT

    def __init__(self) -> None:
        self.model = SimpleNamespace(nq=PF.NQ, nv=PF.NV, nu=PF.NU, nsensordata=PF.NS, sensor_adr=np.array(PF.SENSOR_ADR))
        self.data = SimpleNamespace(qpos=np.zeros(PF.NQ), qvel=np.zeros(PF.NV), time=0.0)
        self.config = SimpleNamespace()

    @property
    def time(self) -> float:
        return self.data.time

    @time.setter
    def time(self, v: float) -> None:
        self.data.time = v

    @property
    def actuator_ctrlrange(self) -> np.ndarray:
        return PF.CTRLRANGE

    def reward(self, states, sensors, controls, system_metadata=None):
        return PF.reward_numpy(states, sensors, controls)

    def pre_rollout(self, curr_state) -> None: ...

    def post_rollout(self, states, sensors, controls, system_metadata=None) -> None: ...

    def task_to_sim_ctrl(self, controls):
        return controls

    def optimizer_warm_start(self) -> np.ndarray:
        return np.zeros(PF.NU)

    def reset(self) -> None:
        self.data.qpos = np.array([0.1, -0.3])
        self.data.qvel = np.zeros(PF.NV)


def build_reference_controller(opt, ctrl_cfg: ControllerConfig) -> Controller:
    c = Controller.__new__(Controller)
    c._controller_cfg = ctrl_cfg
    c.task = RefPluginTask()
    c.optimizer = opt
    c.available_optimizers, c.available_tasks = {}, {}
    c.model = c.task.model
    c.rollout_backend = PF.NumpyBackend(opt.config.num_rollouts)
    c._last_policy_output = None
    c.action_normalizer = c._init_action_normalizer()
    c.system_metadata = {}
    H = c.num_timesteps
    N = opt.config.num_rollouts
    c.states = np.zeros((N, H, PF.NQ + PF.NV))
    c.current_state = np.concatenate([c.task.data.qpos, c.task.data.qvel])
    c.sensors = np.zeros((N, H, PF.NS))
    c.rollout_controls = np.zeros((N, H, PF.NU))
    c.rewards = np.zeros((N,))
    c.reset()
    c.traces = None
    c.trace_sensors = [0, 1]
    c.num_trace_elites = min(c.max_num_traces, len(c.rewards))
    c.num_trace_sensors = 2
    c.sensor_rollout_size = c.num_timesteps - 1
    c.all_traces_rollout_size = c.sensor_rollout_size * c.num_trace_sensors
    return c


class _RecordingRandn:
    """np.random.randn replacement that records every draw (the optimizers call np.random.randn(N-1, K, nu))."""

    def __init__(self, seed: int) -> None:
        self.rs = np.random.RandomState(seed)
        self.draws: list[np.ndarray] = []

    def __call__(self, *shape):
        out = self.rs.randn(*shape)
        self.draws.append(out.copy())
        return out


def gen_traces() -> dict[str, np.ndarray]:
    out: dict[str, np.ndarray] = {}
    rng = np.random.default_rng(3)
    cases = [
        dict(N=8, H=6, ns=9, adr=[1, 5], E=3, rewards=np.array([0.3, 2.0, -1.0, 2.0, 0.5, 2.0, -3.0, 0.1])),  # three-way tie for the best
        dict(N=3, H=5, ns=6, adr=[0, 3], E=5, rewards=np.array([1.0, 1.0, -2.0])),  # more traces asked for than rollouts
        dict(N=6, H=4, ns=4, adr=[1], E=1, rewards=rng.standard_normal(6)),
    ]
    for i, cs in enumerate(cases):
        sensors = rng.standard_normal((cs["N"], cs["H"], cs["ns"]))
        ns = SimpleNamespace(num_timesteps=cs["H"], num_trace_sensors=len(cs["adr"]), num_trace_elites=0, max_num_traces=cs["E"],
                             optimizer_cfg=SimpleNamespace(num_rollouts=cs["N"]), sensors=sensors, rewards=cs["rewards"],
                             model=SimpleNamespace(sensor_adr=np.array(cs["adr"])), trace_sensors=list(range(len(cs["adr"]))))
        Controller.update_traces(ns)
        out[f"traces{i}_sensors"], out[f"traces{i}_rewards"], out[f"traces{i}_adr"] = sensors, cs["rewards"], np.array(cs["adr"])
        out[f"traces{i}_E"], out[f"traces{i}_out"] = np.array(cs["E"]), ns.traces
    out["traces_cases"] = np.array(len(cases))
    return out


PLAN_CASES = {
    # name: (optimizer, config kwargs, controller config kwargs)
    "mppi_linear": ("mppi", dict(num_rollouts=24, num_nodes=4, sigma=0.3, temperature=0.05, use_noise_ramp=True, noise_ramp=2.0),
                    dict(horizon=0.48, spline_order="linear", max_opt_iters=2, max_num_traces=3)),
    "mppi_cubic_minmax": ("mppi", dict(num_rollouts=16, num_nodes=5, sigma=0.2, temperature=0.1, use_noise_ramp=False),
                          dict(horizon=0.6, spline_order="cubic", max_opt_iters=1, max_num_traces=2, action_normalizer="min_max")),
    "cem_zero": ("cem", dict(num_rollouts=20, num_nodes=4, sigma_min=0.05, sigma_max=0.8, num_elites=3, use_noise_ramp=True, noise_ramp=2.5),
                 dict(horizon=0.4, spline_order="zero", max_opt_iters=2, max_num_traces=5)),
    "ps_linear_running": ("ps", dict(num_rollouts=12, num_nodes=4, sigma=0.25, use_noise_ramp=False),
                          dict(horizon=0.48, spline_order="linear", max_opt_iters=3, max_num_traces=1, action_normalizer="running")),
    "mppi_linear_running": ("mppi", dict(num_rollouts=16, num_nodes=4, sigma=0.3, temperature=0.2, use_noise_ramp=True, noise_ramp=2.0),
                            dict(horizon=0.48, spline_order="linear", max_opt_iters=2, max_num_traces=2, action_normalizer="running")),
}
_OPTS = {"mppi": (MPPI, MPPIConfig), "cem": (CrossEntropyMethod, CrossEntropyMethodConfig), "ps": (PredictiveSampling, PredictiveSamplingConfig)}


def gen_plans() -> dict[str, np.ndarray]:
    out: dict[str, np.ndarray] = {}
    real_randn = np.random.randn
    for ci, (name, (opt_name, okw, ckw)) in enumerate(PLAN_CASES.items()):
        cls, cfg_cls = _OPTS[opt_name]
        opt = cls(cfg_cls(**okw), PF.NU)
        ctrl = build_reference_controller(opt, ControllerConfig(**ckw))
        rec = _RecordingRandn(1000 + ci)
        np.random.randn = rec
        try:
            x = np.array([0.1, -0.3, 0.0, 0.0])
            for step in range(3):
                ctrl.current_state = x.copy()
                ctrl.time = 0.05 * step
                n_before = len(rec.draws)
                ctrl.update_action()
                out[f"plan_{name}_step{step}_x0"] = x.copy()
                out[f"plan_{name}_step{step}_nominal"] = ctrl.nominal_knots.copy()
                out[f"plan_{name}_step{step}_times"] = ctrl.times.copy()
                out[f"plan_{name}_step{step}_rewards"] = ctrl.rewards.copy()
                out[f"plan_{name}_step{step}_traces"] = ctrl.traces.copy()
                out[f"plan_{name}_step{step}_action"] = ctrl.action(ctrl.time + 0.013).copy()
                for j, d in enumerate(rec.draws[n_before:]):
                    out[f"plan_{name}_step{step}_noise{j}"] = d
                out[f"plan_{name}_step{step}_ndraws"] = np.array(len(rec.draws) - n_before)
                if opt_name == "cem":
                    out[f"plan_{name}_step{step}_sigma"] = opt.sigma.copy()
                # the plant: apply the plan's first control for one control period (two model steps)
                s, _ = PF.rollout_numpy(x, np.tile(ctrl.action(ctrl.time), (1, 2, 1)))
                x = s[0, -1]
        finally:
            np.random.randn = real_randn
    return out


def main() -> None:
    os.makedirs(OUT, exist_ok=True)
    data = gen_traces()
    data.update(gen_plans())
    np.savez_compressed(os.path.join(OUT, "controller.npz"), **data)
    print(f"wrote {len(data)} arrays to {os.path.join(OUT, 'controller.npz')}")


if __name__ == "__main__":
    main()

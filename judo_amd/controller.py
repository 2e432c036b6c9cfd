"""The plan step (`Controller.update_action`, judo/controller/controller.py:210-299) on the GPU.

Host side (numpy, negligible cost, needs the previous plan's spline): time-shift of the nominal knots (:220-221),
normaliser bookkeeping, per-knot sigma, the H x K spline matrix W (cached).  Device side, per optimiser iteration:
ONE fused kernel launch for sample -> clip -> spline -> rollout -> cost (`jh_rollout_cost`), one shard-local reduction,
one all-gather when several GPUs take part, one merge.  Candidates, controls, states and sensors are never
materialised on this path; the elite traces the GUI wants (`update_traces`, :323-363) are recovered by re-rolling
only the E <= max_num_traces best rollouts in materialise mode.
"""

from __future__ import annotations

import warnings
from typing import Any

import numpy as np
import torch

from judo_amd import _lib
from judo_amd.config import ControllerConfig, OptimizerConfig
from judo_amd.device import current_stream_ptr, require_gpu
from judo_amd.distributed import Shard, all_gather_records, shard_rollouts, world_info
from judo_amd.normalization import Normalizer, make_normalizer, normalizer_registry
from judo_amd.optimizers import Optimizer, get_registered_optimizers
from judo_amd.rollout_backend import GpuRolloutBackend
from judo_amd.spline import SPLINE_KINDS, evaluate, spline_weights
from judo_amd.tasks import Task, get_registered_tasks


class Controller:
    """Same attribute/method surface as the reference Controller for the hot path."""

    def __init__(self, controller_config: ControllerConfig, task: Task, optimizer: Optimizer, device: torch.device | None = None,
                 group: Any = None) -> None:
        self._controller_cfg = controller_config
        self.task = task
        self.optimizer = optimizer
        self.device = device if device is not None else require_gpu()
        self.group = group
        self.available_optimizers = get_registered_optimizers()
        self.available_tasks = get_registered_tasks()
        if task.uses_locomotion_policy:  # judo/controller/controller.py:72-80: the policy backend is selected by the task
            from judo_amd.policy import PolicyRolloutBackend
            from judo_amd.spot_tasks import DEFAULT_SPOT_ROLLOUT_CUTOFF_TIME

            world, rank = world_info(group)
            self.model = None
            self.rollout_backend = PolicyRolloutBackend(shard_rollouts(self.optimizer_cfg.num_rollouts, world, rank).count, physics_substeps=task.physics_substeps,
                                                        policy_path=task.locomotion_policy_path, desc=task.desc, device=self.device)
            self.rollout_cutoff_time: float | None = DEFAULT_SPOT_ROLLOUT_CUTOFF_TIME  # policy_mj_rollout_backend.py:94; None = no deadline
        else:
            self.model = task.gpu_model(self.device)
            self.rollout_backend = GpuRolloutBackend(self.model, self.optimizer_cfg.num_rollouts)
        self._last_policy_output: torch.Tensor | None = None  # (shard count, 12) device tensor once a policy rollout has run
        self.system_metadata: dict[str, Any] = {}
        self.current_state = np.concatenate([task.data.qpos, task.data.qvel])
        self.rewards = np.zeros((self.optimizer_cfg.num_rollouts,))
        self.costs_device: torch.Tensor | None = None
        self.traces = None
        self.trace_sensors = [s for s in task.desc["sensors"] if s["type"] == "framepos" and s["name"].startswith("trace")]
        self._w_cache: dict[tuple, torch.Tensor] = {}
        self._lohi_dev: torch.Tensor | None = None
        self.keep_candidates = False
        self.force_materialize = False  # True: always take the materialise path (rollout arrays + Task.reward), e.g. to inspect trajectories
        self.last_rollout = None  # (states, sensors, controls) device tensors of the last materialised iteration
        self.record_kernel_events = False  # bench.py: HIP events around the rollout kernel on the launch stream
        self.kernel_events: list[tuple[torch.cuda.Event, torch.cuda.Event]] = []
        self.candidate_knots_device: torch.Tensor | None = None
        self.reset()

    # ---- config passthrough (mirrors controller.py:109-208) --------------------------------------------------
    @property
    def controller_cfg(self) -> ControllerConfig:
        return self._controller_cfg

    @controller_cfg.setter
    def controller_cfg(self, cfg: ControllerConfig) -> None:
        self._controller_cfg = cfg

    @property
    def optimizer_cfg(self) -> OptimizerConfig:
        return self.optimizer.config

    @property
    def horizon(self) -> float:
        return self.controller_cfg.horizon

    @property
    def nu(self) -> int:
        return self.task.nu

    @property
    def max_opt_iters(self) -> int:
        return self.controller_cfg.max_opt_iters

    @property
    def max_num_traces(self) -> int:
        return self.controller_cfg.max_num_traces

    @property
    def spline_order(self) -> str:
        return self.controller_cfg.spline_order

    @property
    def num_timesteps(self) -> int:
        return int(np.ceil(self.horizon / self.task.dt))

    @property
    def rollout_times(self) -> np.ndarray:
        return self.task.dt * np.arange(self.num_timesteps)

    @property
    def spline_timesteps(self) -> np.ndarray:
        return np.linspace(0, self.horizon, self.optimizer_cfg.num_nodes, endpoint=True)

    @property
    def time(self) -> float:
        return self.task.time

    @time.setter
    def time(self, value: float) -> None:
        self.task.time = value

    # ---- spline ------------------------------------------------------------------------------------------------
    def spline(self, t) -> np.ndarray:
        """Current plan evaluated at time(s) t, holding the first/last knot outside the knot span (:382-401)."""
        t_arr = np.atleast_1d(np.asarray(t, dtype=np.float64))
        out = evaluate(self._spline_kind, self._spline_times, self._spline_knots, t_arr)
        return out[0] if np.ndim(t) == 0 else out

    def update_spline(self, times: np.ndarray, controls: np.ndarray) -> None:
        self._spline_times, self._spline_knots, self._spline_kind = np.array(times, dtype=np.float64), np.array(controls, dtype=np.float64), self.spline_order

    def action(self, time: float) -> np.ndarray:
        return self.spline(time)

    def _fix_num_nodes(self) -> None:
        if self.optimizer_cfg.num_nodes < 4 and self.spline_order == "cubic":
            warnings.warn("Cubic splines require at least 4 nodes. Setting num_nodes=4.", stacklevel=3)
            self.optimizer_cfg.num_nodes = 4

    def reset(self) -> None:
        self.task.reset()
        self._fix_num_nodes()
        self.nominal_knots = np.tile(self.task.optimizer_warm_start(), (self.optimizer_cfg.num_nodes, 1))
        self.times = self.task.data.time + self.spline_timesteps
        self.update_spline(self.times, self.nominal_knots)
        self.current_state = np.concatenate([self.task.data.qpos, self.task.data.qvel])
        self.action_normalizer = self._init_action_normalizer()  # judo/controller/controller.py:208

    def update_states(self, qpos, qvel: np.ndarray | None = None, time: float | None = None, sim_metadata: dict | None = None) -> None:
        """Either the reference's call `update_states(MujocoState)` (judo/controller/controller.py:188-194) or the unpacked fields."""
        if qvel is None and hasattr(qpos, "qpos"):
            qpos, qvel, time, sim_metadata = qpos.qpos, qpos.qvel, qpos.time, qpos.sim_metadata
        self.current_state = np.concatenate([np.asarray(qpos, dtype=np.float64), np.asarray(qvel, dtype=np.float64)])
        self.time = float(time)
        self.system_metadata = sim_metadata or {}

    @property
    def spline_data(self):
        """The plan as the record the plant consumes (judo/controller/controller.py:175-178)."""
        from judo_amd.structs import SplineData

        return SplineData(self._spline_times, self._spline_knots, self._spline_kind)

    # ---- device-side constants ---------------------------------------------------------------------------------
    def _weights(self, K: int, H: int) -> torch.Tensor:
        """W depends only on (kind, K, H, dt, horizon): the knot grid and the query grid shift together with time."""
        key = (self.spline_order, K, H, self.task.dt, self.horizon)
        W = self._w_cache.get(key)
        if W is None:
            W = torch.from_numpy(spline_weights(self.spline_order, self.spline_timesteps, self.rollout_times).astype(np.float32)).to(self.device)
            self._w_cache = {key: W}
        return W

    def _init_action_normalizer(self) -> Normalizer:
        """judo/controller/controller.py:226-239: min_max over the actuator ctrlranges, running statistics, or none."""
        kind = self.controller_cfg.action_normalizer
        if kind == "min_max":
            r = self.task.actuator_ctrlrange
            return make_normalizer("min_max", self.nu, min=r[:, 0], max=r[:, 1])
        if kind in normalizer_registry:
            return make_normalizer(kind, self.nu)
        warnings.warn(f"Invalid action normalizer type {kind!r}. Available types: {list(normalizer_registry)}. Falling back to 'none' normalizer.", stacklevel=3)
        return make_normalizer("none", self.nu)

    def _current_normalizer(self) -> Normalizer:
        """Re-initialised when the configured type changes (controller.py:240-243)."""
        cls = normalizer_registry.get(self.controller_cfg.action_normalizer, normalizer_registry["none"])
        if getattr(self, "action_normalizer", None) is None or type(self.action_normalizer) is not cls:
            self.action_normalizer = self._init_action_normalizer()
        return self.action_normalizer

    # ---- the plan step -------------------------------------------------------------------------------------------
    def update_action(self) -> None:
        lib = _lib.lib()
        opt, task, dev = self.optimizer, self.task, self.device
        if self.current_state.shape != (task.nq + task.nv,):
            raise ValueError(f"current state must have shape ({task.nq + task.nv},), got {self.current_state.shape}")
        if self.optimizer_cfg.num_rollouts <= 0:
            raise ValueError("need at least one rollout")
        self._fix_num_nodes()
        N, K, nu, H = opt.num_rollouts, opt.num_nodes, self.nu, self.num_timesteps
        world, rank = world_info(self.group)
        shard: Shard = shard_rollouts(N, world, rank)

        # time shift (host; needs the previous plan's spline)
        new_times = self.time + self.spline_timesteps
        nominal_knots = self.spline(new_times)
        want_threads = shard.count if task.uses_locomotion_policy else N
        if self.rollout_backend.num_threads != want_threads:  # controller.py:225-229
            self.rollout_backend.update(want_threads)
            self._last_policy_output = None
        nrm = self._current_normalizer()
        nominal_n = nrm.normalize(nominal_knots)  # the optimiser's state lives in normalised units (controller.py:222)
        opt.pre_optimization(self.times, new_times)

        W = self._weights(K, H)
        ctrl_lo, ctrl_hi = task.actuator_ctrlrange[:, 0], task.actuator_ctrlrange[:, 1]
        mom = torch.empty(2 * nu, dtype=torch.float32, device=dev) if nrm.needs_moments else None
        nx, ntp = task.nq + task.nv, len(task.task_params(self.system_metadata))
        host = np.empty(nx + 2 * K * nu + ntp + 2 * nu, dtype=np.float32)
        costs = torch.empty(shard.count, dtype=torch.float32, device=dev)
        scratch = torch.empty(int(lib.jh_update_scratch_floats(shard.count, K, nu)), dtype=torch.float32, device=dev)
        rec = torch.empty(opt.record_floats(), dtype=torch.float32, device=dev)
        out = torch.empty(2 * K * nu, dtype=torch.float32, device=dev)
        knots_out = torch.empty((K, nu, shard.count), dtype=torch.float32, device=dev) if self.keep_candidates else None
        stream = current_stream_ptr()

        i = 0
        while i < self.max_opt_iters and not opt.stop_cond():
            sigma_n = np.asarray(opt.knot_sigma(), dtype=np.float64)  # normalised units; may advance CEM state
            # every shipped normaliser is affine per actuator: raw = center + scale * normalised (judo_amd/normalization.py)
            scale, center = nrm.noise_scale(), nrm.denormalize(np.zeros(nu))
            nominal_knots = nrm.denormalize(nominal_n)
            with np.errstate(invalid="ignore"):
                lohi_np = np.concatenate([nrm.denormalize(nrm.normalize(ctrl_lo)), nrm.denormalize(nrm.normalize(ctrl_hi))])
            lohi_np = np.nan_to_num(np.where(np.isnan(lohi_np), np.concatenate([ctrl_lo, ctrl_hi]), lohi_np).astype(np.float32), posinf=3.0e38, neginf=-3.0e38)
            task.pre_rollout(self.current_state)
            # one small H2D transfer: x0 | nominal | sigma | task params | ctrl bounds
            o = 0
            host[o : o + nx] = self.current_state; o += nx
            host[o : o + K * nu] = nominal_knots.reshape(-1); o += K * nu
            host[o : o + K * nu] = (sigma_n * scale[None, :]).reshape(-1); o += K * nu
            host[o : o + ntp] = task.task_params(self.system_metadata); o += ntp
            host[o : o + 2 * nu] = lohi_np
            blk = torch.from_numpy(host).to(dev)
            x0_d, nom_d, sig_d, tp_d, lohi_d = torch.split(blk, [nx, K * nu, K * nu, ntp, 2 * nu])
            noise = opt.draw_noise(shard.count, shard.offset, dev)
            self._last_sigma_raw, self._last_nominal_before = sigma_n * scale[None, :], nominal_knots.copy()
            if self.record_kernel_events:
                ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                ev0.record()
            if self.uses_fused_cost:
                st = lib.jh_rollout_cost(self.model.handle, _lib.ptr(x0_d), _lib.ptr(nom_d), _lib.ptr(noise), shard.count, _lib.ptr(sig_d), _lib.ptr(W),
                                         _lib.ptr(lohi_d), _lib.ptr(tp_d), int(task.phase), shard.count, shard.offset, H, K, _lib.ptr(costs),
                                         _lib.ptr(knots_out), stream)
                _lib.check(st, "jh_rollout_cost")
            else:
                costs = self._materialised_costs(x0_d, nom_d, noise, sig_d, lohi_d, W, shard, H, K, stream)
            if self.record_kernel_events:
                ev1.record()
                self.kernel_events.append((ev0, ev1))
            opt.device_partial(costs, None, nom_d, noise, sig_d, lohi_d, shard.count, shard.offset, scratch, rec)
            recs = all_gather_records(rec, self.group)
            opt.device_merge(recs, world, out[: K * nu], out[K * nu :], clip_sigma=False)
            res = out.cpu().numpy().astype(np.float64)  # the only sync of the iteration
            nominal_n = (res[: K * nu].reshape(K, nu) - center[None, :]) / scale[None, :]  # the update acted on the normalised candidates
            if hasattr(opt, "sigma") and isinstance(getattr(opt, "sigma"), np.ndarray):  # CEM: refit in normalised units
                opt.sigma = np.clip(res[K * nu :].reshape(K, nu) / scale[None, :], opt.sigma_min, opt.sigma_max)
            if nrm.needs_moments:  # running statistics over this iteration's raw candidates, all ranks (controller.py:290-291)
                ctr = torch.from_numpy(np.asarray(nrm.mean, dtype=np.float32)).to(dev)
                st = lib.jh_knot_moments(None, _lib.ptr(nom_d), _lib.ptr(noise), shard.count, _lib.ptr(sig_d), _lib.ptr(lohi_d), _lib.ptr(ctr),
                                         shard.count, shard.offset, K, nu, _lib.ptr(mom), stream)
                _lib.check(st, "jh_knot_moments")
                m = all_gather_records(mom, self.group).cpu().numpy().astype(np.float64).reshape(world, 2, nu).sum(0)
                # the kernel centred on the fp32 image of the mean: shift the moments to the fp64 mean
                dm = np.asarray(nrm.mean, dtype=np.float32).astype(np.float64) - nrm.mean
                cnt = N * K
                s1 = m[0] + cnt * dm
                s2 = m[1] + 2 * dm * m[0] + cnt * dm * dm
                nrm.update_from_moments(cnt, s1, s2)
            i += 1

        self.costs_device = costs
        self.candidate_knots_device = knots_out
        self.last_shard = shard
        self.nominal_knots = nrm.denormalize(nominal_n)  # with the statistics as updated in the loop (controller.py:296)
        self.times = new_times
        self.update_spline(self.times, self.nominal_knots)

    @property
    def uses_fused_cost(self) -> bool:
        """True when the task's cost is the one fused into the rollout kernel.  A task that overrides `Task.reward` (a plugin
        registered through `register_task` with its own reward on one of the shipped models) is served by the materialise path:
        candidate controls -> full state/sensor trajectories -> the task's own `reward`, all on device arrays."""
        return not self.force_materialize and type(self.task).reward is Task.reward

    def _materialised_costs(self, x0_d, nom_d, noise, sig_d, lohi_d, W, shard: Shard, H: int, K: int, stream) -> torch.Tensor:
        """judo/controller/controller.py:239-262 on the device: candidate splines at the rollout times, RolloutBackend.rollout,
        Task.reward, Task.post_rollout.  Returns costs = -rewards (fp32, device)."""
        lib, task, nu = _lib.lib(), self.task, self.nu
        controls = torch.empty((shard.count, H, nu), dtype=torch.float32, device=self.device)
        st = lib.jh_spline_controls(_lib.ptr(W), None, _lib.ptr(nom_d), _lib.ptr(noise), shard.count, _lib.ptr(sig_d), _lib.ptr(lohi_d), shard.count,
                                    shard.offset, H, K, nu, _lib.ptr(controls), stream)
        _lib.check(st, "jh_spline_controls")
        if task.uses_locomotion_policy:  # controller.py:265-273: commands -> policy + plant; the policy outputs carry over to the next plan step
            if self._last_policy_output is None:
                self._last_policy_output = torch.zeros((shard.count, 12), dtype=torch.float32, device=self.device)
            states, sensors, self._last_policy_output = self.rollout_backend.rollout(x0_d, task.task_to_sim_ctrl(controls), self._last_policy_output,
                                                                                    cutoff_time=self.rollout_cutoff_time)
        else:
            states, sensors = self.rollout_backend.rollout_device(x0_d, controls)
        if getattr(task, "reward_accepts_torch", True):
            args = (states, sensors, controls)
        else:  # numpy-only plugin reward: one host round trip of the trajectories
            args = tuple(a.cpu().numpy().astype(np.float64) for a in (states, sensors, controls))
        rewards = task.reward(*args, self.system_metadata)
        task.post_rollout(*args, self.system_metadata)
        rewards = torch.as_tensor(rewards, device=self.device).to(torch.float32).reshape(-1)
        if rewards.shape[0] != shard.count:
            raise ValueError(f"Task.reward must return ({shard.count},) rewards, got {tuple(rewards.shape)}")
        self.last_rollout = (states, sensors, controls)
        return (-rewards).contiguous()

    @property
    def rewards_local(self) -> np.ndarray:
        """Rewards of this rank's shard of the last iteration (host copy on demand)."""
        return -self.costs_device.cpu().numpy().astype(np.float64)

    # ---- traces --------------------------------------------------------------------------------------------------
    def update_traces(self) -> None:
        """Line segments of the best rollouts' `trace*` framepos sensors, best rollout first (controller.py:323-363):
        shape (E * n_trace_sensors * (H-1), 2, 3).  Re-rolls only the E elite rollouts in materialise mode."""
        if self.costs_device is None or self.optimizer.last_noise is None:
            raise RuntimeError("update_traces() needs a completed update_action()")
        if not self.trace_sensors:
            self.traces = np.zeros((0, 2, 3))
            return
        if self.task.uses_locomotion_policy:  # the materialise path kept every rollout's sensors: pick the elites' rows, no re-rollout
            sensors = self.last_rollout[1]
            E = min(self.max_num_traces, int(self.costs_device.numel()))
            order = torch.argsort(self.costs_device, stable=True)[:E]
            sel = sensors[order].cpu().numpy().astype(np.float64)
            segs = [np.stack([sel[e, :-1, s["adr"] : s["adr"] + 3], sel[e, 1:, s["adr"] : s["adr"] + 3]], axis=1) for e in range(E) for s in self.trace_sensors]
            self.traces = np.concatenate(segs, axis=0)
            return
        self.traces = elite_traces(self, self._last_sigma_raw, self._last_nominal_before)


def elite_traces(controller: Controller, sigma_raw: np.ndarray, nominal_before: np.ndarray) -> np.ndarray:
    """Trace segments for the E best rollouts of the last plan step.

    nominal_before / sigma_raw are the nominal knots and raw-unit sigma that the step sampled around."""
    opt, task = controller.optimizer, controller.task
    shard = controller.last_shard
    K, nu, H = opt.num_nodes, controller.nu, controller.num_timesteps
    costs = controller.costs_device
    E = min(controller.max_num_traces, int(costs.numel()))
    order = torch.argsort(costs, stable=True)[:E]
    eps = opt.last_noise[:, :, order].permute(2, 0, 1).cpu().numpy().astype(np.float64)  # (E, K, nu)
    gidx = order.cpu().numpy() + shard.offset
    knots = nominal_before[None] + sigma_raw[None] * eps
    knots[gidx == 0] = nominal_before
    r = task.actuator_ctrlrange
    knots = np.clip(knots, r[:, 0], r[:, 1])
    U = evaluate(controller.spline_order, controller.times, knots, controller.times[0] + controller.rollout_times)
    _, sensors, _ = controller.rollout_backend.rollout(controller.current_state, U)
    segs = []
    for e in range(E):
        for s in controller.trace_sensors:
            p = sensors[e, :, s["adr"] : s["adr"] + 3]
            segs.append(np.stack([p[:-1], p[1:]], axis=1))
    return np.concatenate(segs, axis=0) if segs else np.zeros((0, 2, 3))


def make_controller(init_task: str, init_optimizer: str, device: torch.device | None = None, group: Any = None) -> Controller:
    """judo/controller/controller.py:404-442: instantiate task, optimizer (with per-task overrides) and controller."""
    tasks, opts = get_registered_tasks(), get_registered_optimizers()
    if init_task not in tasks:
        raise ValueError(f"Task {init_task} not found in task registry.")
    if init_optimizer not in opts:
        raise ValueError(f"Optimizer {init_optimizer} not found in optimizer registry.")
    task = tasks[init_task][0]()
    opt_cls, opt_cfg_cls = opts[init_optimizer]
    opt_cfg = opt_cfg_cls()
    opt_cfg.set_override(init_task)
    optimizer = opt_cls(opt_cfg, task.nu)
    ctrl_cfg = ControllerConfig()
    ctrl_cfg.set_override(init_task)
    return Controller(ctrl_cfg, task, optimizer, device=device, group=group)


assert set(SPLINE_KINDS) == {"zero", "linear", "cubic"}

"""The plan step (`Controller.update_action`, judo/controller/controller.py:210-299) on the GPU.

Host side (numpy, negligible cost, needs the previous plan's spline): time-shift of the nominal knots (:220-221),
normaliser bookkeeping, per-knot sigma, the H x K spline matrix W (cached).  Device side, per optimiser iteration:
ONE fused kernel launch for sample -> clip -> spline -> rollout -> cost (`jh_rollout_cost`), one shard-local reduction,
one all-gather when several GPUs take part, one merge.  Candidates, controls, states and sensors are never
materialised on this path; the elite traces the GUI wants (`update_traces`, :323-363) are recovered by re-rolling
only the E <= max_num_traces best rollouts in materialise mode, when somebody reads `Controller.traces`.

Three ways through one optimiser iteration, chosen per plan step:
  fused        a `FusedOptimizer` (GpuMPPI / GpuCEM / GpuPS) on a task whose cost is the one compiled into the rollout kernel;
  materialise  a `FusedOptimizer` on a task that brings its own `Task.reward` (plugin), a policy task (Spot), or a knot count
               the fused kernel does not hold in registers: candidate controls -> `RolloutBackend.rollout` -> `Task.reward`,
               all on device arrays, then the same update kernels;
  candidates   any other `Optimizer` (a plugin with the reference's two numpy methods): its candidates are uploaded, rolled
               out and scored on the device, the rewards go back to its `update_nominal_knots`.

All per-step device and pinned-host buffers live in a `_PlanBuffers` cache keyed by the problem size: a plan step allocates
nothing, does one asynchronous H2D copy of a < 2 KB block, one asynchronous D2H copy of the new nominal and one stream wait.
"""

from __future__ import annotations

import ctypes as C
import warnings
from typing import Any

import numpy as np
import torch

from judo_amd import _lib
from judo_amd.config import ControllerConfig, OptimizerConfig
from judo_amd.device import HipEvent, current_stream_ptr, require_gpu
from judo_amd.distributed import Shard, all_gather_records, shard_rollouts, world_info
from judo_amd.normalization import Normalizer, make_normalizer, normalizer_registry
from judo_amd.optimizers import FusedOptimizer, Optimizer, get_registered_optimizers
from judo_amd.rollout_backend import GpuRolloutBackend
from judo_amd.spline import SPLINE_KINDS, evaluate, spline_weights
from judo_amd.tasks import Task, get_registered_tasks

POLICY_OUTPUT_DIM = 12  # judo/tasks/spot/spot_constants.py


class _PlanBuffers:
    """Everything a plan step touches on the device, allocated once per (shard size, K, nu, nx, task params, record size)."""

    def __init__(self, dev: torch.device, n_local: int, K: int, nu: int, nx: int, ntp: int, rec_floats: int, trace_k: int) -> None:
        L = _lib.lib()
        self.sizes = [nx, K * nu, K * nu, ntp, 2 * nu]
        nblk = sum(self.sizes)
        self.offsets = [int(v) for v in np.cumsum([0] + self.sizes)]
        self.host = torch.empty(nblk, dtype=torch.float32).pin_memory()
        self.host_np = self.host.numpy()
        self.host_ptr, self.nblk_bytes = self.host.data_ptr(), 4 * nblk
        self.blk = torch.empty(nblk, dtype=torch.float32, device=dev)
        self.x0, self.nominal, self.sigma, self.tp, self.lohi = torch.split(self.blk, self.sizes)
        self.costs = torch.empty(n_local, dtype=torch.float32, device=dev)
        self.scratch = torch.empty(int(L.jh_update_scratch_floats(n_local, K, nu)), dtype=torch.float32, device=dev)
        self.rec = torch.empty(max(rec_floats, 1), dtype=torch.float32, device=dev)
        self.fused_scratch = torch.zeros(int(L.jh_update_fused_scratch_floats(n_local, K, nu)), dtype=torch.float32, device=dev)  # (zero: it holds jh_update_fused's ticket counter)
        self.dev = dev
        self.done = torch.zeros(4, dtype=torch.int32).pin_memory()  # completion word of jh_plan_step: the update's last workgroup sets it behind the results, jh_download_end polls it
        self.done_ptr = self.done.data_ptr()
        self.size_out(2 * K * nu)
        self.shard_rec, self.shard_all = None, None  # several ranks: this rank's record of the plan step, and the all-gathered records
        self.trace_buf = None   # (n_local * H * trace floats) when the fused kernel writes the trace sensors
        self.trace_rows = None  # elites' records [cost, index, trace row]
        self.trace_recs = [torch.full((max(trace_k, 1) * (2 + K * nu),), float("inf"), dtype=torch.float32, device=dev) for _ in range(2)]  # alternated per plan step
        self.trace_flip = 0
        self.knots_out: torch.Tensor | None = None
        self.knots_nku: torch.Tensor | None = None
        self.mom: torch.Tensor | None = None


    def size_out(self, n: int) -> None:
        """The output block (nominal | sigma | trace elites' records) and its pinned host image hold at least n floats."""
        if getattr(self, "out", None) is not None and self.out.numel() >= n:
            return
        self.out = torch.zeros(n, dtype=torch.float32, device=self.dev)
        self.out_host = torch.zeros(n, dtype=torch.float32).pin_memory()  # (zeros: MPPI / PS never write the sigma region, and a cast of uninitialised bits may meet a signalling NaN)
        self.out_np = self.out_host.numpy()
        self.out_host_ptr = self.out_host.data_ptr()


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
        self._rewards: np.ndarray | None = np.zeros((self.optimizer_cfg.num_rollouts,))
        self.costs_device: torch.Tensor | None = None
        self._last_fused: dict | None = None
        self.trace_sensors = [s for s in task.desc["sensors"] if s["type"] == "framepos" and "trace" in s["name"]]  # visualizers/utils.py:169-178
        self.fused_traces = True  # let the fused kernel write every rollout's trace sensors (include/judo_amd.h, jh_rollout_cost_traced); False: re-roll the elites
        self._traces: np.ndarray | None = None
        self._trace_stage: dict | None = None
        self._w_cache: dict[tuple, torch.Tensor] = {}
        self._shift_cache: tuple = (None, None)
        self._grid_cache: tuple = (None, None)
        self._lohi_cache: tuple = (None, None)
        self._bufs: _PlanBuffers | None = None
        self._bufs_key: tuple | None = None
        self._noise_bufs: list[torch.Tensor] | None = None
        self._noise_cur = 0
        self._noise_ahead = None
        self._side_stream = None  # several ranks: the next iteration's noise draw runs here while the update records are all-gathered
        self._side_events, self._side_flip = None, 0
        self.noise_events: list = []  # (start, end) of the side-stream noise draws when record_kernel_events is set
        self.prefetch_noise = True  # draw the next iteration's noise behind this iteration's download
        self.zero_copy_out = True  # jh_update_fused writes nominal | sigma | trace records into the pinned host block itself (no download command)
        self.force_shard_path = False  # tests: take launch -> all-gather -> merge with ONE rank as well (the RCCL branch of the exchange on a one-GPU box)
        self.poll_completion = True  # one GPU, closed-form models (plan steps of ~0.1 ms): wait for the completion word the update's last workgroup writes behind its results instead of the stream's event (jh_plan_step, out_host_mark); the articulated models' 7-50 ms plan steps keep the event
        self.host_block_in_place = True  # closed-form models: the plan-step kernel reads x0 | nominal | sigma | task params | bounds from the pinned host block (no copy in front of the launch)
        self.fused_update = True  # one GPU: the whole update (block partials, merge, trace elites) in one launch and one download (jh_update_fused); False: the separate kernels
        self._prefetch_args = None
        self.keep_candidates = False
        self.exchange_events: list = []  # (start, end) HIP event pairs around the record all-gather + merge of every iteration while `record_kernel_events` is on
        self.force_materialize = False  # True: always take the materialise path (rollout arrays + Task.reward), e.g. to inspect trajectories
        self.last_rollout = None  # (states, sensors, controls) device tensors of the last materialised iteration
        self.record_kernel_events = False  # bench.py: HIP events around the rollout kernel on the launch stream
        self.kernel_events: list[tuple[torch.cuda.Event, torch.cuda.Event]] = []
        self.candidate_knots_device: torch.Tensor | None = None
        self._candidate_knots: np.ndarray | None = None
        self.solver_warnings = True  # warn when the articulated-body kernels dropped contacts or ran into the Newton cap (checked in `solver_stats`)
        self.reset()

    # ---- config passthrough (mirrors controller.py:109-208) --------------------------------------------------
    @property
    def controller_cfg(self) -> ControllerConfig:
        return self._controller_cfg

    @controller_cfg.setter
    def controller_cfg(self, cfg: ControllerConfig) -> None:
        self._controller_cfg = cfg
        self.action_normalizer = self._init_action_normalizer()  # controller.py:205-208

    @property
    def optimizer_cfg(self) -> OptimizerConfig:
        return self.optimizer.config

    @optimizer_cfg.setter
    def optimizer_cfg(self, cfg: OptimizerConfig) -> None:  # controller.py:164-167 (the GUI swaps whole config objects in)
        self.optimizer.config = cfg

    @property
    def optimizer_cls(self) -> type:  # controller.py:169-177
        return type(self.optimizer)

    @property
    def optimizer_config_cls(self) -> type:
        return type(self.optimizer.config)

    @property
    def task_config(self):  # controller.py:179-187
        return self.task.config

    @task_config.setter
    def task_config(self, cfg) -> None:
        self.task.config = cfg

    @property
    def action_normalizer_type(self) -> str:  # controller.py:139-142
        return self.controller_cfg.action_normalizer

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
        key = (self.horizon, self.task.dt)
        c = getattr(self, "_nts_cache", None)
        if c is None or c[0] != key:  # (read several times per plan step: a small plan step is 0.1 ms)
            c = self._nts_cache = (key, int(np.ceil(self.horizon / self.task.dt)))
        return c[1]

    @property
    def rollout_times(self) -> np.ndarray:
        return self.task.dt * np.arange(self.num_timesteps)

    @property
    def spline_timesteps(self) -> np.ndarray:
        key = (self.horizon, self.optimizer_cfg.num_nodes)
        if self._grid_cache[0] != key:
            g = np.linspace(0, key[0], key[1], endpoint=True)
            g.setflags(write=False)
            self._grid_cache = (key, g)
        return self._grid_cache[1]

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

    def _shifted_nominal(self, new_times: np.ndarray) -> np.ndarray:
        """`nominal = prev_spline(new_times)` (controller.py:220-221).  The weights depend only on the spline kind and on the two knot grids relative to
        each other, which repeat from plan step to plan step at a fixed control period: cached on those (to 1e-9 s)."""
        t = self._spline_times
        key = (self._spline_kind, len(t), len(new_times), round(float(t[-1] - t[0]), 9), round(float(new_times[0] - t[0]), 9), round(float(new_times[-1] - new_times[0]), 9))
        if self._shift_cache[0] != key:
            uniform = np.allclose(np.diff(t), (t[-1] - t[0]) / (len(t) - 1), rtol=0, atol=1e-12)
            if not uniform:
                return evaluate(self._spline_kind, t, self._spline_knots, new_times)
            self._shift_cache = (key, spline_weights(self._spline_kind, t - t[0], new_times - t[0]))
        return self._shift_cache[1] @ self._spline_knots

    def action(self, time: float) -> np.ndarray:
        return self.spline(time)

    def _fix_num_nodes(self) -> None:
        if self.optimizer_cfg.num_nodes < 4 and self.spline_order == "cubic":
            warnings.warn("Cubic splines require at least 4 nodes. Setting num_nodes=4.", stacklevel=3)
            self.optimizer_cfg.num_nodes = 4

    def reset(self) -> None:
        """judo/controller/controller.py:306-321."""
        self.task.reset()
        self._fix_num_nodes()
        self.nominal_knots = np.tile(self.task.optimizer_warm_start(), (self.optimizer_cfg.num_nodes, 1))
        self._candidate_knots, self._last_fused = None, None  # `candidate_knots` reads as tile(nominal) until a plan step has run (controller.py:313)
        self.times = self.task.data.time + self.spline_timesteps
        self.update_spline(self.times, self.nominal_knots)
        self.current_state = np.concatenate([self.task.data.qpos, self.task.data.qvel])
        self._current_normalizer()  # (re-)initialised when missing or when the configured type changed
        if self.task.uses_locomotion_policy:  # :318-321: the policy starts the new episode from zero outputs; so does the plant solver's warm start
            self._last_policy_output = None  # re-created as zeros by the next rollout
            self.rollout_backend.update(self.rollout_backend.num_threads)
        self._traces, self._trace_stage = None, None

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

    def _buffers(self, n_local: int, K: int, nu: int, nx: int, ntp: int, rec_floats: int, trace_k: int) -> _PlanBuffers:
        key = (n_local, K, nu, nx, ntp, rec_floats, trace_k)
        if self._bufs_key != key:
            self._bufs, self._bufs_key = _PlanBuffers(self.device, *key), key
        return self._bufs

    def _noise_buffer(self, shape: tuple[int, int, int]) -> torch.Tensor:
        """The next of two persistent (K, nu, N) noise buffers (the other one still backs the lazy `rewards` / `candidate_knots` of the last plan step)."""
        if self._noise_bufs is None or tuple(self._noise_bufs[0].shape) != shape:
            self._noise_bufs = [torch.empty(shape, dtype=torch.float32, device=self.device) for _ in range(2)]
            self._noise_ahead = None
        self._noise_cur ^= 1
        return self._noise_bufs[self._noise_cur]

    def reserve_timing_events(self, n: int) -> None:
        """Create n timing events ahead of a measured region (`record_kernel_events`): creating them inside it costs a small plan step several microseconds each."""
        self._event_pool = [HipEvent() for _ in range(n)]

    def _timing_event(self) -> HipEvent:
        pool = getattr(self, "_event_pool", None)
        return pool.pop() if pool else HipEvent()

    def _draw_noise(self, n_local: int, n_offset: int) -> torch.Tensor:
        """The optimizer's noise for this shard, generated into a persistent (K, nu, n_local) buffer when it comes from the device noise stream -- or taken from
        the draw `_prefetch_noise` enqueued behind the last iteration's download (same stream, same draw number: the same numbers, earlier)."""
        opt = self.optimizer
        if opt.injected_noise is None:
            K, nu = opt.num_nodes, self.nu
            ahead, self._noise_ahead = self._noise_ahead, None
            if ahead is not None and ahead[0] == (K, nu, n_local, n_offset) and ahead[1] is opt._generator and ahead[1] is not None:
                if ahead[3] is not None:  # drawn on the side stream: the launch stream waits for it (device side; the host does not)
                    _lib.check(_lib.lib().jh_stream_wait_event(current_stream_ptr(), ahead[3].handle), "jh_stream_wait_event")
                opt.last_noise = ahead[2]
                return ahead[2]
            if ahead is not None and ahead[3] is not None:  # a side-stream draw that is not used after all (reseed, new shape): its buffer may be the one drawn into next
                _lib.check(_lib.lib().jh_stream_wait_event(current_stream_ptr(), ahead[3].handle), "jh_stream_wait_event")
            return opt.draw_noise(n_local, n_offset, self.device, out=self._noise_buffer((K, nu, n_local)))
        self._noise_ahead = None
        return opt.draw_noise(n_local, n_offset, self.device)

    def _prefetch_noise(self, n_local: int, n_offset: int, side: bool = False) -> None:
        """Enqueue the next iteration's noise draw now (behind the download of this one's result): it leaves the critical path of the next plan step.
        Only for the library's own `draw_noise` with the device generator; a reseed, an injected noise array or a changed shape discards the draw.
        `side` (several ranks): the draw goes on a second stream, enqueued BEFORE the all-gather of the update records -- the collective (RCCL's own stream, ordered
        behind the launch stream by events) and the merge behind it do not wait for it, it runs while the records travel; the next plan step's rollout
        kernel waits for its event.  The buffer it writes was last read by the plan step before the current one, which the host has waited for."""
        opt = self.optimizer
        if not self.prefetch_noise or opt.injected_noise is not None or type(opt).draw_noise is not Optimizer.draw_noise or opt._generator is None:
            return
        K, nu = opt.num_nodes, self.nu
        keep = opt.last_noise
        ready = None
        if side:
            if self._side_stream is None:
                self._side_stream = torch.cuda.Stream(device=self.device)
            with torch.cuda.stream(self._side_stream):
                t0 = None
                if self.record_kernel_events:
                    t0 = self._timing_event(); t0.record()
                noise = opt.draw_noise(n_local, n_offset, self.device, out=self._noise_buffer((K, nu, n_local)))
                if t0 is not None:
                    ready = self._timing_event()
                    self.noise_events.append((t0, ready))
                else:  # two events, alternated: the one of the draw before this one was waited for by the plan step now in flight
                    if self._side_events is None:
                        self._side_events = [HipEvent(), HipEvent()]
                    self._side_flip ^= 1
                    ready = self._side_events[self._side_flip]
                ready.record()
        else:
            noise = opt.draw_noise(n_local, n_offset, self.device, out=self._noise_buffer((K, nu, n_local)))
        opt.last_noise = keep
        self._noise_ahead = ((K, nu, n_local, n_offset), opt._generator, noise, ready)

    # ---- the plan step -------------------------------------------------------------------------------------------
    @property
    def uses_fused_optimizer(self) -> bool:
        return isinstance(self.optimizer, FusedOptimizer)

    @property
    def uses_fused_cost(self) -> bool:
        """True when the task's cost is the one fused into the rollout kernel.  A task that overrides `Task.reward` (a plugin
        registered through `register_task` with its own reward on one of the shipped models) is served by the materialise path:
        candidate controls -> full state/sensor trajectories -> the task's own `reward`, all on device arrays.  So is a knot count
        above what the fused kernel of this model holds in registers (a live `num_nodes` edit must not kill the control loop)."""
        if self.force_materialize or type(self.task).reward is not Task.reward or self.model is None or not self.uses_fused_optimizer:
            return False
        # the fused kernel neither maps controls through `task_to_sim_ctrl` nor hands trajectories to `post_rollout` (controller.py:262-277): a plugin
        # task that keeps the shipped reward but overrides either hook goes through the materialise path, where both are called
        if type(self.task).post_rollout is not Task.post_rollout or type(self.task).task_to_sim_ctrl is not Task.task_to_sim_ctrl:
            return False
        return self.optimizer.num_nodes <= self.model.max_fused_knots_at(self.num_timesteps)

    def update_action(self) -> None:
        lib = _lib.lib()
        opt, task, dev = self.optimizer, self.task, self.device
        if self.current_state.shape != (task.nq + task.nv,):
            raise ValueError(f"current state must have shape ({task.nq + task.nv},), got {self.current_state.shape}")
        if self.optimizer_cfg.num_rollouts <= 0:
            raise ValueError("need at least one rollout")
        self._fix_num_nodes()
        N, K, nu, H = opt.num_rollouts, opt.num_nodes, self.nu, self.num_timesteps
        if K * nu > _lib.MAX_KNOT_DIM:
            raise ValueError(f"num_nodes * nu = {K * nu} exceeds the update kernels' limit of {_lib.MAX_KNOT_DIM} (include/judo_amd.h JH_MAX_KNOT_DIM)")
        world, rank = world_info(self.group)
        shard: Shard = shard_rollouts(N, world, rank)
        self._prefetch_args = None

        # time shift (host; needs the previous plan's spline)
        new_times = self.time + self.spline_timesteps
        nominal_knots = self._shifted_nominal(new_times)
        want_threads = shard.count if task.uses_locomotion_policy else N
        if self.rollout_backend.num_threads != want_threads:  # controller.py:225-229
            self.rollout_backend.update(want_threads)
            self._last_policy_output = None
        nrm = self._current_normalizer()
        nominal_n = nrm.normalize(nominal_knots)  # the optimiser's state lives in normalised units (controller.py:222)
        opt.pre_optimization(self.times, new_times)

        W = self._weights(K, H)
        x0 = np.array(self.current_state, dtype=np.float64)
        fused_opt = self.uses_fused_optimizer
        E = self._num_trace_elites(N)
        tp0 = task.task_params(self.system_metadata)
        b = self._buffers(shard.count, K, nu, task.nq + task.nv, len(tp0), opt.record_floats() if fused_opt else 0, E)
        stream = self._stream = current_stream_ptr()
        state: dict[str, Any] = dict(E=E, x0=x0, new_times=new_times)
        staged = False

        i = 0
        while i < self.max_opt_iters and not opt.stop_cond():
            task.pre_rollout(self.current_state)
            if fused_opt:
                # the trace records of the (presumably) last iteration are enqueued before its result is waited for: their host cost hides behind the kernels
                state["stage"] = (lambda: self._stage_traces(lib, b, state, shard, world, E, x0, new_times, K, nu, stream)) if i == self.max_opt_iters - 1 else None
                nominal_n = self._fused_iteration(lib, b, nrm, nominal_n, W, shard, world, H, K, nu, N, stream, state)
                staged = state["stage"] is not None
            else:
                nominal_n = self._candidates_iteration(lib, b, nrm, nominal_n, W, shard, H, K, nu, N, stream, state)
            i += 1

        if i > 0:
            self.costs_device = state["costs"]
            self.candidate_knots_device = state.get("knots_out")
            if not staged:
                self._stage_traces(lib, b, state, shard, world, E, x0, new_times, K, nu, stream)
        self.last_shard = shard
        self.nominal_knots = nrm.denormalize(nominal_n)  # with the statistics as updated in the loop (controller.py:296)
        self.times = new_times
        self.update_spline(self.times, self.nominal_knots)

    def _pack_block(self, b: _PlanBuffers, nominal_raw: np.ndarray, sigma_raw: np.ndarray | None, lohi: np.ndarray, upload: bool = True) -> None:
        """x0 | nominal | sigma | task params | ctrl bounds -> pinned host block -> device, one asynchronous copy."""
        h, o = b.host_np, 0
        n = b.sizes[0]; h[o : o + n] = self.current_state; o += n
        n = b.sizes[1]; h[o : o + n] = nominal_raw.reshape(-1); o += n
        n = b.sizes[2]; h[o : o + n] = 0.0 if sigma_raw is None else sigma_raw.reshape(-1); o += n
        n = b.sizes[3]; h[o : o + n] = self.task.task_params(self.system_metadata); o += n
        h[o:] = lohi
        b.blk_stale = not upload
        if upload:
            _lib.check(_lib.lib().jh_upload_async(b.blk.data_ptr(), b.host_ptr, b.nblk_bytes, self._stream), "jh_upload_async")

    def _ensure_block(self, b: _PlanBuffers) -> None:
        """The device copy of the plan block holds what the pinned host block holds.  A plan step whose kernel read the host block in place (`host_block_in_place`)
        leaves the device copy behind; the few consumers of the device views outside the plan step (`candidate_knots`, the knot-record form of the trace stage) bring it up first."""
        if getattr(b, "blk_stale", False):
            _lib.check(_lib.lib().jh_upload_async(b.blk.data_ptr(), b.host_ptr, b.nblk_bytes, current_stream_ptr()), "jh_upload_async")
            b.blk_stale = False

    def _raw_bounds(self, nrm: Normalizer) -> np.ndarray:
        r = self.task.actuator_ctrlrange
        ctrl_lo, ctrl_hi = r[:, 0], r[:, 1]
        if type(nrm).__name__ == "IdentityNormalizer":
            if self._lohi_cache[0] is r:
                return self._lohi_cache[1]
            lohi = np.concatenate([ctrl_lo, ctrl_hi])
            out = np.nan_to_num(lohi.astype(np.float32), posinf=3.0e38, neginf=-3.0e38)
            self._lohi_cache = (r, out)
            return out
        else:
            with np.errstate(invalid="ignore"):
                lohi = np.concatenate([nrm.denormalize(nrm.normalize(ctrl_lo)), nrm.denormalize(nrm.normalize(ctrl_hi))])
            lohi = np.where(np.isnan(lohi), np.concatenate([ctrl_lo, ctrl_hi]), lohi)
        return np.nan_to_num(lohi.astype(np.float32), posinf=3.0e38, neginf=-3.0e38)

    def _fetch(self, b: _PlanBuffers, n: int, behind=None, in_place: bool = False, n_float: int | None = None) -> np.ndarray:
        """Device result -> pinned host memory, one wait: the only synchronisation of an iteration.  `behind` enqueues work that may run after the
        copy (the trace records): it is launched while the copy is in flight and is not waited for.  `in_place`: the update kernel wrote the pinned host
        block itself (jh_update_fused with host pointers): no copy, the completion mark alone."""
        L = _lib.lib()
        _lib.check(L.jh_download_begin(b.out_host_ptr, b.out.data_ptr(), 0 if in_place else 4 * n, self._stream), "jh_download_begin")
        try:  # (the mark set by `begin` must be consumed whatever happens in between: `end` pops the oldest mark of this thread)
            if behind is not None:
                behind()
            if self._prefetch_args is not None:
                self._prefetch_noise(*self._prefetch_args)
        finally:
            _lib.check(L.jh_download_end(), "jh_download_end")
        # only nominal | sigma are floats to be widened: the trace records behind them carry an index column of int bit patterns (they are consumed as the raw fp32 copy)
        return b.out_np[: n if n_float is None else n_float].astype(np.float64)

    def _fused_iteration(self, lib, b: _PlanBuffers, nrm: Normalizer, nominal_n: np.ndarray, W, shard: Shard, world: int, H: int, K: int, nu: int, N: int,
                         stream, state: dict) -> np.ndarray:
        opt, task = self.optimizer, self.task
        sigma_n = np.asarray(opt.knot_sigma(), dtype=np.float64)  # normalised units; may advance CEM state
        # every shipped normaliser is affine per actuator: raw = center + scale * normalised (judo_amd/normalization.py)
        scale, center = nrm.noise_scale(), nrm.denormalize(np.zeros(nu))
        nominal_raw = nrm.denormalize(nominal_n)
        sigma_raw = sigma_n * scale[None, :]
        fused_cost = self.uses_fused_cost
        one_call = world == 1 and not self.force_shard_path and fused_cost and self.fused_update and self.zero_copy_out and hasattr(opt, "fused_update_args") and not nrm.needs_moments
        # several ranks, same conditions: the same shape -- launch (rollout + this rank's record) -> ONE all-gather -> merge -- instead of the chain of separate partial /
        # gather / merge / trace-gather launches with two collectives, which remains for the cases below (plugin costs, running normaliser statistics)
        shard_call = (world > 1 or self.force_shard_path) and fused_cost and self.fused_update and self.zero_copy_out and hasattr(opt, "fused_update_args") and not nrm.needs_moments
        self._pack_block(b, nominal_raw, sigma_raw, self._raw_bounds(nrm), upload=not (one_call or shard_call))
        noise = self._draw_noise(shard.count, shard.offset)  # (K, nu, shard.count), possibly a view into the full draw
        self._prefetch_args = (shard.count, shard.offset)
        ldn, noise_p = int(noise.stride(1)), noise.data_ptr()
        self._last_sigma_raw, self._last_nominal_before = sigma_raw, nominal_raw.copy()
        knots_out = None
        if self.keep_candidates:
            # the rollout kernels write candidate (k, u) of local rollout n at [(k * nu + u) * ldn + n] with the NOISE's row stride (include/judo_amd.h):
            # a sharded draw is a column view of the full (K, nu, total) draw, so the buffer gets rows of that length and the shard's columns are handed out
            if b.knots_out is None or b.knots_out.shape[2] != ldn:
                b.knots_out = torch.empty((K, nu, ldn), dtype=torch.float32, device=self.device)
            knots_out = b.knots_out[:, :, : shard.count]
        state["trace_buf"] = None
        is_cem = hasattr(opt, "sigma") and isinstance(getattr(opt, "sigma"), np.ndarray)
        if one_call or shard_call:
            # ---- one GPU, shipped cost: the whole iteration is ONE library call (jh_plan_step: upload, rollout + cost kernel, one-launch update with the trace elites'
            # records, results written straight into the pinned host block) and one wait
            nfl = self._fused_trace_floats()
            staging = state.get("stage") is not None
            if nfl and (b.trace_buf is None or b.trace_buf.numel() != shard.count * H * nfl):
                b.trace_buf = torch.empty(shard.count * H * nfl, dtype=torch.float32, device=self.device)
            E_t = min(int(state.get("E", 0)), _lib.MAX_ELITES) if (staging and nfl) else 0
            row = H * nfl if E_t else 0
            n_out = 2 * K * nu + E_t * (2 + row)
            b.size_out(n_out)
            mode, lam, k_el, tie = opt.fused_update_args()
            timing = None
            if self.record_kernel_events:
                evs = [self._timing_event() for _ in range(3)]
                timing = (C.c_void_p * 3)(*[e.handle for e in evs])
                self.kernel_events.append((evs[0], evs[1]))
                self.exchange_events.append((evs[1], evs[2]))
            off = b.offsets
            if one_call:
                in_place = self.host_block_in_place and self.model.closed_form and knots_out is None
                blk_dev = b.host_ptr if in_place else b.blk.data_ptr()
                b.blk_stale = in_place  # (jh_plan_step uploads the block itself unless it is read in place)
                st = lib.jh_plan_step(self.model.handle, blk_dev, b.host_ptr, b.nblk_bytes, int(off[1]), int(off[2]), int(off[3]), int(off[4]), noise_p, ldn, _lib.ptr(W), int(task.phase),
                                      shard.count, shard.offset, H, K, _lib.ptr(b.costs), _lib.ptr(knots_out), _lib.ptr(b.trace_buf) if nfl else None, mode, lam, k_el, tie, E_t, row,
                                      int(self._trace_colmajor) if nfl else 0, _lib.ptr(b.fused_scratch), b.out_host_ptr, b.done_ptr if (self.poll_completion and self.model.closed_form) else b.out_host_ptr, timing, stream)
                what = "jh_plan_step"
            else:
                # ---- several ranks: launch (rollout + cost + this rank's record [update record | E trace records]) -> one all-gather -> merge on every rank into
                # the same pinned output block; everything behind this branch is the one-GPU code
                b.blk_stale = False  # (jh_plan_step_shard uploads the block)
                L = int(lib.jh_shard_record_floats(K, nu, mode, k_el, E_t, row))
                if b.shard_rec is None or b.shard_rec.numel() != L:
                    b.shard_rec = torch.empty(L, dtype=torch.float32, device=self.device)
                st = lib.jh_plan_step_shard(self.model.handle, b.blk.data_ptr(), b.host_ptr, b.nblk_bytes, int(off[1]), int(off[2]), int(off[3]), int(off[4]), noise_p, ldn, _lib.ptr(W),
                                            int(task.phase), shard.count, shard.offset, H, K, _lib.ptr(b.costs), _lib.ptr(knots_out), _lib.ptr(b.trace_buf) if nfl else None, mode, lam, k_el,
                                            tie, E_t, row, int(self._trace_colmajor) if nfl else 0, _lib.ptr(b.fused_scratch), _lib.ptr(b.shard_rec), timing, stream)
                _lib.check(st, "jh_plan_step_shard")
                if self._prefetch_args is not None:  # the next iteration's noise on the side stream, enqueued in front of the collective: it overlaps the exchange
                    self._prefetch_noise(*self._prefetch_args, side=True)
                    self._prefetch_args = None
                b.shard_all = all_gather_records(b.shard_rec, self.group, force=self.force_shard_path)  # (world * L,), rank-major; kept alive until the merge has run
                done = None
                if self.record_kernel_events:
                    done = self._timing_event()
                    self.exchange_events[-1] = (evs[1], done)  # exchange = this rank's record + the all-gather + the merge
                st = lib.jh_plan_merge(_lib.ptr(b.shard_all), world, K, nu, mode, lam, k_el, tie, E_t, row, b.out_host_ptr, b.out_host_ptr, done.handle if done is not None else None, stream)
                what = "jh_plan_merge"
            try:
                _lib.check(st, what)
                if self._prefetch_args is not None:
                    self._prefetch_noise(*self._prefetch_args)
            finally:
                if st == 0:
                    _lib.check(lib.jh_download_end(), "jh_download_end")
            res = b.out_np[: 2 * K * nu if is_cem else K * nu].astype(np.float64)  # (not the trace records: their index column is an int bit pattern)
            if nfl:
                state["trace_buf"] = (b.trace_buf, H * nfl)
            state.update(costs=b.costs, knots_out=knots_out, noise_p=noise_p, ldn=ldn, knots_nku=None)
            if E_t:
                self._traces = None
                self._trace_stage = dict(kind="sensors", recs=b.out_np[2 * K * nu : n_out].copy(), stride=2 + row, E=int(state["E"]), x0=state["x0"], times=np.array(state["new_times"]),
                                         order=self.spline_order, H=H, K=K, nu=nu, index_is_bits=True, sorted=True)
            elif staging:
                state["stage"]()  # (no trace buffer: the elites' knots, re-rolled when the traces are read)
            nominal_n = (res[: K * nu].reshape(K, nu) - center[None, :]) / scale[None, :]
            if is_cem:
                opt.sigma = np.clip(res[K * nu : 2 * K * nu].reshape(K, nu) / scale[None, :], opt.sigma_min, opt.sigma_max)
            self._rewards, self._candidate_knots = None, None
            self._last_fused = dict(b=b, noise=noise, noise_p=noise_p, ldn=ldn, shard=shard, K=K, nu=nu)
            return nominal_n
        if self.record_kernel_events:
            ev0, ev1 = self._timing_event(), self._timing_event()
            ev0.record()
        if fused_cost:
            nfl = self._fused_trace_floats()
            if nfl:  # the kernel also writes the trace sensors of every rollout: `traces` becomes a gather of the elites' rows instead of a second rollout
                if b.trace_buf is None or b.trace_buf.numel() != shard.count * H * nfl:
                    b.trace_buf = torch.empty(shard.count * H * nfl, dtype=torch.float32, device=self.device)
            st = lib.jh_rollout_cost_traced(self.model.handle, _lib.ptr(b.x0), _lib.ptr(b.nominal), noise_p, ldn, _lib.ptr(b.sigma), _lib.ptr(W),
                                            _lib.ptr(b.lohi), _lib.ptr(b.tp), int(task.phase), shard.count, shard.offset, H, K, _lib.ptr(b.costs),
                                            _lib.ptr(knots_out), _lib.ptr(b.trace_buf) if nfl else None, stream)
            _lib.check(st, "jh_rollout_cost")
            costs = b.costs
            if nfl:
                state["trace_buf"] = (b.trace_buf, H * nfl)
        else:
            costs = self._materialised_costs(b.x0, b.nominal, noise_p, ldn, b.sigma, b.lohi, None, W, shard, H, K, stream)
            if knots_out is not None:  # the materialise path never wrote the candidates: sample them into the (K, nu, N) layout the fused kernel uses
                tmp = torch.empty((shard.count, K, nu), dtype=torch.float32, device=self.device)
                st = lib.jh_sample_knots(_lib.ptr(b.nominal), noise_p, ldn, _lib.ptr(b.sigma), _lib.ptr(b.lohi), shard.count, shard.offset, K, nu, _lib.ptr(tmp), stream)
                _lib.check(st, "jh_sample_knots")
                knots_out.copy_(tmp.permute(1, 2, 0))
        if self.record_kernel_events:
            ev1.record()
            self.kernel_events.append((ev0, ev1))
        is_cem = hasattr(opt, "sigma") and isinstance(getattr(opt, "sigma"), np.ndarray)
        state.update(costs=costs, knots_out=knots_out, noise_p=noise_p, ldn=ldn, knots_nku=None)
        tail = world == 1 and self.fused_update and hasattr(opt, "fused_update_args")
        if tail:
            # one GPU: block partials, merge and -- in the last iteration, when the rollout kernel wrote the trace buffer -- the trace elites' records in ONE launch,
            # nominal | sigma | records in one output block, one download (controller.py:288-299 without the seven-launch chain)
            tb, staging = state.get("trace_buf"), state.get("stage") is not None
            E_t = min(int(state.get("E", 0)), _lib.MAX_ELITES) if (staging and tb is not None) else 0
            row = tb[1] if E_t else 0
            n_out = 2 * K * nu + E_t * (2 + row)
            b.size_out(n_out)
            mode, lam, k_el, tie = opt.fused_update_args()
            if self.record_kernel_events:
                ex0, ex1 = self._timing_event(), self._timing_event()
                ex0.record()
            # the results go straight into the pinned host block (device-visible: hipHostMalloc): a few KB of stores from the last workgroup instead of a copy command
            o = b.out_host_ptr if self.zero_copy_out else b.out.data_ptr()
            st = lib.jh_update_fused(_lib.ptr(costs), None, _lib.ptr(b.nominal), noise_p, ldn, _lib.ptr(b.sigma), _lib.ptr(b.lohi), shard.count, shard.offset, K, nu, mode, lam,
                                     k_el, tie, E_t, _lib.ptr(tb[0]) if E_t else None, row, int(self._trace_colmajor) if E_t else 0, _lib.ptr(b.fused_scratch),
                                     o, o + 4 * K * nu, (o + 8 * K * nu) if E_t else None, stream)
            _lib.check(st, "jh_update_fused")
            if self.record_kernel_events:
                ex1.record()
                self.exchange_events.append((ex0, ex1))
            if staging and E_t:
                res = self._fetch(b, n_out, in_place=self.zero_copy_out, n_float=2 * K * nu if is_cem else K * nu)
                self._traces = None
                self._trace_stage = dict(kind="sensors", recs=b.out_np[2 * K * nu : n_out].copy(), stride=2 + row, E=int(state["E"]), x0=state["x0"], times=np.array(state["new_times"]),
                                         order=self.spline_order, H=H, K=K, nu=nu, index_is_bits=True, sorted=True)
            else:
                res = self._fetch(b, 2 * K * nu if is_cem else K * nu, behind=state.get("stage"), in_place=self.zero_copy_out)
        else:
            opt.device_partial(costs, None, b.nominal, noise_p, b.sigma, b.lohi, shard.count, shard.offset, b.scratch, b.rec, ldn=ldn, stream=stream)
            if self.record_kernel_events:  # the exchange of the per-rank records and the merge (bench.py attributes the plan step: kernel / exchange / host)
                ex0, ex1 = self._timing_event(), self._timing_event()
                ex0.record()
            recs = all_gather_records(b.rec, self.group)
            opt.device_merge(recs, world, b.out.data_ptr(), b.out.data_ptr() + 4 * K * nu, clip_sigma=False, stream=stream)
            if self.record_kernel_events:
                ex1.record()
                self.exchange_events.append((ex0, ex1))
            res = self._fetch(b, 2 * K * nu if is_cem else K * nu, behind=state.get("stage"))
        nominal_n = (res[: K * nu].reshape(K, nu) - center[None, :]) / scale[None, :]  # the update acted on the normalised candidates
        if is_cem:  # CEM: refit in normalised units
            opt.sigma = np.clip(res[K * nu : 2 * K * nu].reshape(K, nu) / scale[None, :], opt.sigma_min, opt.sigma_max)
        if nrm.needs_moments:  # running statistics over this iteration's raw candidates, all ranks (controller.py:290-291)
            if b.mom is None:
                b.mom = torch.empty(2 * nu, dtype=torch.float32, device=self.device)
            ctr = torch.from_numpy(np.asarray(nrm.mean, dtype=np.float32)).to(self.device)
            st = lib.jh_knot_moments(None, _lib.ptr(b.nominal), noise_p, ldn, _lib.ptr(b.sigma), _lib.ptr(b.lohi), _lib.ptr(ctr),
                                     shard.count, shard.offset, K, nu, _lib.ptr(b.mom), stream)
            _lib.check(st, "jh_knot_moments")
            m = all_gather_records(b.mom, self.group).cpu().numpy().astype(np.float64).reshape(world, 2, nu).sum(0)
            # the kernel centred on the fp32 image of the mean: shift the moments to the fp64 mean
            dm = np.asarray(nrm.mean, dtype=np.float32).astype(np.float64) - nrm.mean
            cnt = N * K
            s1 = m[0] + cnt * dm
            s2 = m[1] + 2 * dm * m[0] + cnt * dm * dm
            nrm.update_from_moments(cnt, s1, s2)
        state.update(costs=costs, knots_out=knots_out, noise_p=noise_p, ldn=ldn, knots_nku=None)
        self._rewards, self._candidate_knots = None, None
        self._last_fused = dict(b=b, noise=noise, noise_p=noise_p, ldn=ldn, shard=shard, K=K, nu=nu)
        return nominal_n

    def _candidates_iteration(self, lib, b: _PlanBuffers, nrm: Normalizer, nominal_n: np.ndarray, W, shard: Shard, H: int, K: int, nu: int, N: int,
                              stream, state: dict) -> np.ndarray:
        """judo/controller/controller.py:246-291 for an optimizer that only has the reference's numpy methods: its candidates are clipped on the
        host as the reference does, the shard's rows go to the device, rollout and reward run there, the rewards come back for its update.
        With several ranks every rank must sample the same candidates (seed numpy alike) -- the reference has no multi-process form of this."""
        opt, task = self.optimizer, self.task
        cand_n = np.asarray(opt.sample_control_knots(nominal_n), dtype=np.float64)
        if cand_n.shape != (N, K, nu):
            raise ValueError(f"sample_control_knots must return ({N}, {K}, {nu}), got {cand_n.shape}")
        r = task.actuator_ctrlrange
        cand_n = np.clip(cand_n, nrm.normalize(r[:, 0]), nrm.normalize(r[:, 1]))
        self._candidate_knots, self._last_fused = nrm.denormalize(cand_n), None
        self._pack_block(b, nrm.denormalize(nominal_n), None, self._raw_bounds(nrm))
        if b.knots_nku is None:
            b.knots_nku = torch.empty((shard.count, K, nu), dtype=torch.float32, device=self.device)
        b.knots_nku.copy_(torch.from_numpy(np.ascontiguousarray(self._candidate_knots[shard.offset : shard.offset + shard.count], dtype=np.float32)))
        costs = self._materialised_costs(b.x0, None, None, 0, None, b.lohi, b.knots_nku, W, shard, H, K, stream)
        rewards = -costs.cpu().numpy().astype(np.float64)
        if shard.world > 1:
            from judo_amd.distributed import all_gather_costs

            rewards = -all_gather_costs(costs, shard, self.group).cpu().numpy().astype(np.float64)
        self._rewards = rewards
        nominal_n = np.asarray(opt.update_nominal_knots(cand_n, rewards), dtype=np.float64)
        nrm.update(self._candidate_knots)
        state.update(costs=costs, knots_out=None, noise_p=None, ldn=0, knots_nku=b.knots_nku)
        return nominal_n

    def _materialised_costs(self, x0_d, nom_d, noise_p, ldn: int, sig_d, lohi_d, knots_nku, W, shard: Shard, H: int, K: int, stream) -> torch.Tensor:
        """judo/controller/controller.py:239-262 on the device: candidate splines at the rollout times, RolloutBackend.rollout,
        Task.reward, Task.post_rollout.  Returns costs = -rewards (fp32, device)."""
        lib, task, nu = _lib.lib(), self.task, self.nu
        controls = torch.empty((shard.count, H, nu), dtype=torch.float32, device=self.device)
        st = lib.jh_spline_controls(_lib.ptr(W), _lib.ptr(knots_nku), _lib.ptr(nom_d), noise_p, ldn, _lib.ptr(sig_d), _lib.ptr(lohi_d) if knots_nku is None else None,
                                    shard.count, shard.offset, H, K, nu, _lib.ptr(controls), stream)
        _lib.check(st, "jh_spline_controls")
        if task.uses_locomotion_policy:  # controller.py:265-273: commands -> policy + plant; the policy outputs carry over to the next plan step
            if self._last_policy_output is None:
                self._last_policy_output = torch.zeros((shard.count, POLICY_OUTPUT_DIM), dtype=torch.float32, device=self.device)
            states, sensors, self._last_policy_output = self.rollout_backend.rollout(x0_d, task.task_to_sim_ctrl(controls), self._last_policy_output,
                                                                                    cutoff_time=self.rollout_cutoff_time)
        elif hasattr(self.rollout_backend, "rollout_device"):
            states, sensors = self.rollout_backend.rollout_device(x0_d, task.task_to_sim_ctrl(controls))
        else:  # a RolloutBackend plugin with the reference's numpy signature only (assigned to `controller.rollout_backend`)
            s_np, y_np, _ = self.rollout_backend.rollout(np.asarray(self.current_state), task.task_to_sim_ctrl(controls).cpu().numpy().astype(np.float64), None)
            states, sensors = torch.as_tensor(s_np, dtype=torch.float32, device=self.device), torch.as_tensor(y_np, dtype=torch.float32, device=self.device)
        if getattr(task, "reward_accepts_torch", True):
            args = (states, sensors, controls)
        else:  # numpy-only plugin reward: one host round trip of the trajectories
            args = tuple(a.cpu().numpy().astype(np.float64) for a in (states, sensors, controls))
        task.post_rollout(*args, self.system_metadata)
        rewards = task.reward(*args, self.system_metadata)
        rewards = torch.as_tensor(rewards, device=self.device).to(torch.float32).reshape(-1)
        if rewards.shape[0] != shard.count:
            raise ValueError(f"Task.reward must return ({shard.count},) rewards, got {tuple(rewards.shape)}")
        self.last_rollout = (states, sensors, controls)
        return (-rewards).contiguous()

    @property
    def rewards_local(self) -> np.ndarray:
        """Rewards of this rank's shard of the last iteration (host copy on demand)."""
        return -self.costs_device.cpu().numpy().astype(np.float64)

    @property
    def rewards(self) -> np.ndarray:
        """`Controller.rewards` of the reference (controller.py:279): the last iteration's rewards; on the fused path they stay on the
        device until somebody reads them here (this rank's shard when the rollouts are sharded)."""
        if self._rewards is None:
            self._rewards = self.rewards_local
        return self._rewards

    @rewards.setter
    def rewards(self, value) -> None:
        self._rewards = value

    @property
    def candidate_knots(self) -> np.ndarray:
        """`Controller.candidate_knots` of the reference (controller.py:258, :313): the (N, K, nu) clipped candidates of the last iteration.  The
        fused path never stores them; they are re-derived from the noise of the last plan step (`jh_sample_knots`) when read (this rank's shard)."""
        if self._candidate_knots is None:
            f = self._last_fused
            if f is None:
                return np.tile(self.nominal_knots, (self.optimizer_cfg.num_rollouts, 1, 1))
            b, sh = f["b"], f["shard"]
            self._ensure_block(b)
            out = torch.empty((sh.count, f["K"], f["nu"]), dtype=torch.float32, device=self.device)
            st = _lib.lib().jh_sample_knots(_lib.ptr(b.nominal), f["noise_p"], f["ldn"], _lib.ptr(b.sigma), _lib.ptr(b.lohi), sh.count, sh.offset, f["K"], f["nu"],
                                            _lib.ptr(out), current_stream_ptr())
            _lib.check(st, "jh_sample_knots")
            self._candidate_knots = out.cpu().numpy().astype(np.float64)
        return self._candidate_knots

    def solver_stats(self, reset: bool = True) -> dict:
        """Counters of the articulated-body kernels since the last call (synchronises): contacts dropped above the per-rollout capacity,
        constraint solves stopped by the Newton iteration cap, iterations, steps.  Warns when rollouts were degraded by either limit."""
        if self.model is None:
            st = self.rollout_backend.engine.stats(reset)
            st = {"contact_overflow": st["contacts_dropped"], "newton_cap_hits": st["steps_at_cap"], "newton_iters": st["newton_iterations"], "steps": st["steps"]}
        else:
            st = self.model.stats(reset)
        if self.solver_warnings and st.get("overflow_pool_fallbacks", 0) > 0:
            warnings.warn(f"{self.task.name}: {st['overflow_pool_fallbacks']} launches ran without their overflow rows (the device's memory pool refused the scratch block): "
                          "contact capacity was what the LDS pool holds, not what jh_model_limits reports", stacklevel=2)
        if self.solver_warnings and st["steps"] > 0 and (st["contact_overflow"] > 1e-4 * st["steps"] or st["newton_cap_hits"] > 1e-2 * st["steps"]):
            warnings.warn(f"{self.task.name}: {st['contact_overflow']} contacts dropped above the kernel's per-rollout capacity and {st['newton_cap_hits']} "
                          f"constraint solves stopped at the iteration cap in {st['steps']} physics steps -- those rollouts are approximate", stacklevel=2)
        return st

    # ---- traces --------------------------------------------------------------------------------------------------
    def _fused_trace_floats(self) -> int:
        """Floats per rollout-step the fused kernel writes into a trace buffer, when they are exactly this task's trace sensors in order; else 0."""
        if not (self.fused_traces and self.model is not None and self.trace_sensors and self._num_trace_elites(1 << 30) > 0):
            return 0
        key = (self.model.kernel_generation, id(self.model))
        if getattr(self, "_trace_layout_key", None) != key:
            adr, nfl, cm = self.model.trace_layout()
            ok = nfl > 0 and [s["adr"] for s in self.trace_sensors] == [adr + 3 * k for k in range(nfl // 3)]
            self._trace_layout_key, self._trace_layout_nfl, self._trace_colmajor = key, (nfl if ok else 0), cm
        return self._trace_layout_nfl

    def _num_trace_elites(self, N: int) -> int:
        E = min(int(self.max_num_traces), N)  # controller.py:333-334
        if E > _lib.MAX_ELITES:
            warnings.warn(f"max_num_traces = {E} clamped to {_lib.MAX_ELITES} (include/judo_amd.h JH_MAX_ELITES)", stacklevel=3)
            E = _lib.MAX_ELITES
        return max(E, 0) if self.trace_sensors else 0

    def _stage_traces(self, lib, b: _PlanBuffers, state: dict, shard: Shard, world: int, E: int, x0: np.ndarray, new_times: np.ndarray, K: int, nu: int, stream) -> None:
        """What `update_traces` (controller.py:323-363) needs from this plan step, kept on the device: the E best rollouts of this shard as
        records [cost, global index, payload], all-gathered over the ranks.  payload = their clipped candidate knots (re-rolled in materialise
        mode when the traces are read) or, when the iteration materialised the sensors anyway, their trace-sensor rows.  No synchronisation here."""
        self._traces, self._trace_stage = None, None
        if E <= 0:
            self._traces = np.zeros((0, 2, 3))
            return
        costs = state["costs"]
        kl = min(E, shard.count)
        H = self.num_timesteps
        if self.last_rollout is not None and (not self.uses_fused_optimizer or not self.uses_fused_cost):
            # ties: the reference takes argsort(rewards)[-E:][::-1], i.e. among equal rewards the higher index first
            order = (shard.count - 1 - torch.argsort(costs.flip(0), stable=True))[:kl]
            cols = [s["adr"] + k for s in self.trace_sensors for k in range(3)]
            rows = self.last_rollout[1][order][:, :, cols].reshape(kl, -1)
            rec = torch.full((E, 2 + rows.shape[1]), float("inf"), dtype=torch.float32, device=self.device)
            rec[:kl, 0] = costs[order]
            rec[:kl, 1] = (order + shard.offset).to(torch.float32)
            rec[:kl, 2:] = rows
            rec[kl:, 1] = -1.0
            kind, stride = "sensors", 2 + rows.shape[1]
            recs = all_gather_records(rec.reshape(-1), self.group)
        else:
            stride = 2 + K * nu
            b.trace_flip ^= 1
            trace_rec = b.trace_recs[b.trace_flip]  # the previous plan step's stage may still be read from the other buffer
            if kl < E:
                trace_rec.fill_(float("inf"))
            self._ensure_block(b)
            st = lib.jh_topk_partial(_lib.ptr(costs), _lib.ptr(state["knots_nku"]), _lib.ptr(b.nominal), state["noise_p"], state["ldn"], _lib.ptr(b.sigma), _lib.ptr(b.lohi),
                                     shard.count, shard.offset, K, nu, kl, 1, _lib.ptr(b.scratch), _lib.ptr(trace_rec), stream)
            _lib.check(st, "jh_topk_partial")
            tb = state.get("trace_buf")
            if tb is not None:  # the elites' rows of the trace buffer travel instead of their knots: nothing is re-rolled when the traces are read
                row = tb[1]
                if b.trace_rows is None or b.trace_rows.numel() != 2 * E * (2 + row):
                    b.trace_rows = torch.empty(2 * E * (2 + row), dtype=torch.float32, device=self.device)
                out = b.trace_rows[b.trace_flip * E * (2 + row) : (b.trace_flip + 1) * E * (2 + row)]
                if kl < E:
                    out.fill_(float("inf"))  # (records beyond this shard's rollouts: an infinite cost marks them empty)
                st = lib.jh_trace_gather(_lib.ptr(trace_rec), kl, stride, shard.offset, shard.count, _lib.ptr(tb[0]), row, int(self._trace_colmajor), _lib.ptr(out), stream)
                _lib.check(st, "jh_trace_gather")
                kind, stride = "sensors", 2 + row
                recs = all_gather_records(out, self.group)
                self._trace_stage = dict(kind=kind, recs=recs, stride=stride, E=E, x0=x0, times=np.array(new_times), order=self.spline_order, H=H, K=K, nu=nu, index_is_bits=True)
                return
            kind = "knots"
            recs = all_gather_records(trace_rec, self.group)
        self._trace_stage = dict(kind=kind, recs=recs, stride=stride, E=E, x0=x0, times=np.array(new_times), order=self.spline_order, H=H, K=K, nu=nu,
                                 index_is_bits=(kind == "knots"))

    @property
    def traces(self) -> np.ndarray | None:
        """Line segments of the best rollouts' `trace*` framepos sensors, best rollout first: (E * n_trace_sensors * (H-1), 2, 3)
        (controller.py:323-363).  The reference fills this inside `update_action`; here the elite rollouts are re-rolled when it is first read."""
        if self._traces is None and self._trace_stage is not None:
            self._traces = self._finish_traces(self._trace_stage)
        return self._traces

    @traces.setter
    def traces(self, value) -> None:
        self._traces = value

    def update_traces(self) -> None:
        if self._trace_stage is None and self._traces is None:
            raise RuntimeError("update_traces() needs a completed update_action()")
        if self._trace_stage is not None:
            self._traces = self._finish_traces(self._trace_stage)

    def _finish_traces(self, st: dict) -> np.ndarray:
        S, H, E = len(self.trace_sensors), st["H"], st["E"]
        recs = st["recs"]
        recs = (recs.cpu().numpy() if torch.is_tensor(recs) else np.asarray(recs)).reshape(-1, st["stride"])
        idx = recs[:, 1].view(np.int32) if st["index_is_bits"] else recs[:, 1].astype(np.int64)
        cost = recs[:, 0]
        if st.get("sorted"):  # jh_update_fused: one rank's records, already best first (ties: higher index first), the empty ones last
            n = int(np.count_nonzero((idx >= 0) & np.isfinite(cost)))
            p = recs[: min(n, E), 2:].reshape(-1, H, S, 3).transpose(0, 2, 1, 3)  # (elites, S, H, 3)
            segs = np.empty((p.shape[0], S, H - 1, 2, 3))
            segs[:, :, :, 0] = p[:, :, :-1]
            segs[:, :, :, 1] = p[:, :, 1:]
            return segs.reshape(-1, 2, 3)  # elite-major, then sensor, then time
        idx = idx.astype(np.int64)
        cost = cost.astype(np.float64)
        ok = np.nonzero((idx >= 0) & np.isfinite(cost))[0]
        # best first; among equal costs the higher global index first (argsort(rewards)[-E:][::-1] with a stable sort)
        ok = ok[np.lexsort((-idx[ok], cost[ok]))][:E]
        if st["kind"] == "sensors":
            pts = recs[ok, 2:].astype(np.float64).reshape(len(ok), H, S, 3)
        else:
            knots = recs[ok, 2:].astype(np.float64).reshape(len(ok), st["K"], st["nu"])
            U = evaluate(st["order"], st["times"], knots, st["times"][0] + self.task.dt * np.arange(H))
            U = np.asarray(self.task.task_to_sim_ctrl(U), dtype=np.float64)  # the plant sees the mapped controls (controller.py:262-263)
            _, sensors, _ = GpuRolloutBackend(self.model, len(ok)).rollout(st["x0"], U)
            pts = np.stack([sensors[:, :, s["adr"] : s["adr"] + 3] for s in self.trace_sensors], axis=2)
        segs = np.stack([pts[:, :-1], pts[:, 1:]], axis=-2)  # (E, H-1, S, 2, 3)
        return np.ascontiguousarray(segs.transpose(0, 2, 1, 3, 4)).reshape(-1, 2, 3)  # elite-major, then sensor, then time


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

"""Optimizer plugin surface on the GPU: GpuMPPI, GpuCEM, GpuPS.

Same interface as the reference (judo/optimizers/base.py:27-96): `sample_control_knots(nominal (K,nu)) ->
(N,K,nu)`, `update_nominal_knots(sampled (N,K,nu), rewards (N,)) -> (K,nu)`, `pre_optimization`, `stop_cond`,
config read through properties on every call.  numpy in / numpy out at this boundary (the unmodified reference
`Controller` can drive these objects); the arithmetic runs in `jh_sample_knots`, `jh_mppi_partial/merge`,
`jh_topk_partial` / `jh_elite_merge`.

The build's own controller uses the fused path instead (`knot_sigma` + `device_update`): candidates are never
materialised, the update kernel re-derives them from the noise tensor.

Reference semantics kept on purpose (SURVEY.md section 8a "quirks"): sample 0 is the unperturbed nominal; MPPI shifts
by the minimum cost and includes sample 0 in the weights; CEM multiplies its sigma state by the ramp cumulatively on
every sampling call and only the elite refit resets it; CEM elites = flip(argsort(rewards))[:k]; PS = first argmax.
"""

from __future__ import annotations

from abc import ABC, abstractmethod
from typing import Generic, TypeVar

import numpy as np
import torch

from judo_amd import _lib
from judo_amd.config import CrossEntropyMethodConfig, MPPIConfig, OptimizerConfig, PredictiveSamplingConfig
from judo_amd.device import current_stream_ptr, f32, require_gpu

OptimizerConfigT = TypeVar("OptimizerConfigT", bound=OptimizerConfig)


def ramp_linspace(lo: float, hi: float, K: int) -> np.ndarray:
    return np.linspace(lo, hi, K, endpoint=True)


def _noise_ld(noise, n_local: int) -> int:
    """Row stride of a (K, nu, n) noise tensor (a view into a wider draw keeps the wide stride)."""
    return int(noise.stride(1)) if isinstance(noise, torch.Tensor) and noise.ndim == 3 else n_local


class _NoiseStream:
    """State of the optimizer's device noise stream: the seed and how many draws were taken from it (jh_noise_normal is counter-based)."""

    def __init__(self, seed: int) -> None:
        self.seed, self.draws = int(seed) & 0xFFFFFFFFFFFFFFFF, 0


class Optimizer(ABC, Generic[OptimizerConfigT]):
    """Base class (mirror of judo/optimizers/base.py:27-96)."""

    def __init__(self, config: OptimizerConfigT, nu: int, override_task_name: str | None = None) -> None:
        self.config = config
        self.nu = nu
        if override_task_name is not None:
            self.config.set_override(override_task_name)
        self._generator: _NoiseStream | None = None  # (seed, number of draws so far) of the device noise stream
        self._seed = 1234
        self.injected_noise: np.ndarray | None = None  # (N-1, K, nu): "same noise on both sides" parity runs
        self.last_noise: torch.Tensor | None = None  # (K, nu, N) device tensor used by the last sampling call

    # -- config passthrough (read on every call so that live edits take effect next plan step)
    @property
    def num_rollouts(self) -> int:
        return self.config.num_rollouts

    @property
    def num_nodes(self) -> int:
        return self.config.num_nodes

    @property
    def use_noise_ramp(self) -> bool:
        return self.config.use_noise_ramp

    @property
    def noise_ramp(self) -> float:
        return self.config.noise_ramp

    def pre_optimization(self, old_times: np.ndarray, new_times: np.ndarray) -> None:
        """Entry point before sampling (no-op by default)."""

    def stop_cond(self) -> bool:
        return False

    # -- noise ---------------------------------------------------------------------------------------------------
    def seed(self, seed: int) -> None:
        self._seed = int(seed)
        self._generator = None

    def draw_noise(self, n_local: int, n_offset: int, device: torch.device, out: torch.Tensor | None = None) -> torch.Tensor:
        """Standard-normal fp32 noise in the kernels' (K, nu, N) layout (rollout index fastest): a (K, nu, n_local) tensor whose last
        dimension is contiguous and whose row stride `stride(1)` is the `ldn` the kernels take (a view into the full draw when sharded).

        With `injected_noise` (reference layout (N-1, K, nu), sample n uses row n-1; or a list of such arrays, consumed one per call: one per
        optimiser iteration) the shard's slice is uploaded instead, so that a run can be replayed against the CPU oracle bit-for-bit on the input side."""
        K, nu = self.num_nodes, self.nu
        if self.injected_noise is not None:
            inj = self.injected_noise
            if isinstance(inj, (list, tuple)):
                if not inj:
                    raise ValueError("injected noise list is exhausted: one array per optimiser iteration is needed")
                inj, self.injected_noise = inj[0], list(inj[1:])
            inj = np.asarray(inj, dtype=np.float32)
            if inj.shape[1:] != (K, nu) or inj.shape[0] < n_offset + n_local - 1:
                raise ValueError(f"injected noise has shape {inj.shape}, need (>= {n_offset + n_local - 1}, {K}, {nu})")
            full = np.zeros((n_local, K, nu), dtype=np.float32)
            lo = max(1, n_offset)
            full[lo - n_offset :] = inj[lo - 1 : n_offset + n_local - 1]
            noise = torch.from_numpy(np.ascontiguousarray(full.transpose(1, 2, 0))).to(device)
        else:
            if self._generator is None:
                self._generator = _NoiseStream(self._seed)
            # Counter-based normals (jh_noise_normal: Philox4x32-10 + Box-Muller): element (k, u, n) of draw number d under the seed is a pure function of
            # (seed, d, k * nu + u, GLOBAL rollout index), so a rank generates exactly its shard's columns -- N/G * K * nu normals, not N * K * nu -- and with one
            # seed the candidates, and so the plan, do not depend on the number of GPUs (a sharded run reproduces the single-GPU costs bit for bit).
            from judo_amd import _lib
            from judo_amd.device import current_stream_ptr

            if out is not None and out.dim() == 3 and tuple(out.shape[:2]) == (K, nu) and out.shape[2] >= n_local and out.stride(2) == 1 and out.stride(0) == nu * out.stride(1):
                buf = out
            else:
                buf = torch.empty((K, nu, n_local), dtype=torch.float32, device=device)
            st = _lib.lib().jh_noise_normal(self._generator.seed, self._generator.draws, K * nu, int(n_offset), int(n_local), buf.data_ptr(), int(buf.stride(1)), current_stream_ptr())
            _lib.check(st, "jh_noise_normal")
            self._generator.draws += 1
            noise = buf if buf.shape[2] == n_local else buf[:, :, :n_local]
        self.last_noise = noise
        return noise

    # -- drop-in API (the reference's two abstract methods, judo/optimizers/base.py:58-84) -----------------------------------
    @abstractmethod
    def sample_control_knots(self, nominal_knots: np.ndarray) -> np.ndarray:
        """(K, nu) nominal knots -> (N, K, nu) candidates; row 0 is the nominal itself."""

    @abstractmethod
    def update_nominal_knots(self, sampled_knots: np.ndarray, rewards: np.ndarray) -> np.ndarray:
        """(N, K, nu) candidates and their (N,) rewards -> new (K, nu) nominal knots."""


class FusedOptimizer(Optimizer[OptimizerConfigT]):
    """An optimizer the controller can run without ever materialising the candidates: it supplies the per-knot sigma, a shard-local
    device reduction and a merge of the per-GPU records.  A plugin that only implements the reference's two methods (any
    `Optimizer` subclass, e.g. the mock of the reference's tests/test_controller/test_controller.py:16-33) is served by the controller's candidate-array path."""

    # -- per-knot sigma (K, nu) -------------------------------------------------------------------------------
    @abstractmethod
    def knot_sigma(self) -> np.ndarray:
        """Per-(knot, actuator) standard deviation used by the next sampling call (may advance optimizer state)."""

    def sample_control_knots(self, nominal_knots: np.ndarray) -> np.ndarray:
        dev = require_gpu()
        N, K, nu = self.num_rollouts, self.num_nodes, self.nu
        nominal_knots = np.asarray(nominal_knots)
        if nominal_knots.shape != (K, nu):
            raise ValueError(f"nominal_knots must have shape ({K}, {nu}), got {nominal_knots.shape}")
        sigma = f32(self.knot_sigma(), dev)
        noise = self.draw_noise(N, 0, dev)
        out = torch.empty((N, K, nu), dtype=torch.float32, device=dev)
        nom = f32(nominal_knots, dev)
        st = _lib.lib().jh_sample_knots(_lib.ptr(nom), _lib.ptr(noise), _noise_ld(noise, N), _lib.ptr(sigma), None, N, 0, K, nu, _lib.ptr(out), current_stream_ptr())
        _lib.check(st, "jh_sample_knots")
        return out.cpu().numpy().astype(np.float64)

    # -- fused path ----------------------------------------------------------------------------------------------
    @abstractmethod
    def device_partial(self, costs, knots_nku, nominal, noise, sigma, lohi, n_local, n_offset, scratch, rec, K=None, ldn=None, stream=None) -> None:
        """Shard-local reduction into `rec` (see include/judo_amd.h)."""

    @abstractmethod
    def record_floats(self) -> int:
        ...

    @abstractmethod
    def device_merge(self, recs, G, nominal_out, sigma_out, K=None, clip_sigma=True, stream=None) -> None:
        ...

    def _check_update_args(self, sampled_knots, rewards):
        sampled_knots = np.asarray(sampled_knots)
        rewards = np.asarray(rewards)
        if sampled_knots.ndim != 3 or sampled_knots.shape[2] != self.nu:
            raise ValueError(f"sampled_knots must be (N, K, {self.nu}), got {sampled_knots.shape}")
        if rewards.shape != (sampled_knots.shape[0],):
            raise ValueError(f"rewards must be ({sampled_knots.shape[0]},), got {rewards.shape}")
        return sampled_knots, rewards

    def _update_via_device(self, sampled_knots, rewards) -> tuple[np.ndarray, np.ndarray | None]:
        dev = require_gpu()
        sampled_knots, rewards = self._check_update_args(sampled_knots, rewards)
        N, K, nu = sampled_knots.shape
        knots = f32(sampled_knots, dev)
        costs = f32(-rewards, dev)
        scratch = torch.empty(int(_lib.lib().jh_update_scratch_floats(N, K, nu)), dtype=torch.float32, device=dev)
        rec = torch.empty(self.record_floats_for(K), dtype=torch.float32, device=dev)
        self.device_partial(costs, knots, None, None, None, None, N, 0, scratch, rec, K=K)
        nominal = torch.empty(K * nu, dtype=torch.float32, device=dev)
        sigma = torch.empty(K * nu, dtype=torch.float32, device=dev)
        self.device_merge(rec, 1, nominal, sigma, K=K)
        return nominal.cpu().numpy().astype(np.float64).reshape(K, nu), sigma.cpu().numpy().astype(np.float64).reshape(K, nu)

    def record_floats_for(self, K: int) -> int:
        return 2 + K * self.nu


class GpuMPPI(FusedOptimizer[MPPIConfig]):
    """MPPI (judo/optimizers/mppi.py:21-82)."""

    def __init__(self, config: MPPIConfig, nu: int) -> None:
        super().__init__(config, nu)

    @property
    def sigma(self) -> float:
        return self.config.sigma

    @property
    def temperature(self) -> float:
        return self.config.temperature

    def knot_sigma(self) -> np.ndarray:
        return _ramped_sigma(self)

    def update_nominal_knots(self, sampled_knots: np.ndarray, rewards: np.ndarray) -> np.ndarray:
        return self._update_via_device(sampled_knots, rewards)[0]

    def record_floats(self) -> int:
        return self.record_floats_for(self.num_nodes)

    def device_partial(self, costs, knots_nku, nominal, noise, sigma, lohi, n_local, n_offset, scratch, rec, K=None, ldn=None, stream=None) -> None:
        K = K or self.num_nodes
        ldn = _noise_ld(noise, n_local) if ldn is None else int(ldn)
        st = _lib.lib().jh_mppi_partial(_lib.ptr(costs), _lib.ptr(knots_nku), _lib.ptr(nominal), _lib.ptr(noise), ldn, _lib.ptr(sigma), _lib.ptr(lohi),
                                        n_local, n_offset, K, self.nu, float(self.temperature), _lib.ptr(scratch), _lib.ptr(rec), current_stream_ptr() if stream is None else stream)
        _lib.check(st, "jh_mppi_partial")

    def fused_update_args(self) -> tuple[int, float, int, int]:
        """(mode, temperature, elites, tie rule) of jh_update_fused: the one-launch form of device_partial + device_merge on one GPU."""
        return 0, float(self.temperature), 0, 0

    def device_merge(self, recs, G, nominal_out, sigma_out, K=None, clip_sigma=True, stream=None) -> None:
        K = K or self.num_nodes
        st = _lib.lib().jh_mppi_merge(_lib.ptr(recs), G, K, self.nu, float(self.temperature), _lib.ptr(nominal_out), current_stream_ptr() if stream is None else stream)
        _lib.check(st, "jh_mppi_merge")


def _ramped_sigma(opt) -> np.ndarray:
    """(K, nu) per-knot sigma of MPPI / PS (mppi.py:44-52): a function of four config values, cached on the optimizer (it is asked for in every iteration)."""
    key = (opt.num_nodes, opt.nu, bool(opt.use_noise_ramp), float(opt.noise_ramp), float(opt.sigma))
    hit = getattr(opt, "_sigma_cache", None)
    if hit is None or hit[0] != key:
        K = opt.num_nodes
        s = opt.noise_ramp * ramp_linspace(1 / K, 1, K)[:, None] * opt.sigma if opt.use_noise_ramp else np.full((K, 1), opt.sigma)
        hit = opt._sigma_cache = (key, np.broadcast_to(s, (K, opt.nu)).astype(np.float64))
    return hit[1].copy()


class _EliteOptimizer(FusedOptimizer[OptimizerConfigT]):
    """Shared top-k machinery of CEM (k = num_elites, ties high-index-first) and PS (k = 1, first argmax)."""

    tie_high = 1

    def num_keep(self) -> int:
        return 1

    def sigma_bounds(self) -> tuple[float, float]:
        return 0.0, float("inf")

    def record_floats_for(self, K: int) -> int:
        return self.num_keep() * (2 + K * self.nu)

    def record_floats(self) -> int:
        return self.record_floats_for(self.num_nodes)

    def device_partial(self, costs, knots_nku, nominal, noise, sigma, lohi, n_local, n_offset, scratch, rec, K=None, ldn=None, stream=None) -> None:
        K = K or self.num_nodes
        ldn = _noise_ld(noise, n_local) if ldn is None else int(ldn)
        st = _lib.lib().jh_topk_partial(_lib.ptr(costs), _lib.ptr(knots_nku), _lib.ptr(nominal), _lib.ptr(noise), ldn, _lib.ptr(sigma), _lib.ptr(lohi),
                                        n_local, n_offset, K, self.nu, self.num_keep(), self.tie_high, _lib.ptr(scratch), _lib.ptr(rec), current_stream_ptr() if stream is None else stream)
        _lib.check(st, "jh_topk_partial")

    def fused_update_args(self) -> tuple[int, float, int, int]:
        return 1, 0.0, int(self.num_keep()), int(self.tie_high)

    def device_merge(self, recs, G, nominal_out, sigma_out, K=None, clip_sigma=True, stream=None) -> None:
        """clip_sigma=False returns the raw population std: the controller clips it after mapping it back to the action normaliser's
        units (cem.py:91 clips the std of the NORMALISED elites)."""
        K = K or self.num_nodes
        smin, smax = self.sigma_bounds() if clip_sigma else (0.0, float("inf"))
        st = _lib.lib().jh_elite_merge(_lib.ptr(recs), G, self.num_keep(), K, self.nu, self.tie_high, smin, min(smax, 3.0e38), _lib.ptr(nominal_out),
                                       _lib.ptr(sigma_out), current_stream_ptr() if stream is None else stream)
        _lib.check(st, "jh_elite_merge")


class GpuCEM(_EliteOptimizer[CrossEntropyMethodConfig]):
    """Cross-entropy method (judo/optimizers/cem.py:20-92)."""

    tie_high = 1

    def __init__(self, config: CrossEntropyMethodConfig, nu: int) -> None:
        super().__init__(config, nu)
        self.sigma = ((self.sigma_min + self.sigma_max) / 2) * np.ones((config.num_nodes, nu))

    @property
    def sigma_min(self) -> float:
        return self.config.sigma_min

    @property
    def sigma_max(self) -> float:
        return self.config.sigma_max

    @property
    def num_elites(self) -> int:
        return self.config.num_elites

    def num_keep(self) -> int:
        return int(self.num_elites)

    def sigma_bounds(self) -> tuple[float, float]:
        return float(self.sigma_min), float(self.sigma_max)

    def pre_optimization(self, old_times: np.ndarray, new_times: np.ndarray) -> None:
        """Node-count change: sigma is re-interpolated linearly in time with extrapolation (cem.py:44-53)."""
        if len(self.sigma) != self.num_nodes:
            old_times, new_times = np.asarray(old_times, dtype=np.float64), np.asarray(new_times, dtype=np.float64)
            if len(old_times) != len(self.sigma):  # (scipy's interp1d raises the same way in the reference)
                raise ValueError(f"CEM sigma has {len(self.sigma)} rows but the previous plan has {len(old_times)} knots: x and y arrays must be equal in length along the interpolation axis")
            i = np.clip(np.searchsorted(old_times, new_times, side="right") - 1, 0, len(old_times) - 2)
            a = ((new_times - old_times[i]) / (old_times[i + 1] - old_times[i]))[:, None]
            self.sigma = self.sigma[i] + a * (self.sigma[i + 1] - self.sigma[i])

    def knot_sigma(self) -> np.ndarray:
        K = self.num_nodes
        if self.use_noise_ramp:  # cumulative: the state itself is overwritten (cem.py:69-72)
            ramp = ramp_linspace(self.noise_ramp / K, self.noise_ramp, K)[:, None]
            self.sigma = np.clip(self.sigma * ramp, self.sigma_min, self.sigma_max)
        return self.sigma

    def update_nominal_knots(self, sampled_knots: np.ndarray, rewards: np.ndarray) -> np.ndarray:
        nominal, sigma = self._update_via_device(sampled_knots, rewards)
        self.sigma = sigma
        return nominal


class GpuPS(_EliteOptimizer[PredictiveSamplingConfig]):
    """Predictive sampling (judo/optimizers/ps.py:17-65)."""

    tie_high = 0  # np.argmax: the first maximum wins

    def __init__(self, config: PredictiveSamplingConfig, nu: int) -> None:
        super().__init__(config, nu)

    @property
    def sigma(self) -> float:
        return self.config.sigma

    def knot_sigma(self) -> np.ndarray:
        return _ramped_sigma(self)

    def update_nominal_knots(self, sampled_knots: np.ndarray, rewards: np.ndarray) -> np.ndarray:
        return self._update_via_device(sampled_knots, rewards)[0]


# names the reference registers (judo/optimizers/__init__.py:28-46)
_registered_optimizers: dict[str, tuple[type, type]] = {
    "cem": (GpuCEM, CrossEntropyMethodConfig),
    "mppi": (GpuMPPI, MPPIConfig),
    "ps": (GpuPS, PredictiveSamplingConfig),
}


def get_registered_optimizers() -> dict[str, tuple[type, type]]:
    return _registered_optimizers


def register_optimizer(name: str, optimizer_type: type, optimizer_config_type: type) -> None:
    _registered_optimizers[name] = (optimizer_type, optimizer_config_type)

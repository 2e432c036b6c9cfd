"""Action normalisers of the controller (mirror of judo/utils/normalization.py:40-230), as the affine maps the GPU path needs.

The reference samples, clips and updates the knots in *normalised* units and denormalises the candidates before the rollout.
Every normaliser it ships is affine per actuator, `raw = center + scale * normalised`, so the kernels keep working on raw knots
when the host hands them

    nominal_eff = denormalize(normalize(nominal)),   sigma_eff = sigma * scale,   bounds_eff = denormalize(normalize(bounds)),

and maps the kernels' results back with `normalize`.  Host state only (a few floats per actuator); the running statistics are
updated from two moments per actuator that a reduction kernel (`jh_knot_moments`) takes over the candidate knots.
"""

from __future__ import annotations

import warnings
from abc import ABC, abstractmethod

import numpy as np


class Normalizer(ABC):
    def __init__(self, dim: int) -> None:
        self.dim = dim

    @abstractmethod
    def normalize(self, x: np.ndarray) -> np.ndarray: ...

    @abstractmethod
    def denormalize(self, x: np.ndarray) -> np.ndarray: ...

    def update(self, x: np.ndarray) -> None:
        """Update the statistics from raw data (..., dim); a no-op for the static normalisers."""

    def noise_scale(self) -> np.ndarray:
        """d raw / d normalised per actuator: what a unit of normalised sampling noise is worth in raw units."""
        return self.denormalize(np.ones(self.dim)) - self.denormalize(np.zeros(self.dim))

    needs_moments = False


class IdentityNormalizer(Normalizer):
    def normalize(self, x):
        return x

    def denormalize(self, x):
        return x


class MinMaxNormalizer(Normalizer):
    """[-1, 1] over the finite ctrlranges; actuators with an infinite bound are left alone (normalization.py:94-137)."""

    def __init__(self, dim: int, min: np.ndarray, max: np.ndarray, eps: float = 1e-6) -> None:
        super().__init__(dim)
        self.min, self.max, self.eps = np.asarray(min, dtype=np.float64), np.asarray(max, dtype=np.float64), eps
        self.norm_dims = np.where((self.min != -np.inf) & (self.max != np.inf))[0]
        if len(self.norm_dims) != dim:
            excluded = np.where((self.min == -np.inf) | (self.max == np.inf))[0]
            warnings.warn(f"MinMaxNormalizer: {len(excluded)} action dimensions ({excluded.tolist()}) have infinite range and will not be normalized.",
                          UserWarning, stacklevel=2)

    def normalize(self, x):
        out = np.array(x, dtype=np.float64, copy=True)
        d = self.norm_dims
        out[..., d] = 2 * (out[..., d] - self.min[d]) / (self.max[d] - self.min[d]) - 1
        return out

    def denormalize(self, x):
        out = np.array(x, dtype=np.float64, copy=True)
        d = self.norm_dims
        out[..., d] = (out[..., d] + 1) * (self.max[d] - self.min[d]) / 2 + self.min[d]
        return out


class RunningMeanStdNormalizer(Normalizer):
    """Running mean / std per actuator (normalization.py:138-213), including the reference's batch form of Welford's update and its
    asymmetric pair normalize = (x - mean) / (std + eps), denormalize = x * std + mean."""

    needs_moments = True

    def __init__(self, dim: int, init_std: float = 1.0, min_std: float = 1e-5, max_std: float = 1e3, eps: float = 1e-6) -> None:
        super().__init__(dim)
        self.eps, self.min_std, self.max_std = eps, min_std, max_std
        self.count = 0
        self.mean = np.zeros(dim)
        self.std = np.ones(dim) * init_std
        self.M2 = np.zeros(dim)

    def update(self, x: np.ndarray) -> None:
        x = np.asarray(x, dtype=np.float64)
        assert x.shape[-1] == self.dim, f"Expected dimension {self.dim}, but got {x.shape[-1]}"
        flat = x.reshape(-1, self.dim)
        d = flat - self.mean
        self.update_from_moments(flat.shape[0], d.sum(0), (d * d).sum(0))

    def update_from_moments(self, batch: int, s1: np.ndarray, s2: np.ndarray) -> None:
        """Same update from s1 = sum(x - mean_old), s2 = sum((x - mean_old)^2) over the batch:
        sum((x - mean_old)(x - mean_new)) = s2 - (mean_new - mean_old) * s1."""
        self.count += int(batch)
        shift = np.asarray(s1, dtype=np.float64) / self.count
        self.mean = self.mean + shift
        self.M2 = np.maximum(self.M2 + np.asarray(s2, dtype=np.float64) - shift * s1, 0)
        self.std = np.clip(np.sqrt(self.M2 / self.count), self.min_std, self.max_std)

    def normalize(self, x):
        return (np.asarray(x, dtype=np.float64) - self.mean) / (self.std + self.eps)

    def denormalize(self, x):
        return np.asarray(x, dtype=np.float64) * self.std + self.mean


normalizer_registry = {"none": IdentityNormalizer, "min_max": MinMaxNormalizer, "running": RunningMeanStdNormalizer}


def make_normalizer(normalizer_type: str, dim: int, **kwargs) -> Normalizer:
    if normalizer_type not in normalizer_registry:
        raise ValueError(f"Invalid normalizer type: {normalizer_type}")
    return normalizer_registry[normalizer_type](dim, **kwargs)

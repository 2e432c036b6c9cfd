"""Hyper-parameter dataclasses of the hot path, with the reference's task-keyed override table.

Mirrors (names, fields, defaults, override semantics):
  judo/config.py:13-96                  OverridableConfig.set_override / set_config_overrides
  judo/optimizers/base.py:15-21         OptimizerConfig
  judo/optimizers/mppi.py:14-18         MPPIConfig
  judo/optimizers/cem.py:12-17          CrossEntropyMethodConfig
  judo/optimizers/ps.py:11-14           PredictiveSamplingConfig
  judo/controller/controller.py:34-42   ControllerConfig
  judo/optimizers/overrides.py, judo/controller/overrides.py   (the shipped per-task values; checked
                                        against tests/golden/configs.json, generated from the reference)
The GUI `@slider` metadata of the reference is not part of the hot path and is not mirrored.
"""

from __future__ import annotations

import dataclasses
from dataclasses import dataclass
from typing import Any, Literal

import numpy as np

# class -> override key -> field name -> value.  (`_OVERRIDE_REGISTRY` is the reference's name, judo/config.py:9; its tests reach into it.)
_OVERRIDE_REGISTRY: dict[type, dict[str, dict[str, Any]]] = {}
_OVERRIDES = _OVERRIDE_REGISTRY


def set_config_overrides(override_key: str, cls: type, field_override_values: dict[str, Any]) -> None:
    """Register (or update) per-key field values for a config class (judo/config.py:66-96).  A class that is not a dataclass is a TypeError; a name that
    is not a field of the class is skipped with a UserWarning, the other names of the same call still register; an empty dict registers the key."""
    import warnings

    if not dataclasses.is_dataclass(cls):
        raise TypeError(f"Provided class {cls.__name__} is not a dataclass.")
    per_key = _OVERRIDE_REGISTRY.setdefault(cls, {}).setdefault(override_key, {})
    known = {f.name for f in dataclasses.fields(cls)}
    for name, value in field_override_values.items():
        if name not in known:
            warnings.warn(f"Field '{name}' not found in class '{cls.__name__}'. No override value added for this field under key '{override_key}'.", UserWarning, stacklevel=2)
            continue
        per_key[name] = value


def _same(a: Any, b: Any) -> bool:
    if isinstance(a, np.ndarray) or isinstance(b, np.ndarray):
        return isinstance(a, np.ndarray) and isinstance(b, np.ndarray) and np.array_equal(a, b)
    return bool(a == b)


@dataclass
class OverridableConfig:
    """A config whose fields can be switched to a registered per-key (per-task) set of values (judo/config.py:12-63)."""

    def __post_init__(self) -> None:
        _OVERRIDE_REGISTRY.setdefault(type(self), {})  # an existing entry (overrides registered before the first instance) is left alone

    def set_override(self, key: str, reset_to_defaults: bool = True) -> None:
        """Fields with a (non-None) value registered under `key` take it (the registered object itself, not a copy).  With `reset_to_defaults` every other field
        goes back to its default -- a fresh `default_factory()` result for factory fields -- and a field that has neither is left as it is with a UserWarning;
        without it the other fields keep their current values, silently.  An unknown key behaves like a key without values."""
        import warnings

        chosen = _OVERRIDE_REGISTRY.get(type(self), {}).get(key, {})
        for f in dataclasses.fields(self):
            value = chosen.get(f.name)
            if value is not None:
                if not _same(getattr(self, f.name, dataclasses.MISSING), value):
                    setattr(self, f.name, value)
            elif reset_to_defaults:
                if f.default is not dataclasses.MISSING:
                    default = f.default
                elif f.default_factory is not dataclasses.MISSING:  # type: ignore[misc]
                    default = f.default_factory()  # type: ignore[misc]
                else:
                    warnings.warn(f"Field '{f.name}' has no default value to reset to and no override for key '{key}'. Its current value remains unchanged.", UserWarning, stacklevel=2)
                    continue
                if not _same(getattr(self, f.name, dataclasses.MISSING), default):
                    setattr(self, f.name, default)


@dataclass
class OptimizerConfig(OverridableConfig):
    num_rollouts: int = 16
    num_nodes: int = 4
    use_noise_ramp: bool = False
    noise_ramp: float = 2.5


@dataclass
class MPPIConfig(OptimizerConfig):
    sigma: float = 0.1
    temperature: float = 0.05


@dataclass
class CrossEntropyMethodConfig(OptimizerConfig):
    sigma_min: float = 0.1
    sigma_max: float = 1.0
    num_elites: int = 2


@dataclass
class PredictiveSamplingConfig(OptimizerConfig):
    sigma: float = 0.05


@dataclass
class ControllerConfig(OverridableConfig):
    horizon: float = 1.0
    spline_order: Literal["zero", "linear", "cubic"] = "linear"
    control_freq: float = 20.0
    max_opt_iters: int = 1
    max_num_traces: int = 5
    action_normalizer: Literal["none", "min_max", "running"] = "none"


def _register_shipped_overrides() -> None:
    """The values the reference ships for the four BASELINE tasks (+ the leap variants)."""
    ramp = {"num_nodes": 4, "num_rollouts": 32, "use_noise_ramp": True}
    for task in ("cylinder_push", "cartpole"):
        set_config_overrides(task, PredictiveSamplingConfig, dict(ramp))
        set_config_overrides(task, CrossEntropyMethodConfig, dict(ramp, num_elites=2))
        set_config_overrides(task, MPPIConfig, dict(ramp))
        set_config_overrides(task, ControllerConfig, {"horizon": 1.0, "spline_order": "zero"})
    for task, n_cem_mppi in (("leap_cube", 32), ("leap_cube_down", 64), ("caltech_leap_cube", 32)):
        set_config_overrides(task, PredictiveSamplingConfig, dict(ramp, noise_ramp=4.0, sigma=0.2))
        set_config_overrides(task, CrossEntropyMethodConfig, dict(ramp, num_rollouts=n_cem_mppi, num_elites=3, noise_ramp=4.0))
        set_config_overrides(task, MPPIConfig, dict(ramp, num_rollouts=n_cem_mppi, noise_ramp=4.0, sigma=0.2, temperature=0.0025))
        set_config_overrides(task, ControllerConfig, {"horizon": 1.0, "spline_order": "cubic", "max_num_traces": 1})
    fr3 = {"num_nodes": 4, "num_rollouts": 64, "use_noise_ramp": True, "noise_ramp": 4.0}
    set_config_overrides("fr3_pick", PredictiveSamplingConfig, dict(fr3, num_nodes=8, sigma=0.2))
    set_config_overrides("fr3_pick", CrossEntropyMethodConfig, dict(fr3, sigma_min=0.01, sigma_max=0.3, num_elites=3))
    set_config_overrides("fr3_pick", MPPIConfig, dict(fr3, sigma=0.01, temperature=0.002))
    set_config_overrides("fr3_pick", ControllerConfig, {"horizon": 1.0, "spline_order": "linear", "max_num_traces": 3})
    spot = {"num_rollouts": 24, "num_nodes": 3, "use_noise_ramp": True, "noise_ramp": 3.5}   # judo/optimizers/overrides.py:188-199
    for task in ("spot_base", "spot_box_push", "spot_navigate", "spot_tire_roll", "spot_tire_upright"):
        set_config_overrides(task, PredictiveSamplingConfig, dict(spot))
        set_config_overrides(task, CrossEntropyMethodConfig, dict(spot, num_elites=3))
        set_config_overrides(task, MPPIConfig, dict(spot))
        set_config_overrides(task, ControllerConfig, {"horizon": 2.0})                          # judo/controller/overrides.py:79-88


_register_shipped_overrides()


def as_plain_dict(cfg: Any) -> dict[str, Any]:
    out = {}
    for f in dataclasses.fields(cfg):
        v = getattr(cfg, f.name)
        out[f.name] = v.tolist() if isinstance(v, np.ndarray) else v
    return out

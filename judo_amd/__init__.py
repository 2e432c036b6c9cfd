"""judo_amd -- MI355X-native sampling-MPC rollout engine behind judo's Optimizer / Task / RolloutBackend surface.

Only the hot path is here: sample -> clip -> spline -> rollout -> cost -> update, as hand-written gfx950 HIP kernels in
`libjudo_amd.so` (C ABI in include/judo_amd.h), driven through ctypes with PyTorch-ROCm tensors as device memory.
Importing the package does not need a GPU; any compute call does (there is no CPU fallback).
"""

from judo_amd.config import (
    ControllerConfig,
    CrossEntropyMethodConfig,
    MPPIConfig,
    OptimizerConfig,
    PredictiveSamplingConfig,
    set_config_overrides,
)

__all__ = [
    "ControllerConfig",
    "CrossEntropyMethodConfig",
    "MPPIConfig",
    "OptimizerConfig",
    "PredictiveSamplingConfig",
    "set_config_overrides",
]
__version__ = "0.1.0"

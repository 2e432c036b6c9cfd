"""Records exchanged either side of the plan step (mirror of judo/app/structs.py:30-84) and their Arrow form.

`MujocoState` is what the plant publishes (Controller.update_states consumes qpos/qvel/time/sim_metadata);
`SplineData` is what the controller publishes (knot times, knot values, interpolation kind) and what the plant evaluates to
obtain the control at its own clock.  The reference ships both through `dora_utils.dataclasses.to_arrow/from_arrow`
(third-party, absent from /root/reference); `to_arrow` / `from_arrow` below define an explicit, self-describing layout with the
same information: one flat float64 Arrow array plus a string metadata dict carrying the field order, shapes and scalars.
"""

from __future__ import annotations

import json
from dataclasses import dataclass, field, fields
from typing import Any, Callable

import numpy as np

from judo_amd.spline import SPLINE_KINDS, evaluate


@dataclass
class MujocoState:
    """judo/app/structs.py:30-41."""

    time: float
    qpos: np.ndarray
    qvel: np.ndarray
    xpos: np.ndarray = field(default_factory=lambda: np.zeros((0, 3)))
    xquat: np.ndarray = field(default_factory=lambda: np.zeros((0, 4)))
    mocap_pos: np.ndarray = field(default_factory=lambda: np.zeros((0, 3)))
    mocap_quat: np.ndarray = field(default_factory=lambda: np.zeros((0, 4)))
    sim_metadata: dict[str, Any] = field(default_factory=dict)


@dataclass
class SplineData:
    """(Possibly batched) spline: t (T,), x (..., T, m), kind as scipy.interpolate.interp1d (judo/app/structs.py:58-84).
    Queries outside [t0, t_end] return the first / last knot when `extrapolate` (the reference's fill_value), else raise."""

    t: np.ndarray
    x: np.ndarray
    kind: str = "zero"
    extrapolate: bool = True

    def __post_init__(self) -> None:
        if self.kind not in SPLINE_KINDS:
            raise ValueError(f"spline kind {self.kind!r} is not one of {SPLINE_KINDS}")
        self.t = np.asarray(self.t, dtype=np.float64)
        self.x = np.asarray(self.x, dtype=np.float64)
        if self.t.ndim != 1 or self.x.shape[-2] != self.t.shape[0]:
            raise ValueError(f"t must be (T,) and x (..., T, m); got {self.t.shape} and {self.x.shape}")

    def spline(self) -> Callable[[Any], np.ndarray]:
        """Callable q -> values, the role of the interp1d object the reference returns."""

        def f(q):
            qq = np.atleast_1d(np.asarray(q, dtype=np.float64))
            if not self.extrapolate and ((qq < self.t[0]).any() or (qq > self.t[-1]).any()):
                raise ValueError("A value in the query is outside the interpolation range.")
            flat = self.x.reshape((-1,) + self.x.shape[-2:])
            out = np.stack([evaluate(self.kind, self.t, k, qq) for k in flat]).reshape(self.x.shape[:-2] + (qq.shape[0], self.x.shape[-1]))
            return out[..., 0, :] if np.ndim(q) == 0 else out

        return f


# ------------------------------------------------------------------------------------------------ Arrow form
def to_arrow(obj) -> tuple[Any, dict[str, str]]:
    """dataclass -> (pyarrow float64 array of every ndarray field, concatenated; metadata: layout + scalar fields as JSON strings)."""
    import pyarrow as pa

    chunks, layout, scalars = [], [], {}
    for f in fields(obj):
        v = getattr(obj, f.name)
        if isinstance(v, np.ndarray):
            layout.append([f.name, list(v.shape)])
            chunks.append(np.asarray(v, dtype=np.float64).reshape(-1))
        elif isinstance(v, dict):
            scalars[f.name] = {k: (np.asarray(x).tolist() if isinstance(x, (np.ndarray, np.generic)) else x) for k, x in v.items()}
        else:
            scalars[f.name] = v
    flat = np.concatenate(chunks) if chunks else np.zeros(0)
    return pa.array(flat, type=pa.float64()), {"type": type(obj).__name__, "layout": json.dumps(layout), "scalars": json.dumps(scalars)}


def from_arrow(arr, metadata: dict[str, str], cls: type):
    """Inverse of `to_arrow` (a record of another type or with a foreign layout raises ValueError)."""
    if metadata.get("type") != cls.__name__:
        raise ValueError(f"record is a {metadata.get('type')!r}, expected {cls.__name__!r}")
    flat = np.asarray(arr.to_numpy(zero_copy_only=False), dtype=np.float64)
    kw, o = dict(json.loads(metadata["scalars"])), 0
    for name, shape in json.loads(metadata["layout"]):
        n = int(np.prod(shape)) if shape else 1
        kw[name] = flat[o : o + n].reshape(shape).copy()
        o += n
    if o != flat.shape[0]:
        raise ValueError(f"record holds {flat.shape[0]} values, layout describes {o}")
    names = {f.name for f in fields(cls)}
    if set(kw) - names:
        raise ValueError(f"unknown fields {sorted(set(kw) - names)} for {cls.__name__}")
    return cls(**kw)

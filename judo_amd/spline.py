"""Control-spline evaluation as a weight matrix (host side, numpy).

The reference builds `scipy.interpolate.interp1d(times, knots, kind, axis=-2, fill_value=(first,last),
bounds_error=False)` and evaluates it at `t + dt*arange(H)` (judo/controller/controller.py:261-262,382-401).
interp1d is linear in the knot values, so for fixed (kind, knot times, query times) the whole evaluation is
`U[n,h,:] = sum_k W[h,k] * knots[n,k,:]`; the kernels take W (H x K) and never see scipy.  Pinned against the
reference by tests/golden/spline.npz.
"""

from __future__ import annotations

import numpy as np

SPLINE_KINDS = {"zero": 0, "linear": 1, "cubic": 3}


def _not_a_knot_second_derivs(t: np.ndarray) -> np.ndarray:
    """G (K x K) with m = G @ y the knot second derivatives of the not-a-knot cubic spline through (t, y)."""
    K = len(t)
    h = np.diff(t)
    T = np.zeros((K, K))
    B = np.zeros((K, K))
    for i in range(1, K - 1):
        T[i, i - 1 : i + 2] = (h[i - 1], 2 * (h[i - 1] + h[i]), h[i])
        B[i, i - 1 : i + 2] = (6 / h[i - 1], -6 / h[i - 1] - 6 / h[i], 6 / h[i])
    T[0, :3] = (h[1], -(h[0] + h[1]), h[0])  # third derivative continuous across t[1]
    T[-1, -3:] = (h[-1], -(h[-2] + h[-1]), h[-2])  # ... and across t[K-2]
    return np.linalg.solve(T, B)


def spline_weights(kind: str, knot_times: np.ndarray, query_times: np.ndarray) -> np.ndarray:
    """W (len(query), K), float64.  Queries before/after the knot span hold the first/last knot."""
    if kind not in SPLINE_KINDS:
        raise ValueError(f"spline kind must be one of {list(SPLINE_KINDS)}, got {kind!r}")
    t = np.asarray(knot_times, dtype=np.float64)
    q = np.asarray(query_times, dtype=np.float64)
    K = len(t)
    if K < 2 or (kind == "cubic" and K < 4):
        raise ValueError("cubic splines require at least 4 nodes" if kind == "cubic" else "need at least 2 nodes")
    W = np.zeros((len(q), K))
    below, above = q < t[0], q > t[-1]
    W[below, 0] = 1.0
    W[above, -1] = 1.0
    inside = ~(below | above)
    qi = q[inside]
    i = np.clip(np.searchsorted(t, qi, side="right") - 1, 0, K - 2)  # interval [t_i, t_{i+1})
    rows = np.nonzero(inside)[0]
    if kind == "zero":  # previous-knot hold; the value at a knot is that knot
        idx = np.where(qi >= t[-1], K - 1, i)
        W[rows, idx] = 1.0
    elif kind == "linear":
        a = (qi - t[i]) / (t[i + 1] - t[i])
        W[rows, i] = 1 - a
        W[rows, i + 1] += a
    else:
        G = _not_a_knot_second_derivs(t)
        hh = t[i + 1] - t[i]
        a, b = t[i + 1] - qi, qi - t[i]
        c0 = a**3 / (6 * hh) - hh * a / 6
        c1 = b**3 / (6 * hh) - hh * b / 6
        W[rows] = c0[:, None] * G[i] + c1[:, None] * G[i + 1]
        W[rows, i] += a / hh
        W[rows, i + 1] += b / hh
    return W


def evaluate(kind: str, knot_times: np.ndarray, knots: np.ndarray, query_times: np.ndarray) -> np.ndarray:
    """interp1d(...)(query) for knots of shape (..., K, nu) -> (..., len(query), nu)."""
    W = spline_weights(kind, knot_times, query_times)
    return np.einsum("hk,...ku->...hu", W, np.asarray(knots, dtype=np.float64))

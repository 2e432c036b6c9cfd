"""Test infrastructure: checks of the physics oracle (`oracle/jo_engine.c`) by code that shares no line with it.

VERDICT round 2 asked for exactly this: kernel-vs-oracle agreement is one reading of MuJoCo agreeing with itself twice unless something that was NOT derived
from the same page looks at the oracle too.  Two such things live here (numpy / scipy only, fp64):

* `primal_objective` -- MuJoCo's documented primal problem  min_a  1/2 (a - a0)' M (a - a0) + s(J a - aref)  written down from the documentation's row
  types (computation chapter, "Constraint model" / "Solver": quadratic equality rows, Huber friction-loss rows, one-sided limit / pyramidal rows, the
  three-zone elliptic cone), with its gradient; `generic_minimise` minimises it with scipy's trust-region Newton on a finite-difference Hessian of that
  gradient (no line search, no cone Hessian: nothing of `solve_constraints`).  The arrays (M, a0, J, aref, R) come out of the oracle
  (`Model.problem`), so this checks the SOLVER and the cone functions; `fd_contact_jacobian` below checks the Jacobian rows themselves against finite
  differences of the oracle's kinematics.
* `gjk_distance` / `support_depth` -- box / sphere / capsule / cylinder support functions, a plain GJK for separated shapes and a sampled
  minimum-translation search for overlapping ones: an independent statement of "deepest penetration, along which normal" for the narrow phase
  (SAT + face clipping in the oracle and in the kernels).
"""

from __future__ import annotations

import numpy as np

EQUALITY, FRICTION, LIMIT, FRICTIONLESS, PYRAMIDAL, ELLIPTIC = range(6)  # jo_engine.h efc types


# ------------------------------------------------------------------------------------------------ primal objective
def constraint_cost(P: dict, jar: np.ndarray, want_grad: bool = True):
    """s(jar) and ds/djar for the rows of problem P (see module docstring)."""
    tp, R, D = P["type"], P["R"], 1.0 / P["R"]
    cost, g = 0.0, np.zeros_like(jar)
    eq = tp == EQUALITY
    cost += 0.5 * np.sum(D[eq] * jar[eq] ** 2)
    g[eq] = D[eq] * jar[eq]
    fr = np.nonzero(tp == FRICTION)[0]
    for r in fr:  # Huber: quadratic inside |x| < R eta, linear outside, C1 at the joins
        eta, x = P["frictionloss"][r], jar[r]
        if abs(x) < R[r] * eta:
            cost += 0.5 * D[r] * x * x
            g[r] = D[r] * x
        else:
            cost += eta * abs(x) - 0.5 * R[r] * eta * eta
            g[r] = eta * np.sign(x)
    one = (tp == LIMIT) | (tp == FRICTIONLESS) | (tp == PYRAMIDAL)
    xm = np.minimum(jar, 0.0)
    cost += 0.5 * np.sum(D[one] * xm[one] ** 2)
    g[one] = D[one] * xm[one]
    if P["cone"] == 1:
        for c in range(P["ncon"]):
            r0, dim = P["con_adr"][c], P["con_dim"][c]
            if r0 < 0 or dim < 3 or tp[r0] != ELLIPTIC:
                continue
            mu, f = P["con_mu"][c], P["con_friction"][c]
            x = jar[r0 : r0 + dim]
            scale = np.concatenate([[mu], f[: dim - 1]])
            U = x * scale
            N, T = U[0], float(np.linalg.norm(U[1:]))
            if N >= mu * T or (T <= 0 and N >= 0):  # top zone: inside the dual cone, no force
                continue
            if mu * N + T <= 0 or (T <= 0 and N < 0):  # bottom zone: every row quadratic
                Dr = D[r0 : r0 + dim]
                cost += 0.5 * np.sum(Dr * x * x)
                g[r0 : r0 + dim] = Dr * x
                continue
            Dm = D[r0] / (mu * mu * (1.0 + mu * mu))  # middle zone: squared distance to the cone in the scaled coordinates
            NT = N - mu * T
            cost += 0.5 * Dm * NT * NT
            dU = np.concatenate([[1.0], -mu * U[1:] / T])
            g[r0 : r0 + dim] = Dm * NT * dU * scale
    return (cost, g) if want_grad else cost


def primal_objective(P: dict, a: np.ndarray):
    """cost(a), grad(a) of 1/2 (a - a0)' M (a - a0) + s(J a - aref)."""
    da = a - P["qacc_smooth"]
    Mda = P["M"] @ da
    s, gs = constraint_cost(P, P["J"] @ a - P["aref"])
    return 0.5 * da @ Mda + s, Mda + P["J"].T @ gs


def generic_minimise(P: dict, a_init: np.ndarray | None = None, rounds: int = 4):
    """Minimise the primal objective with scipy's trust-region Newton (`trust-exact`) on a finite-difference Hessian of the analytic gradient.  The variables
    are scaled by 1/sqrt(diag M) (inertias span 1e-5 .. 1 kg m^2).  Returns the minimiser and its objective."""
    from scipy.optimize import minimize

    nv = P["M"].shape[0]
    sc = 1.0 / np.sqrt(np.diag(P["M"]))
    a0 = P["qacc_smooth"]

    def f(z):
        c, g = primal_objective(P, a0 + sc * z)
        return c, g * sc

    def hess(z):  # central differences of the gradient (the objective is piecewise quadratic: exact away from the zone boundaries)
        H = np.zeros((nv, nv))
        for i in range(nv):
            h = 1e-4 * max(1.0, abs(z[i]))
            e = np.zeros(nv); e[i] = h
            H[:, i] = (f(z + e)[1] - f(z - e)[1]) / (2 * h)
        H = 0.5 * (H + H.T)
        return H + 1e-12 * np.trace(H) / nv * np.eye(nv)

    z = np.zeros(nv) if a_init is None else (a_init - a0) / sc
    best = None
    for _ in range(rounds):
        res = minimize(lambda zz: f(zz)[0], z, jac=lambda zz: f(zz)[1], hess=hess, method="trust-exact", options=dict(gtol=1e-14, maxiter=400))
        z = res.x
        if best is None or res.fun < best[1]:
            best = (a0 + sc * z, float(res.fun))
    return best


def kkt_residual(P: dict, a: np.ndarray) -> float:
    """|grad| scaled as MuJoCo scales its solver statistics: by 1 / trace(M) (mj_solNewton's `scale` up to the 1/nv factor)."""
    return float(np.linalg.norm(primal_objective(P, a)[1]) / np.trace(P["M"]))


def kkt_excess(P: dict, a: np.ndarray, ulps: float = 32.0) -> float:
    """KKT residual beyond what fp64 can resolve, in the same scaling.  One ulp of the iterate moves gradient row i by sum_j |H_ij| eps |a_j| (H = M + J' D J
    over the rows that carry force): with cylinder_push's friction clamp (mu = 1e-5, R = 1.4e-11, D = 7e10) that is ~1e-4 -- no solver can push the gradient of
    such a problem below it, and asking for 1e-9 there would only test the rounding.  Returns max_i (|g_i| - ulps * floor_i)_+ / trace(M)."""
    g = primal_objective(P, a)[1]
    gs = constraint_cost(P, P["J"] @ a - P["aref"])[1]
    act = gs != 0
    H = P["M"] + (P["J"][act].T * (1.0 / P["R"][act])) @ P["J"][act]
    floor = np.finfo(np.float64).eps * (np.abs(H) @ np.abs(a))
    return float(np.maximum(np.abs(g) - ulps * floor, 0.0).max() / np.trace(P["M"]))


# ------------------------------------------------------------------------------------------------ support-function geometry
def _rot(R):
    return np.asarray(R, dtype=np.float64).reshape(3, 3)


class Shape:
    """Convex primitive with a support function in world coordinates.  kind: box (half sizes 3), sphere (r), capsule (r, half length along local z),
    cylinder (r, half height along local z) -- MuJoCo's size conventions."""

    def __init__(self, kind: str, size, pos, R) -> None:
        self.kind, self.size, self.pos, self.R = kind, np.atleast_1d(np.asarray(size, dtype=np.float64)), np.asarray(pos, dtype=np.float64), _rot(R)

    def support(self, d: np.ndarray) -> np.ndarray:
        dl = self.R.T @ d
        if self.kind == "box":
            pl = np.where(dl >= 0, self.size[:3], -self.size[:3])
        elif self.kind == "sphere":
            pl = self.size[0] * dl / max(np.linalg.norm(dl), 1e-300)
        elif self.kind == "capsule":
            pl = self.size[0] * dl / max(np.linalg.norm(dl), 1e-300) + np.array([0, 0, self.size[1] if dl[2] >= 0 else -self.size[1]])
        elif self.kind == "cylinder":
            rad = np.hypot(dl[0], dl[1])
            pl = np.array([self.size[0] * dl[0] / rad, self.size[0] * dl[1] / rad, 0.0]) if rad > 1e-300 else np.zeros(3)
            pl[2] = self.size[1] if dl[2] >= 0 else -self.size[1]
        else:
            raise ValueError(self.kind)
        return self.pos + self.R @ pl

    def contains(self, p: np.ndarray, tol: float = 0.0) -> bool:
        q = self.R.T @ (np.asarray(p) - self.pos)
        if self.kind == "box":
            return bool(np.all(np.abs(q) <= self.size[:3] + tol))
        if self.kind == "sphere":
            return bool(np.linalg.norm(q) <= self.size[0] + tol)
        if self.kind == "capsule":
            z = np.clip(q[2], -self.size[1], self.size[1])
            return bool(np.linalg.norm(q - np.array([0, 0, z])) <= self.size[0] + tol)
        return bool(np.hypot(q[0], q[1]) <= self.size[0] + tol and abs(q[2]) <= self.size[1] + tol)


def separation_along(A: Shape, B: Shape, d: np.ndarray) -> float:
    """Signed gap between the supporting planes of A and B normal to unit d (from A towards B): > 0 separated by at least that much along d,
    < 0 -> translating B by that much along d brings the shapes into touch.  max over d = signed distance for convex shapes."""
    return float(d @ B.support(-d) - d @ A.support(d))


def _sphere_dirs(n: int, rng) -> np.ndarray:
    v = rng.standard_normal((n, 3))
    return v / np.linalg.norm(v, axis=1, keepdims=True)


def signed_distance(A: Shape, B: Shape, n_dirs: int = 4000, refine: int = 60, seed: int = 0, extra_dirs=None):
    """max over unit directions d of separation_along(A, B, d), by sampling + local refinement (Nelder-Mead on the sphere): the signed distance of two
    convex shapes (positive: gap; negative: penetration depth = length of the minimum translation) and the direction attaining it (A -> B)."""
    from scipy.optimize import minimize

    rng = np.random.default_rng(seed)
    dirs = _sphere_dirs(n_dirs, rng)
    c = B.pos - A.pos
    cand = [dirs]
    if np.linalg.norm(c) > 0:
        cand.append((c / np.linalg.norm(c))[None])
    for S in (A, B):  # face normals and edge-edge cross products are where polytopes attain the maximum
        cand.append(S.R.T); cand.append(-S.R.T)
    ee = np.array([np.cross(A.R[:, i], B.R[:, j]) for i in range(3) for j in range(3)])
    nn = np.linalg.norm(ee, axis=1)
    ee = ee[nn > 1e-9] / nn[nn > 1e-9, None]
    if len(ee):
        cand.append(ee); cand.append(-ee)
    if extra_dirs is not None:
        cand.append(np.asarray(extra_dirs, dtype=np.float64))
    dirs = np.concatenate(cand)
    vals = np.array([separation_along(A, B, d) for d in dirs])
    order = np.argsort(-vals)[:4]
    best_v, best_d = vals[order[0]], dirs[order[0]]

    def neg(t, base):
        d = base + t[0] * u1 + t[1] * u2
        d = d / np.linalg.norm(d)
        return -separation_along(A, B, d)

    for k in order:
        base = dirs[k]
        u1 = np.cross(base, [1.0, 0, 0] if abs(base[0]) < 0.9 else [0, 1.0, 0]); u1 /= np.linalg.norm(u1)
        u2 = np.cross(base, u1)
        r = minimize(neg, np.zeros(2), args=(base,), method="Nelder-Mead", options=dict(xatol=1e-10, fatol=1e-13, maxiter=refine * 10))
        if -r.fun > best_v:
            d = base + r.x[0] * u1 + r.x[1] * u2
            best_v, best_d = -r.fun, d / np.linalg.norm(d)
    return float(best_v), best_d

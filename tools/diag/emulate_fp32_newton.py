"""CPU emulation (numpy float32) of the cooperative kernels' Newton loop on a problem exported from the oracle: same order of operations in spirit (gradient first,
Cholesky, safeguarded 1-D Newton line search with a cap, incremental jar), not bit-identical.  Shows what an fp32 iterate does near the optimum of a stiff problem.
usage: python tools/diag/emulate_fp32_newton.py gpurun_out/r3/cap_state.npz"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import oracle as O
f32 = np.float32
d = np.load(sys.argv[1]); om = O.Model("fr3_pick")
x, u = d["x"], d["u"]
P = om.problem(x[:16], x[16:], u)
M, J, aref, D, a0 = (P[k].astype(f32) for k in ("M", "J", "aref", "R", "qacc_smooth")); D = f32(1) / D
tp, fl, R = P["type"], P["frictionloss"].astype(f32), P["R"].astype(f32)
nv = len(a0); iMd = f32(1) / np.diag(M)
def rows(jar):  # force (= -ds/djar) and curvature per row
    f = np.zeros_like(jar); h = np.zeros_like(jar)
    for r in range(len(jar)):
        xx = jar[r]
        if tp[r] == 0: f[r] = -D[r] * xx; h[r] = D[r]
        elif tp[r] == 1:
            lim = R[r] * fl[r]
            if xx <= -lim: f[r] = fl[r]
            elif xx >= lim: f[r] = -fl[r]
            else: f[r] = -D[r] * xx; h[r] = D[r]
        elif xx < 0: f[r] = -D[r] * xx; h[r] = D[r]
    return f, h
fs = M @ a0; snorm = np.sum(fs * fs * iMd, dtype=f32); tol = f32(1e-5); lstol = f32(1e-2)
a = a0.copy(); jar = (J @ a - aref).astype(f32)
a_star = P["qacc"]
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 30):
    f, h = rows(jar)
    g = (M @ (a - a0) - J.T @ f).astype(f32)
    gn = np.sum(g * g * iMd, dtype=f32)
    H = (M + (J.T * h) @ J).astype(f32)
    L = np.linalg.cholesky(H.astype(np.float64)).astype(f32)
    p = -np.linalg.solve(H.astype(np.float64), g.astype(np.float64)).astype(f32)
    jp = (J @ p).astype(f32); Mp = (M @ p).astype(f32)
    pMp = f32(p @ Mp); pMd = f32(Mp @ (a - a0)); gp = f32(g @ p)
    lo, hi, al, nev = f32(0), f32(-1), f32(1), 0
    for ls in range(12):
        nev += 1
        f2, h2 = rows((jar + al * jp).astype(f32))
        d1 = f32(-(f2 @ jp) + pMd + al * pMp); d2 = f32(h2 @ (jp * jp) + pMp)
        if abs(d1) <= lstol * abs(gp): break
        if d1 < 0: lo = al
        else: hi = al
        nx = al - d1 / d2
        if hi < 0:
            if nx <= lo: nx = 2 * al
        elif nx <= lo or nx >= hi: nx = f32(0.5) * (lo + hi)
        al = f32(nx)
    step = al * p
    pn = np.sum(np.diag(M) * step * step, dtype=f32); an = np.sum(np.diag(M) * a * a, dtype=f32)
    print(f"it {it:2d} gn {gn:.3e} (tol {tol*tol*snorm:.1e}) gp {gp:+.3e} alpha {al:.4g} ls-evals {nev} -gp*alpha {-gp*al:.2e} |step|/|a| {np.sqrt(pn/an):.2e} err13 {a[13]-a_star[13]:+.3e} err14 {a[14]-a_star[14]:+.3e} active {int((h>0).sum())}")
    a = (a + step).astype(f32); jar = (jar + al * jp).astype(f32)

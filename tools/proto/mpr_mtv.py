"""Prototype of the kernel-side general convex routine (jh_coop.h `convex_mtv`): iterated portal refinement.  A ray cast on the Minkowski difference M = A - B (XenoCollide's
portal refinement, O(1) state: three support points) returns the boundary point the ray leaves M through and the supporting normal there; casting again from the ORIGIN along that
normal can only shorten the exit distance, and the fixed point -- ray direction = surface normal -- is the foot of the perpendicular from the origin onto the boundary of M: the
minimum translation.  Checked here against the oracle's GJK + EPA (a different algorithm) in fp64 and with fp32 arithmetic.  usage: python tools/proto/mpr_mtv.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

def support(kind, size, pos, R, d, F):
    dl = R.T.astype(F) @ d
    if kind == "box":
        pl = np.where(dl >= 0, size[:3], -size[:3]).astype(F)
    elif kind == "sphere":
        pl = (size[0] * dl / max(np.linalg.norm(dl), F(1e-30))).astype(F)
    elif kind == "capsule":
        pl = (size[0] * dl / max(np.linalg.norm(dl), F(1e-30))).astype(F); pl[2] += size[1] if dl[2] >= 0 else -size[1]
    else:
        rad = np.hypot(dl[0], dl[1]); pl = np.zeros(3, F)
        if rad > F(1e-12) * (abs(dl[2]) + rad): pl[:2] = size[0] * dl[:2] / rad
        pl[2] = size[1] if dl[2] >= 0 else -size[1]
    return (pos.astype(F) + R.astype(F) @ pl).astype(F)

class Mink:
    def __init__(self, A, B, F): self.A, self.B, self.F, self.n = A, B, F, 0
    def __call__(self, d):
        self.n += 1
        a = support(*self.A, d, self.F); b = support(*self.B, -d, self.F)
        return a - b, a

INNER = int(os.environ.get('INNER', '24'))
def raycast(S, p0, r, F, tol):
    """ray p0 + t r (r unit) leaves M at t*; returns (t*, unit normal n of the supporting plane there, A-side witness) -- None when the portal cannot be built"""
    cr = np.cross
    v1, a1 = S(r)
    if np.linalg.norm(cr(v1 - p0, r)) < F(1e-7) * max(np.linalg.norm(v1 - p0), F(1e-30)):
        return F((v1 - p0) @ r), r, a1
    d = cr(v1 - p0, r); d /= np.linalg.norm(d)
    v2, a2 = S(d)
    d = cr(v1 - p0, v2 - p0)
    if np.linalg.norm(d) < F(1e-20): return None
    d /= np.linalg.norm(d)
    if d @ r < 0: d = -d; v1, v2, a1, a2 = v2, v1, a2, a1   # orientation: (v1-p0) x (v2-p0) . r > 0
    v3, a3 = S(d)
    for it in range(24):   # bring the ray inside the cone (p0; v1, v2, v3); invariant: r . ((v1-p0) x (v2-p0)) >= 0
        e1, e2, e3 = v1 - p0, v2 - p0, v3 - p0
        if r @ cr(e2, e3) < 0:      # outside the face (v2, v3): replace v1 by the support beyond it
            d = cr(e3, e2); d /= max(np.linalg.norm(d), F(1e-30)); v1, a1 = S(d); v1, v2, a1, a2 = v2, v1, a2, a1; v1, v3, a1, a3 = v3, v1, a3, a1  # -> (v2, v3, new) keeps the orientation
            continue
        if r @ cr(e3, e1) < 0:      # outside the face (v3, v1): replace v2
            d = cr(e1, e3); d /= max(np.linalg.norm(d), F(1e-30)); v2, a2 = S(d); v2, v3, a2, a3 = v3, v2, a3, a2   # -> (v1, v3, new)
            continue
        break
    for it in range(INNER):
        n = cr(v2 - v1, v3 - v1); nn = np.linalg.norm(n)
        if nn < F(1e-30): return None
        n /= nn
        if n @ r < 0: n = -n
        v4, a4 = S(n)
        if (v4 - v1) @ n <= tol or it == INNER - 1:
            den = n @ r
            t = ((v1 - p0) @ n) / den
            # witness on A: barycentric coordinates of the exit point in the portal
            x = p0 + t * r
            T = np.stack([v1, v2, v3], 1).astype(np.float64)
            lam = np.linalg.lstsq(np.vstack([T, np.ones(3)]), np.append(x.astype(np.float64), 1.0), rcond=None)[0]
            return F(t), n, (lam[0] * a1 + lam[1] * a2 + lam[2] * a3).astype(F)
        c = cr(v4 - p0, r)
        t1, t2, t3 = (v1 - p0) @ c, (v2 - p0) @ c, (v3 - p0) @ c
        if t2 >= 0 and t1 <= 0: v3, a3 = v4, a4
        elif t3 >= 0 and t2 <= 0: v1, a1 = v4, a4
        else: v2, a2 = v4, a4
    return None

def convex_mtv(A, B, F=np.float64, tol=None, passes=int(os.environ.get('PASSES', '6'))):
    """(dist < 0, normal A->B, position) or None when the shapes do not overlap"""
    tol = tol if tol is not None else (F(1e-10) if F is np.float64 else F(2e-7))
    S = Mink(A, B, F)
    v0 = (A[2] - B[2]).astype(F)          # interior point of M = A - B
    l0 = np.linalg.norm(v0)
    if l0 < F(1e-9): v0 = np.array([1e-6, 0, 0], F); l0 = np.linalg.norm(v0)
    res = raycast(S, v0, -v0 / l0, F, tol)
    if res is None: return None
    t, n, wa = res
    if t < l0: return None                # the ray leaves M before it reaches the origin: separated
    # from here the apex is the origin (inside M).  Start the descent from the best of a fixed set of directions -- the normals of the flat pieces of the boundary of M (face
    # normals / axes of both shapes and the cross products of their axes) and the normal of the first cast: h(d) = d . support(d) is an upper bound of the depth for every d
    zero = np.zeros(3, F); depth = S(n)[0] @ n
    axes = []
    for (kind, size, pos, R) in (A, B):
        axes += [R[:, k].astype(F) for k in range(3)] if kind == "box" else ([R[:, 2].astype(F)] if kind in ("cylinder", "capsule") else [])
    na = len(axes)
    cand = list(axes)
    for i in range(na):
        for j in range(i + 1, na):
            c = np.cross(axes[i], axes[j]); l = np.linalg.norm(c)
            if l > F(1e-3): cand.append((c / l).astype(F))
    for c in cand:
        for sg in (F(1), F(-1)):
            h = S(sg * c)[0] @ (sg * c)
            if h < depth: depth, n = h, sg * c
    for k in range(passes):
        res = raycast(S, zero, n, F, tol)
        if res is None: break
        t2, n2, wa2 = res
        d2 = S(n2)[0] @ n2
        done = (n2 @ n) > 1 - (F(1e-12) if F is np.float64 else F(3e-7))
        if d2 <= depth: depth, wa, nbest = d2, wa2, n2
        n = n2
        if done: break
    # normal of M = A - B pointing out of M along which A must move; contact normal from A to B is the opposite
    return -depth, n, wa - n * (depth * F(0.5)), S.n

if __name__ == "__main__":
    from oracle import oracle as O
    rng = np.random.default_rng(3)
    def rot():
        q = rng.standard_normal(4); q /= np.linalg.norm(q); w, x, y, z = q
        return np.array([[1-2*(y*y+z*z), 2*(x*y-z*w), 2*(x*z+y*w)], [2*(x*y+z*w), 1-2*(x*x+z*z), 2*(y*z-x*w)], [2*(x*z-y*w), 2*(y*z+x*w), 1-2*(x*x+y*y)]]), q
    for kinds in (("box", "cylinder"), ("cylinder", "cylinder"), ("capsule", "cylinder"), ("sphere", "cylinder"), ("box", "box")):
        for F in (np.float64, np.float32):
            errs, nerrs, evals, deps, miss, extra, n = [], [], [], [], 0, 0, 0
            for trial in range(600):
                def mk(kind):
                    R, q = rot()
                    size = rng.uniform(0.01, 0.05, 3) if kind == "box" else (np.array([rng.uniform(0.01, 0.04), 0, 0]) if kind == "sphere" else np.array([rng.uniform(0.008, 0.02), rng.uniform(0.005, 0.05), 0]))
                    return kind, size, R, q
                A, B = mk(kinds[0]), mk(kinds[1])
                pb = rng.standard_normal(3); pb *= rng.uniform(0.0, 0.07) / np.linalg.norm(pb)
                ref = O.collide_pair(A[0], A[1], np.zeros(3), A[3], B[0], B[1], pb, B[3])
                if kinds == ("box", "box") and ref: ref = [min(ref, key=lambda o: o[0])]
                got = convex_mtv((A[0], A[1].astype(F), np.zeros(3), A[2]), (B[0], B[1].astype(F), pb, B[2]), F)
                if not ref:
                    extra += got is not None and got[0] < -1e-5; continue
                if ref[0][0] > -2e-4: continue
                if got is None: miss += 1; continue
                n += 1; errs.append(abs(got[0] - ref[0][0])); nerrs.append(1 - got[1] @ ref[0][2]); evals.append(got[3]); deps.append(-ref[0][0])
            errs, nerrs, deps = np.array(errs), np.array(nerrs), np.array(deps)
            sh = deps < 3e-3
            print(f"   shallow (< 3 mm, {sh.sum()}): depth error max {errs[sh].max():.1e}, worst normal {nerrs[sh].max():.1e}; wrong by more than 1e-5 m: shallow {(errs[sh] > 1e-5).sum()}, deep {(errs[~sh] > 1e-5).sum()} of {(~sh).sum()}")
            print(f"{kinds} {F.__name__}: {n} overlapping pairs, missed {miss}, spurious {extra}; |depth error| median {np.median(errs):.1e} 99% {np.percentile(errs, 99):.1e} max {errs.max():.1e} m; "
                  f"1 - n.n_ref median {np.median(nerrs):.1e} 99% {np.percentile(nerrs, 99):.1e} max {nerrs.max():.1e}; support evaluations mean {np.mean(evals):.0f} max {np.max(evals)}")

"""Prototype of the kernel-side cylinder routines (jh_coop.h): the minimum translation of box-cylinder and cylinder-cylinder WITHOUT a polytope, from the structure of the
boundary of K = A (+) (-B).  depth = min over unit d of g(d) = h_A(d) + h_B(-d) - d.(cB - cA), and the minimiser is the outward normal of K at the boundary point nearest to
the centre offset -- a point where K has a two-dimensional patch (an inscribed ball cannot touch a crease).  The patches of a box (+) cylinder: box faces, the caps, box edge (+)
side generator (flat: normals e_i, a, e_i x a); box vertex (+) side (a cylinder of radius r around the line through the vertex: normal = radial direction from that line); box edge (+)
rim (the rim circle swept along the edge: closest point of an ellipse, in the plane across the edge).  Every candidate direction is scored with the full g(d) -- an upper bound of the
depth for ANY d -- so a candidate from the wrong patch can never win wrongly.  Checked against the oracle's GJK + EPA.  usage: python tools/proto/cyl_candidates.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

def unit(v, F):
    n = np.linalg.norm(v)
    return (v / n).astype(F) if n > 1e-12 else None

def ellipse_closest(q0, A, B, F, its=int(os.environ.get("ITS", "6")), ns=int(os.environ.get("NS", "8"))):
    """angles phi at which |q0 + cos(phi) A + sin(phi) B| has a local minimum (a point of the plane against an ellipse with conjugate radii A, B centred at q0; from inside
    an ellipse there can be two): Newton on the derivative from every sample that is lower than both its neighbours"""
    f = []
    for k in range(ns):
        ph = F(k * 2 * np.pi / ns); v = q0 + np.cos(ph) * A + np.sin(ph) * B; f.append(v @ v)
    out = []
    for k in range(ns):
        if f[k] <= f[k - 1] and f[k] <= f[(k + 1) % ns]:
            ph = F(k * 2 * np.pi / ns)
            for _ in range(its):
                c, s_ = np.cos(ph), np.sin(ph)
                v = q0 + c * A + s_ * B; dv = -s_ * A + c * B; ddv = -c * A - s_ * B
                f1 = v @ dv; f2 = dv @ dv + v @ ddv
                if f2 <= 0: break
                ph = ph - f1 / f2
            out.append(ph)
    return out


def box_cylinder(hb, Rb, pb, r, L, Rc, pc, F=np.float64):
    """box (half sizes hb, rotation Rb, centre pb) against cylinder (radius r, half length L along Rc[:, 2], centre pc): (dist <= 0, normal box -> cylinder) or None"""
    E = [Rb[:, k].astype(F) for k in range(3)]; a = Rc[:, 2].astype(F); c = (pc - pb).astype(F); hb = hb.astype(F)
    def g(d):  # overlap along d (pointing from the box to the cylinder)
        da = d @ a
        return sum(hb[k] * abs(d @ E[k]) for k in range(3)) + L * abs(da) + r * np.sqrt(max(F(0), 1 - da * da)) - d @ c
    cands = []
    for k in range(3): cands += [E[k], -E[k]]
    cands += [a, -a]
    for k in range(3):
        u = unit(np.cross(E[k], a), F)
        if u is not None: cands += [u, -u]
    # vertex (+) side: radial direction from the line {v + t a} to the cylinder centre... in K = box (+) cylinder the patch is the cylinder of radius r around the line through the
    # box vertex v; the boundary point nearest to c lies along the radial direction from that line through c, outward normal = that direction
    for sx in (-1, 1):
        for sy in (-1, 1):
            for sz in (-1, 1):
                v = sx * hb[0] * E[0] + sy * hb[1] * E[1] + sz * hb[2] * E[2]
                w = c - v; w = w - (w @ a) * a
                u = unit(w, F)
                if u is not None: cands += [u, -u]
    # edge (+) rim: in the plane across edge direction e_i the rim projects to an ellipse around the projected edge point + cap centre
    u1 = unit(np.cross(a, E[int(np.argmin([abs(a @ e) for e in E]))]), F); u2 = np.cross(a, u1).astype(F)
    for i in range(3):
        j, k = (i + 1) % 3, (i + 2) % 3
        P = lambda x: x - (x @ E[i]) * E[i]
        A_, B_ = r * P(u1), r * P(u2)
        for sj in (-1, 1):
            for sk in (-1, 1):
                for sc in (-1, 1):
                    m = sj * hb[j] * E[j] + sk * hb[k] * E[k] + sc * L * a    # edge midpoint + cap centre (cylinder taken about the origin of K's frame)
                    q0 = P(m - c)
                    for ph in ellipse_closest(q0, A_, B_, F):
                        x = q0 + np.cos(ph) * A_ + np.sin(ph) * B_        # from c to the boundary point, in the plane across the edge
                        u = unit(x, F)
                        if u is not None: cands += [u, -u]
    best, bd = None, None
    for d in cands:
        v = g(d)
        if best is None or v < best: best, bd = v, d
    if best <= 0: return None
    return -best, bd

if __name__ == "__main__":
    from oracle import oracle as O
    rng = np.random.default_rng(5)
    def rot():
        q = rng.standard_normal(4); q /= np.linalg.norm(q); w, x, y, z = q
        return np.array([[1-2*(y*y+z*z), 2*(x*y-z*w), 2*(x*z+y*w)], [2*(x*y+z*w), 1-2*(x*x+z*z), 2*(y*z-x*w)], [2*(x*z-y*w), 2*(y*z+x*w), 1-2*(x*x+y*y)]]), q
    for F in (np.float64, np.float32):
        errs, nerrs, deps, miss, extra = [], [], [], 0, 0
        for trial in range(1500):
            Rb, qb = rot(); Rc, qc = rot()
            hb = rng.uniform(0.01, 0.05, 3); r, L = rng.uniform(0.008, 0.02), rng.uniform(0.004, 0.03)
            pc = rng.standard_normal(3); pc *= rng.uniform(0.0, 0.08) / np.linalg.norm(pc)
            ref = O.collide_pair("box", hb, np.zeros(3), qb, "cylinder", np.array([r, L]), pc, qc)
            got = box_cylinder(hb, Rb, np.zeros(3), F(r), F(L), Rc, pc, F)
            if not ref:
                extra += got is not None and got[0] < -1e-6; continue
            if got is None:
                miss += ref[0][0] < -1e-6; continue
            errs.append(abs(got[0] - ref[0][0])); nerrs.append(1 - got[1] @ ref[0][2]); deps.append(-ref[0][0])
        errs, nerrs, deps = np.array(errs), np.array(nerrs), np.array(deps)
        sh = deps < 3e-3
        print(f"box-cylinder {F.__name__}: {len(errs)} overlapping pairs ({sh.sum()} shallower than 3 mm), missed {miss}, spurious {extra}; |depth error| median {np.median(errs):.1e} 99% {np.percentile(errs, 99):.1e} "
              f"max {errs.max():.1e} m; wrong by > 1e-5 m: {(errs > 1e-5).sum()}; 1 - n.n_ref: median {np.median(nerrs):.1e} 99% {np.percentile(nerrs, 99):.1e} max {nerrs.max():.1e}; shallow: max depth error {errs[sh].max():.1e}")

"""Device model image for the articulated-body rollout engine (`judo_amd/csrc/jh_engine.hip`).

Structure the engine exploits (it holds for leap_cube and fr3_pick): ONE free body (the manipulated cube) plus
articulated "blocks" (serial chains / small trees of hinge or slide joints) hanging off bodies that are welded to
the world.  Consequences used by the kernels:
  * the joint-space inertia is block diagonal: diag(m,m,m,I1,I2,I3) for the free body (its centre of mass sits at
    the body origin) and one small dense block per chain -- no nv x nv factorisation of M is ever needed;
  * static bodies and their collision geoms have constant world poses, folded in here on the host;
  * every contact modelled this round is cube <-> hand geom, so a contact row touches the 6 cube DoF and at most
    one block: the Newton Hessian is an arrow matrix (cube block + per-chain blocks + cube-chain couplings).

Layout of the float (F) and int (I) sections is mirrored by `struct EngineModel` in jh_engine.hip.
"""

from __future__ import annotations

import numpy as np

from judo_amd.models import (
    MINMU,
    MINVAL,
    TASK_KIND,
    _pack,
    clamp_solimp,
    inverse_weights,
    layout,
    quat_mul,
    quat_to_mat,
    solref_to_kb,
)

JFREE, JSLIDE, JHINGE = 0, 2, 3
GBOX, GSPHERE, GCAPSULE = 6, 2, 3
MAX_MOVING, MAX_DOF, MAX_BLOCKS, MAX_BLOCK_DOF, MAX_GEOM, MAX_SITE = 20, 24, 4, 9, 80, 8

# ints per record
BODY_I, GEOM_I, ACT_I, BLOCK_I = 6, 2, 2, 4
# floats per record
BODY_F, DOF_F, ACT_F, GEOM_F, SITE_F = 32, 20, 8, 20, 3
HEADER_I, HEADER_F = 24, 24
# Newton termination on the GPU: |grad|_Minv <= tol * |qfrc_smooth|_Minv (or the expected decrease of a step falls below
# tol^2 of the same scale), at most SOLVER_MAX_ITER iterations (MuJoCo: tolerance 1e-8 in fp64, 100 iterations); the exact line search
# stops at |slope| <= SOLVER_LS_TOL * |slope at 0| (MuJoCo's ls_tolerance default is 0.01 too; 1e-3 costs 2.4 % more on fixed plan inputs
# and does not save a single Newton iteration, tools/diag/ab_fixed_inputs.py).
# tol = 1e-5 is where the returned MPPI nominal stops moving with the tolerance on the recorded 40 plan steps of the headline workload
# (profiles/r02_tolerance_sweep.txt: against tol 1e-6 with a 48-contact pool, 25 of the 40 plans pick another winner at 1e-3, 5 at 1e-4, 2 at 1e-5
# and 2 at 1e-6 -- the rest is the 32-contact pool); it costs 2.7 % over 1e-4 (6.86 instead of 6.57 iterations per step).
# 50 iterations instead of round 1's 20: the random-state sweep tools/diag/fuzz_leap.py has states on which the fp64 oracle needs 21-31; on the headline workload the
# higher cap costs nothing measurable (4e-4 solves per step used to stop at 20).
SOLVER_TOL, SOLVER_MAX_ITER, SOLVER_LS_TOL = 1e-5, 50, 1e-2
# diagnostics: (phase, repeats) read only by -DJH_V2_ABLATE builds of the cooperative kernel (tools/diag/time_ablate.py)
ABLATE = (0, 1)


def _static_world_pose(desc: dict, b: int) -> tuple[np.ndarray, np.ndarray]:
    """World pose of a body whose whole ancestry is joint-free."""
    chain = []
    while b > 0:
        chain.append(b)
        b = desc["bodies"][b]["parent"]
    pos, quat = np.zeros(3), np.array([1.0, 0, 0, 0])
    for bb in reversed(chain):
        body = desc["bodies"][bb]
        pos = pos + quat_to_mat(quat) @ np.array(body["pos"])
        quat = quat_mul(quat, body["quat"])
    return pos, quat / np.linalg.norm(quat)


def engine_structure(desc: dict) -> dict:
    """Classify bodies: static / free / articulated blocks; returns index maps used by the packer and the tests."""
    lay = layout(desc)
    nb = len(desc["bodies"])
    joints_of = lay.body_joints
    is_static = [False] * nb
    is_static[0] = True
    for b in range(1, nb):
        is_static[b] = (len(joints_of[b]) == 0) and is_static[desc["bodies"][b]["parent"]]
    moving = [b for b in range(1, nb) if not is_static[b]]
    for b in moving:
        if len(joints_of[b]) != 1:
            raise NotImplementedError(f"body {desc['bodies'][b]['name']}: the engine needs exactly one joint per moving body")
    free = [b for b in moving if desc["joints"][joints_of[b][0]]["type"] == "free"]
    if len(free) != 1 or moving[0] != free[0]:
        raise NotImplementedError("the engine needs exactly one free body, listed before the articulated bodies")
    fb = desc["bodies"][free[0]]
    if desc["bodies"][fb["parent"]]["parent"] != -1 and fb["parent"] != 0:
        raise NotImplementedError("free body must hang off the world")
    if np.abs(fb["ipos"]).max() > 0 or abs(abs(fb["iquat"][0]) - 1) > 1e-12:
        raise NotImplementedError("free body: centre of mass must be the body origin and the inertia axis-aligned")
    midx = {b: i for i, b in enumerate(moving)}
    # blocks: connected components of articulated bodies (root = body whose parent is static)
    block_of, blocks = {}, []
    for b in moving[1:]:
        p = desc["bodies"][b]["parent"]
        if is_static[p]:
            block_of[b] = len(blocks)
            blocks.append([b])
        else:
            if p not in block_of:
                raise NotImplementedError("articulated body attached to the free body is not supported")
            block_of[b] = block_of[p]
            blocks[block_of[b]].append(b)
    for blk in blocks:
        if blk != list(range(blk[0], blk[0] + len(blk))):
            raise NotImplementedError("bodies of a block must be contiguous (depth-first MJCF order)")
    return dict(layout=lay, is_static=is_static, moving=moving, midx=midx, blocks=blocks, block_of=block_of, free=free[0])


def _mat_to_quat(R: np.ndarray) -> np.ndarray:
    """Rotation matrix -> unit quaternion (w, x, y, z)."""
    t = np.trace(R)
    if t > 0:
        s = np.sqrt(t + 1.0) * 2
        q = np.array([0.25 * s, (R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s])
    else:
        i = int(np.argmax(np.diag(R)))
        j, k = (i + 1) % 3, (i + 2) % 3
        s = np.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0) * 2
        q = np.zeros(4)
        q[0] = (R[k, j] - R[j, k]) / s
        q[1 + i] = 0.25 * s
        q[1 + j] = (R[j, i] + R[i, j]) / s
        q[1 + k] = (R[k, i] + R[i, k]) / s
    return q / np.linalg.norm(q)


def fuse_fixed_bodies(desc: dict) -> dict:
    """Merge every joint-less body whose parent moves into that parent (fr3_pick: `hand` is welded to `fr3_link7`).

    MuJoCo keeps such bodies and lets them ride on the parent; the engine wants exactly one joint per moving body, so the
    child's inertia is added to the parent (parallel-axis, re-diagonalised), its children / geoms / sites are re-expressed
    in the parent frame, and sensors that referenced the body frame get an equivalent site.  Geoms remember the body
    they came from (`orig_body`) because contact regularisation and distance sensors are defined per original body."""
    import copy

    d = copy.deepcopy(desc)
    for g in d["geoms"]:
        g.setdefault("orig_body", g["body"])
    lay = layout(d)
    nb = len(d["bodies"])
    static = [True] + [False] * (nb - 1)
    for b in range(1, nb):
        static[b] = (len(lay.body_joints[b]) == 0) and static[d["bodies"][b]["parent"]]
    victims = [b for b in range(1, nb) if len(lay.body_joints[b]) == 0 and not static[b]]
    if not victims:
        return d
    b = victims[0]
    body = d["bodies"][b]
    p = body["parent"]
    par = d["bodies"][p]
    Rb, pb = quat_to_mat(body["quat"]), np.array(body["pos"])
    # inertia merge in the parent frame
    m1, m2 = par["mass"], body["mass"]
    c1, c2 = np.array(par["ipos"]), pb + Rb @ np.array(body["ipos"])
    R1 = quat_to_mat(par["iquat"])
    R2 = Rb @ quat_to_mat(body["iquat"])
    I1, I2 = R1 @ np.diag(par["inertia"]) @ R1.T, R2 @ np.diag(body["inertia"]) @ R2.T
    m = m1 + m2
    c = (m1 * c1 + m2 * c2) / m if m > 0 else c1
    shift = lambda mm, r: mm * (r @ r * np.eye(3) - np.outer(r, r))  # noqa: E731
    I = I1 + shift(m1, c1 - c) + I2 + shift(m2, c2 - c)
    w, V = np.linalg.eigh(I)
    if np.linalg.det(V) < 0:
        V[:, 2] = -V[:, 2]
    par.update(mass=m, ipos=c.tolist(), iquat=_mat_to_quat(V).tolist(), inertia=w.tolist())
    for ch in d["bodies"]:
        if ch.get("parent") == b:
            ch["pos"] = (pb + Rb @ np.array(ch["pos"])).tolist()
            ch["quat"] = quat_mul(body["quat"], ch["quat"]).tolist()
            ch["parent"] = p
    for coll in (d["geoms"], d["sites"]):
        for g in coll:
            if g["body"] == b:
                g["pos"] = (pb + Rb @ np.array(g["pos"])).tolist()
                g["quat"] = quat_mul(body["quat"], g.get("quat", [1, 0, 0, 0])).tolist()
                g["body"] = p
    # sensors on the body frame -> an equivalent site on the parent
    for sn in d["sensors"]:
        if sn["type"] in ("framepos", "framezaxis") and sn.get("objtype") == "body" and sn["obj"] == b:
            d["sites"].append(dict(name=f"{body['name']}_frame", body=p, pos=pb.tolist(), quat=list(body["quat"])))
            sn["objtype"], sn["obj"] = "site", len(d["sites"]) - 1
    # drop body b, renumber
    remap = {i: (i if i < b else i - 1) for i in range(nb) if i != b}
    del d["bodies"][b]
    for ch in d["bodies"]:
        if ch["parent"] >= 0:
            ch["parent"] = remap[ch["parent"]]
    for coll in (d["geoms"], d["sites"], d["joints"]):
        for g in coll:
            g["body"] = remap[g["body"]]
    for sn in d["sensors"]:
        if sn["type"] in ("framepos", "framezaxis", "framequat") and sn.get("objtype") == "body":
            sn["obj"] = remap[sn["obj"]]
    d["excludes"] = [[remap[a], remap[c2]] for a, c2 in d["excludes"] if a != b and c2 != b]
    d.setdefault("fused", []).append(body["name"])
    return fuse_fixed_bodies(d)


def sensor_reference_frames(desc: dict) -> list[float] | None:
    """World pose of the (static) reference frames of `framepos ... reftype=site` / `framequat ... reftype=body` sensors, resolved on the description as
    compiled (before static bodies are fused away): [p_ref(3), R_ref(9), q_ref(4)], or None when no sensor has a reference frame on the engine's models
    (caltech_leap_cube: cube position in the frame of `grasp_site` on the hand mount, cube orientation relative to the mocap goal body at its model pose --
    the pose every rollout thread's MjData starts with, judo/utils/mj_rollout_backend.py:38-43)."""
    weld = lambda b: all(j["body"] != b for j in desc["joints"]) and (desc["bodies"][b]["parent"] < 0 or weld(desc["bodies"][b]["parent"]))  # noqa: E731
    p_ref, R_ref, q_ref, found = np.zeros(3), np.eye(3), np.array([1.0, 0, 0, 0]), False
    for sn in desc["sensors"]:
        if sn["type"] == "framepos" and sn.get("reftype") == "site" and sn.get("objtype") == "body":
            site = desc["sites"][sn["ref"]]
            if not weld(site["body"]):
                raise NotImplementedError("framepos relative to a moving site")
            bp, bq = _static_world_pose(desc, site["body"])
            p_ref = bp + quat_to_mat(bq) @ np.array(site["pos"])
            R_ref = quat_to_mat(quat_mul(bq, site.get("quat", [1, 0, 0, 0])))
            found = True
        elif sn["type"] == "framequat":
            if sn.get("reftype") == "body":
                if not weld(sn["ref"]):
                    raise NotImplementedError("framequat relative to a moving body")
                q_ref = np.array(_static_world_pose(desc, sn["ref"])[1], dtype=np.float64)
            found = True
    return [*p_ref, *R_ref.reshape(-1), *q_ref] if found else None


def generic_pairs(desc: dict, fused: dict, st: dict, cube_only: bool | None = None) -> list[tuple[int, int]]:
    """Candidate box-box / box-sphere pairs for the generic (reference-kernel) contact path, MuJoCo's static filters
    applied on the ORIGINAL bodies: same welded body, parent-child (unless the parent is welded to the world), excludes.
    leap_cube: restricted to pairs that involve the cube (hand self-collision is out of scope this round)."""
    bodies = desc["bodies"]
    njnt = [0] * len(bodies)
    for j in desc["joints"]:
        njnt[j["body"]] += 1

    def weld(b):
        while b > 0 and njnt[b] == 0:
            b = bodies[b]["parent"]
        return b

    def weld_parent(b):
        w = weld(b)
        return weld(bodies[w]["parent"]) if w > 0 else 0

    excl = {tuple(sorted(e)) for e in desc["excludes"]}
    geoms = fused["geoms"]
    free_fused = st["free"]
    out = []
    for a in range(len(geoms)):
        for b in range(a + 1, len(geoms)):
            ia, ib = a, b
            if geoms[a]["type"] == "capsule" and geoms[b]["type"] == "box":  # (box, capsule) is the order the kernels take, whichever geom the model lists first
                ia, ib = b, a
            ga, gb = geoms[ia], geoms[ib]
            ta, tb = ga["type"], gb["type"]
            if not ((ta == "box" and tb in ("box", "sphere", "capsule")) or (ta == "sphere" and tb == "box") or (ta == "sphere" and tb == "sphere" and cube_only is False)):
                continue  # (sphere-sphere: the fingertips of two fingers; only jh_engine_v5.hip collides the hand with itself)
            ba, bb = ga["orig_body"], gb["orig_body"]
            wa, wb = weld(ba), weld(bb)
            if tb == "capsule" and not (wa == 0 or ga["body"] == free_fused):
                continue  # the arm links' stand-ins meet static geometry and the free body only (oracle/oracle.py::collision_pairs has the reason)
            if wa == wb or tuple(sorted((ba, bb))) in excl:
                continue
            if (weld_parent(ba) == wb and wb != 0) or (weld_parent(bb) == wa and wa != 0):
                continue
            leap = desc.get("family", desc["task"]) == "leap_cube"
            if (leap if cube_only is None else cube_only) and free_fused not in (ga["body"], gb["body"]):
                continue  # (the one-lane reference kernel and jh_engine_v2.hip model the cube's contacts only; jh_engine_v5.hip adds the hand's own)
            out.append((ia, ib))
    # order by importance for the fixed-capacity contact pools: pairs with the free body first, then pairs against static
    # geometry, pairs between two articulated bodies (e.g. the two fingers' pad stacks) last -- those are the ones dropped on overflow
    def rank(pr):
        ba, bb = geoms[pr[0]]["body"], geoms[pr[1]]["body"]
        if free_fused in (ba, bb):
            return 0
        return 1 if (st["is_static"][ba] or st["is_static"][bb]) else 2
    out.sort(key=rank)  # stable: original order within a class
    return out


def kernel_stand_ins(desc: dict) -> dict:
    """The description as the leap KERNEL collides it (jh_engine_v5.hip has box and sphere narrow phases only): caltech_leap_cube's fingertip cylinders
    (judo/models/xml/caltech_leap_components/leap_rh.xml:131,175,219,259: r = 14 mm, half length 7 mm, the tip's sphere 7 mm further out) become spheres of the
    cylinder's radius at its centre -- a STATED DEVIATION of the kernel, not of the model: the description and the oracle keep the cylinder (MuJoCo's general convex
    collider, restated as GJK + EPA in oracle/jo_engine.c::collide_convex), and tests/test_gpu_leap.py measures what the stand-in costs at trajectory level."""
    if not any(g["type"] == "cylinder" for g in desc["geoms"]) or desc.get("family", desc["task"]) != "leap_cube":
        return desc
    out = dict(desc)
    out["geoms"] = [dict(g, type="sphere", size=[g["size"][0]], substitute_for_cylinder=list(g["size"])) if g["type"] == "cylinder" else g for g in desc["geoms"]]
    return out


def pack_engine_model(desc: dict) -> bytes:
    desc = kernel_stand_ins(desc)
    orig = desc
    ref_frames = sensor_reference_frames(orig)
    dofw_o, bodyw_o = inverse_weights(orig)
    desc = fuse_fixed_bodies(orig)
    st = engine_structure(desc)
    lay, moving, midx, blocks = st["layout"], st["moving"], st["midx"], st["blocks"]
    o = desc["option"]
    dofw = dofw_o  # dof order is unchanged by fusing
    bodyw_geom = lambda g: bodyw_o[g.get("orig_body", g["body"])][0]  # noqa: E731  (contact diagApprox uses the geom's ORIGINAL body)
    _, bodyw = inverse_weights(desc)
    NM, NBLK = len(moving), len(blocks)
    if NM > MAX_MOVING or lay.nv > MAX_DOF or NBLK > MAX_BLOCKS or max(len(b) for b in blocks) > MAX_BLOCK_DOF:
        raise NotImplementedError("model exceeds the engine's compile-time limits")

    cube_geoms = [g for g in desc["geoms"] if g["body"] == st["free"]]
    if len(cube_geoms) != 1 or cube_geoms[0]["type"] != "box" or np.abs(cube_geoms[0]["pos"]).max() > 0:
        raise NotImplementedError("free body must carry exactly one box geom centred on the body origin")
    cube = cube_geoms[0]
    # contacts modelled: cube geom vs every other collision geom (hand self-collision: out of scope this round)
    others = [g for g in desc["geoms"] if g["body"] != st["free"] and g["type"] in ("box", "sphere")]
    # geoms sorted by owning body so that the kernel loads each body pose once
    others.sort(key=lambda g: (-1 if st["is_static"][g["body"]] else midx[g["body"]]))
    if len(others) > MAX_GEOM:
        raise NotImplementedError("too many collision geoms")

    I: list[int] = [0] * HEADER_I
    F: list[float] = [0.0] * HEADER_F
    integ = {"euler": 0, "implicitfast": 3}[o["integrator"]]
    cone = {"pyramidal": 0, "elliptic": 1}[o["cone"]]
    if integ != 3:
        raise NotImplementedError("engine kernels implement the implicitfast integrator (leap_cube / fr3_pick)")
    sites = desc["sites"]
    sens = desc["sensors"]
    I[0:12] = [NM, NBLK, lay.nv, lay.nq, lay.nu, len(others), len(sites), lay.ns, integ, cone, len(sens), 0]
    # contact solver parameters are shared by all geoms in these models (checked)
    uniform = all(g["solref"] == cube["solref"] and g["solimp"] == cube["solimp"] for g in others)
    for g in others:
        if g["condim"] != 3 or g["margin"] != 0 or g["gap"] != 0:
            raise NotImplementedError("engine assumes condim 3 and zero margin/gap")
    cK, cB = solref_to_kb(cube["solref"], cube["solimp"], o["timestep"])
    F[0:3] = [o["timestep"], o["impratio"], SOLVER_TOL]
    F[3:6] = o["gravity"]
    F[6:8] = [cK, cB]
    F[8:13] = clamp_solimp(cube["solimp"])
    fbody = desc["bodies"][st["free"]]
    F[13:17] = [fbody["mass"], *fbody["inertia"]]
    F[17:20] = cube["size"]
    F[20] = float(np.linalg.norm(cube["size"]))
    F[21] = bodyw_geom(cube)
    F[22] = float(SOLVER_MAX_ITER)
    F[23] = float(SOLVER_LS_TOL)

    # ---- moving bodies
    for i, b in enumerate(moving):
        body = desc["bodies"][b]
        j = lay.body_joints[b][0]
        jn = desc["joints"][j]
        jt = {"free": JFREE, "slide": JSLIDE, "hinge": JHINGE}[jn["type"]]
        p = body["parent"]
        if st["is_static"][p]:
            ppos, pquat = _static_world_pose(desc, p)
            lpos = ppos + quat_to_mat(pquat) @ np.array(body["pos"])
            lquat = quat_mul(pquat, body["quat"])
            par = -1
        else:
            lpos, lquat, par = np.array(body["pos"]), np.array(body["quat"]), midx[p]
        if jt != JFREE and np.abs(jn["pos"]).max() > 0:
            raise NotImplementedError("engine assumes joint anchors at the body origin")
        blk = st["block_of"].get(b, -1)
        I += [par, jt, lay.jnt_dofadr[j], lay.jnt_qposadr[j], blk, (b - blocks[blk][0]) if blk >= 0 else 0]
        lR = quat_to_mat(lquat / np.linalg.norm(lquat))
        iR = quat_to_mat(body["iquat"])
        F += [*lpos, *lR.reshape(-1), body["mass"], *body["ipos"], *iR.reshape(-1), *body["inertia"], *jn["axis"], bodyw[b][0]]
    assert len(F) == HEADER_F + NM * BODY_F and len(I) == HEADER_I + NM * BODY_I
    # ---- blocks
    for blk in blocks:
        d0 = lay.jnt_dofadr[lay.body_joints[blk[0]][0]]
        I += [midx[blk[0]], len(blk), d0, len(blk)]
    # ---- dofs
    act_kv = np.zeros(lay.nv)
    for a in desc["actuators"]:
        act_kv[lay.jnt_dofadr[a["joint"]]] += a["kv"] * a["gear"] ** 2
    for d in range(lay.nv):
        jn = desc["joints"][lay.dof_jnt[d]]
        fl = jn["frictionloss"]
        # friction-loss row: pos = 0 -> impedance = dmin, K = 0, B from solreffriction, R = (1-d)/d * invweight
        si = clamp_solimp(jn["solimpfriction"])
        imp0 = si[0] if not (si[0] == si[1] or si[2] <= MINVAL) else 0.5 * (si[0] + si[1])
        _, fB = solref_to_kb(jn["solreffriction"], jn["solimpfriction"], o["timestep"])
        fR = max(MINVAL, (1 - imp0) / imp0 * dofw[d])
        lK, lB = solref_to_kb(jn["solreflimit"], jn["solimplimit"], o["timestep"])
        rng = jn["range"] if (jn["range"] is not None and jn["type"] != "free") else None
        frc = jn["actuatorfrcrange"]
        F += [jn["damping"], jn["armature"], fl, fB, 1.0 / fR if fl > 0 else 0.0, dofw[d], float(rng is not None), *(rng or (0, 0)), lK, lB,
              *clamp_solimp(jn["solimplimit"]), float(frc is not None), *(frc or (0, 0)), act_kv[d]]
    assert len(F) == HEADER_F + NM * BODY_F + lay.nv * DOF_F
    # ---- actuators (position servos on joints)
    for a in desc["actuators"]:
        j = a["joint"]
        I += [lay.jnt_dofadr[j], lay.jnt_qposadr[j]]
        if a["forcerange"] is not None:
            raise NotImplementedError("actuator forcerange")
        F += [a["kp"] * a["gear"], a["kv"] * a["gear"], float(a["ctrlrange"] is not None), *(a["ctrlrange"] or (0, 0)), 0, 0, 0]
    # ---- collision geoms (vs the cube)
    mu_cube = cube["friction"][0]
    for g in others:
        b = g["body"]
        if st["is_static"][b]:
            bpos, bquat = _static_world_pose(desc, b)
            pos = bpos + quat_to_mat(bquat) @ np.array(g["pos"])
            R = quat_to_mat(quat_mul(bquat, g["quat"]))
            mb = -1
        else:
            pos, R, mb = np.array(g["pos"]), quat_to_mat(g["quat"]), midx[b]
        size = (list(g["size"]) + [0, 0, 0])[:3]
        rb = size[0] if g["type"] == "sphere" else float(np.linalg.norm(size))
        mu = max(MINMU, max(g["friction"][0], mu_cube))
        I += [mb, GBOX if g["type"] == "box" else GSPHERE]
        F += [*size, *pos, *R.reshape(-1), rb, mu, bodyw_geom(g), max(MINMU, g["friction"][0]), 0]
    # ---- sites + sensors
    for s in sites:
        b = s["body"]
        if st["is_static"][b]:  # world-fixed site: body index -1, world position
            bp, bq = _static_world_pose(desc, b)
            I += [-1]
            F += list(bp + quat_to_mat(bq) @ np.array(s["pos"]))
            continue
        I += [midx[b]]
        F += list(s["pos"])
    for s in sens:
        if s["type"] == "framepos" and s.get("reftype") is not None and s["objtype"] == "body":
            I += [6, midx[s["obj"]], s["adr"]]  # body position in the static reference frame (tail block I[18])
        elif s["type"] == "framequat":
            I += [7, midx[s["obj"]], s["adr"]]  # body orientation relative to the static reference quaternion
        elif s["type"] == "framepos" and s["objtype"] == "site":
            I += [0, s["obj"], s["adr"]]
        elif s["type"] == "framepos":
            I += [1, midx[s["obj"]], s["adr"]]
        elif s["type"] == "jointpos":
            I += [2, lay.jnt_qposadr[s["obj"]], s["adr"]]
        elif s["type"] == "framezaxis" and s.get("objtype") == "body":
            I += [3, midx[s["obj"]], s["adr"]]
        elif s["type"] == "framezaxis":
            I += [5, s["obj"], s["adr"]]  # z axis of a site frame (generic section carries the rotation)
        else:
            I += [4, 0, s["adr"]]  # geom distance: generic section
    # ---- cooperative kernel (16 lanes per rollout, one lane per finger link): per-lane list of geoms to broad-phase.
    # lane l tests the geoms of moving body 1+l; static geoms are dealt greedily to the least-loaded lanes.
    if NM - 1 == 16 and NBLK == 4 and uniform:  # leap_cube layout of the cooperative kernel (needs uniform contact parameters)
        lists: list[list[int]] = [[] for _ in range(16)]
        for gi, g in enumerate(others):
            b = g["body"]
            if not st["is_static"][b]:
                lists[midx[b] - 1].append(gi)
        for gi, g in enumerate(others):
            if st["is_static"][g["body"]]:
                min(lists, key=len).append(gi)
        lgm = max(len(x) for x in lists)
        I[11], I[12] = len(I), lgm
        for x in lists:
            I += x + [-1] * (lgm - len(x))
        # hand self-collision (jh_engine_v5.hip): candidate geom pairs between hand bodies after MuJoCo's static filters (same welded body, parent-child,
        # the 18 excludes), grouped by body pair; a bounding sphere per body (static geometry: one sphere in world coordinates)
        oidx = {id(g): i for i, g in enumerate(others)}
        og = [g for g in desc["geoms"] if g["type"] in ("box", "sphere")]
        hh = [(a, b) for a, b in generic_pairs(orig, dict(desc, geoms=og), st, cube_only=False) if st["free"] not in (og[a]["body"], og[b]["body"])]
        # hand "bodies" of the self-collision tables: 1..16 = finger links; the static geometry is one body (0) when all of it collides with the same links
        # (leap_cube: the palm), else one body per set of static geoms with the same partners (caltech_leap_cube: floor + hand mount, palm): 0, 17, 18, 19
        link_code = lambda g: midx[g["body"]]  # noqa: E731
        partners: dict[int, set] = {}
        for a, b in hh:
            for x, y in ((og[a], og[b]), (og[b], og[a])):
                if st["is_static"][x["body"]] and not st["is_static"][y["body"]]:
                    partners.setdefault(x["body"], set()).add(link_code(y))
        sgroups: list[frozenset] = []
        for g in others:
            if st["is_static"][g["body"]]:
                key = frozenset(partners.get(g["body"], ()))
                if key not in sgroups:
                    sgroups.append(key)
        if len(sgroups) > 4:
            raise NotImplementedError("more than four groups of static collision geometry")
        SCODES = [0, 17, 18, 19]
        NBC = 20  # body codes in the image (jh_engine_v5.hip NBC)
        code = lambda g: SCODES[sgroups.index(frozenset(partners.get(g["body"], ())))] if st["is_static"][g["body"]] else link_code(g)  # noqa: E731
        is_static_code = lambda c: c == 0 or c > 16  # noqa: E731
        groups: dict[tuple[int, int], list[tuple[int, int]]] = {}
        for a, b in hh:
            ga, gb = og[a], og[b]
            if is_static_code(code(gb)) or (not is_static_code(code(ga)) and code(ga) > code(gb)):  # side A: static geometry, else the lower link
                ga, gb = gb, ga
            if is_static_code(code(gb)):
                continue  # (static against static never collides)
            groups.setdefault((code(ga), code(gb)), []).append((oidx[id(ga)], oidx[id(gb)]))
        # MuJoCo's static filters act on bodies and every collision geom of the hand has the same contype / conaffinity: a surviving body pair collides
        # all of A's geoms with all of B's, and the geoms of a body are contiguous in `others` -> the image carries the body pairs and one
        # (first geom, count) range per body, no geom-pair table
        rng = {}
        for c in range(NBC):
            idxs = [i for i, g in enumerate(others) if code(g) == c]
            assert not idxs or idxs == list(range(idxs[0], idxs[0] + len(idxs))), "geoms of a body must be contiguous"
            rng[c] = (idxs[0], len(idxs)) if idxs else (0, 0)
        for (ca, cb), lst in groups.items():
            assert len(lst) == rng[ca][1] * rng[cb][1] and rng[ca][1] + rng[cb][1] <= 16, (ca, cb, len(lst))
        I[15], I[17] = len(I), len(groups)
        for (ca, cb) in groups:
            I += [ca, cb]
        for c in range(NBC):
            I += list(rng[c])
        I[16] = len(F)
        for c in range(NBC):  # per hand body (static geometry: in world coordinates): bounding sphere and bounding box of its collision geoms, body frame
            gs = [g for g in others if code(g) == c]
            if not gs:
                F += [0.0] * 8
                continue
            corners = []
            for g in gs:
                if is_static_code(c):
                    bpos, bquat = _static_world_pose(desc, g["body"])
                    gp, gR = bpos + quat_to_mat(bquat) @ np.array(g["pos"]), quat_to_mat(quat_mul(bquat, g["quat"]))
                else:
                    gp, gR = np.array(g["pos"]), quat_to_mat(g["quat"])
                hs = np.array([g["size"][0]] * 3) if g["type"] == "sphere" else np.array(g["size"][:3])
                for sx in (-1, 1):
                    for sy in (-1, 1):
                        for sz in (-1, 1):
                            corners.append(gp + gR @ (hs * np.array([sx, sy, sz])))
            corners = np.array(corners)
            lo, hi = corners.min(0), corners.max(0)
            ctr, half = 0.5 * (lo + hi), 0.5 * (hi - lo)
            F += [*ctr, float(np.linalg.norm(half)), *half, 0.0]
    # ---- generic sections (reference kernel): every collision geom incl. the cube, explicit candidate pairs, joint
    # equalities, sensor frames with orientation, geom-distance sensors
    allg = [g for g in desc["geoms"] if g["type"] in ("box", "sphere", "capsule")]
    gidx = {id(g): i for i, g in enumerate(allg)}
    pairs_all = generic_pairs(orig, dict(desc, geoms=allg), st)
    frames = []
    for sx in sites:
        if st["is_static"][sx["body"]]:  # world-fixed frame
            bp, bq = _static_world_pose(desc, sx["body"])
            frames.append(dict(body=-1, pos=list(bp + quat_to_mat(bq) @ np.array(sx["pos"])), quat=list(quat_mul(bq, sx.get("quat", [1, 0, 0, 0])))))
        else:
            frames.append(dict(body=midx[sx["body"]], pos=sx["pos"], quat=sx.get("quat", [1, 0, 0, 0])))
    dists = [sx for sx in sens if sx["type"] == "distance"]
    eqs = desc["equalities"]
    I[13], I[14] = len(I), len(F)
    I[20], I[21] = ABLATE
    I += [len(allg), len(pairs_all), len(eqs), len(frames), len(dists), len(sens), 0, 0]
    for g in allg:
        b = g["body"]
        if st["is_static"][b]:
            bpos, bquat = _static_world_pose(desc, b)
            pos = bpos + quat_to_mat(bquat) @ np.array(g["pos"])
            R = quat_to_mat(quat_mul(bquat, g["quat"]))
            mb = -1
        else:
            pos, R, mb = np.array(g["pos"]), quat_to_mat(g["quat"]), midx[b]
        size = (list(g["size"]) + [0, 0, 0])[:3]
        rb = size[0] if g["type"] == "sphere" else (size[0] + size[1] if g["type"] == "capsule" else float(np.linalg.norm(size)))
        I += [mb, {"box": GBOX, "sphere": GSPHERE, "capsule": GCAPSULE}[g["type"]]]
        F += [*size, *pos, *R.reshape(-1), rb, max(MINMU, g["friction"][0]), bodyw_geom(g), 0, 0]
    for g in allg:  # per-geom solver parameters, mixed per contact (solref/solimp averaged, friction max)
        F += [*g["solref"], *clamp_solimp(g["solimp"]), 0.0]
    for a, b in pairs_all:
        I += [a, b]
    for e in eqs:
        j1, j2 = e["joint1"], e["joint2"]
        d1, d2 = lay.jnt_dofadr[j1], lay.jnt_dofadr[j2]
        if any(abs(x) > 0 for x in e["polycoef"][2:]):
            raise NotImplementedError("joint equality: only linear coupling")
        eK, eB = solref_to_kb(e["solref"], e["solimp"], o["timestep"])
        blk1 = st["block_of"][desc["joints"][j1]["body"]]
        if blk1 != st["block_of"][desc["joints"][j2]["body"]]:
            raise NotImplementedError("joint equality across blocks")
        d0 = lay.jnt_dofadr[lay.body_joints[blocks[blk1][0]][0]]
        I += [d1, d2, blk1, d1 - d0, d2 - d0]
        F += [e["polycoef"][0], e["polycoef"][1], eK, eB, *clamp_solimp(e["solimp"]), dofw[d1] + dofw[d2], 0, 0]
    for fr in frames:
        I += [fr["body"]]
        F += [*fr["pos"], *quat_to_mat(fr["quat"]).reshape(-1)]
    glists: list[int] = []
    for sx in dists:
        la = [gidx[id(g)] for g in allg if g["orig_body"] == sx["body1"] and g["type"] == "box"]
        lb = [gidx[id(g)] for g in allg if g["orig_body"] == sx["body2"] and g["type"] == "box"]
        I += [len(glists), len(la), len(glists) + len(la), len(lb)]
        glists += la + lb
        F += [sx["cutoff"]]
    idist = 0
    for sx in sens:  # generic sensor table: (type, object, aux, address)
        if sx["type"] == "framepos" and sx.get("reftype") is not None and sx["objtype"] == "body":
            I += [6, midx[sx["obj"]], 0, sx["adr"]]
        elif sx["type"] == "framequat":
            I += [7, midx[sx["obj"]], 0, sx["adr"]]
        elif sx["type"] == "framepos" and sx.get("objtype") == "site":
            I += [0, sx["obj"], 0, sx["adr"]]
        elif sx["type"] == "framepos":
            I += [1, midx[sx["obj"]], 0, sx["adr"]]
        elif sx["type"] == "jointpos":
            I += [2, lay.jnt_qposadr[sx["obj"]], 0, sx["adr"]]
        elif sx["type"] == "framezaxis" and sx.get("objtype") == "site":
            I += [5, sx["obj"], 0, sx["adr"]]
        elif sx["type"] == "framezaxis":
            I += [3, midx[sx["obj"]], 0, sx["adr"]]
        else:
            I += [4, idist, 0, sx["adr"]]
            idist += 1
    I += glists
    if ref_frames is not None:  # reference frames of the sensors (jh_engine_v5.hip, caltech_leap_cube layout)
        I[18] = len(F)
        F += ref_frames
    ntp = 9 if desc.get("family", desc["task"]) == "leap_cube" else 22
    return _pack(TASK_KIND[desc.get("family", desc["task"])], lay, ntp, F, I)


def geom_order(desc: dict) -> list[str]:
    """Names of the collision geoms in the order the engine tests them (for diagnostics)."""
    st = engine_structure(desc)
    others = [g for g in desc["geoms"] if g["body"] != st["free"] and g["type"] in ("box", "sphere")]
    others.sort(key=lambda g: (-1 if st["is_static"][g["body"]] else st["midx"][g["body"]]))
    return [g["name"] for g in others]

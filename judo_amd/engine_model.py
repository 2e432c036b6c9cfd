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
GBOX, GSPHERE = 6, 2
MAX_MOVING, MAX_DOF, MAX_BLOCKS, MAX_BLOCK_DOF, MAX_GEOM, MAX_SITE = 20, 24, 4, 9, 80, 8

# ints per record
BODY_I, GEOM_I, ACT_I, BLOCK_I = 6, 2, 2, 4
# floats per record
BODY_F, DOF_F, ACT_F, GEOM_F, SITE_F = 32, 20, 8, 20, 3
HEADER_I, HEADER_F = 24, 24
# Newton termination on the GPU: |grad|_Minv <= tol * |qfrc_smooth|_Minv (or the expected decrease of a step falls below
# tol^2 of the same scale), at most SOLVER_MAX_ITER iterations (MuJoCo: tolerance 1e-8 in fp64, 100 iterations)
SOLVER_TOL, SOLVER_MAX_ITER, SOLVER_LS_TOL = 1e-4, 20, 1e-3


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


def pack_engine_model(desc: dict) -> bytes:
    st = engine_structure(desc)
    lay, moving, midx, blocks = st["layout"], st["moving"], st["midx"], st["blocks"]
    o = desc["option"]
    dofw, bodyw = inverse_weights(desc)
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
    if cone != 1 or integ != 3:
        raise NotImplementedError("engine kernels implement implicitfast + elliptic cones (leap_cube / fr3_pick)")
    sites = desc["sites"]
    sens = desc["sensors"]
    I[0:12] = [NM, NBLK, lay.nv, lay.nq, lay.nu, len(others), len(sites), lay.ns, integ, cone, len(sens), 0]
    # contact solver parameters are shared by all geoms in these models (checked)
    for g in others:
        if g["solref"] != cube["solref"] or g["solimp"] != cube["solimp"] or g["condim"] != 3 or g["margin"] != 0 or g["gap"] != 0:
            raise NotImplementedError("engine assumes uniform geom solref/solimp, condim 3, zero margin/gap")
    cK, cB = solref_to_kb(cube["solref"], cube["solimp"], o["timestep"])
    F[0:3] = [o["timestep"], o["impratio"], SOLVER_TOL]
    F[3:6] = o["gravity"]
    F[6:8] = [cK, cB]
    F[8:13] = clamp_solimp(cube["solimp"])
    fbody = desc["bodies"][st["free"]]
    F[13:17] = [fbody["mass"], *fbody["inertia"]]
    F[17:20] = cube["size"]
    F[20] = float(np.linalg.norm(cube["size"]))
    F[21] = bodyw[st["free"]][0]
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
        F += [*size, *pos, *R.reshape(-1), rb, mu, bodyw[b][0], 0, 0]
    # ---- sites + sensors
    for s in sites:
        b = s["body"]
        if st["is_static"][b]:
            raise NotImplementedError("site on a static body")
        I += [midx[b]]
        F += list(s["pos"])
    for s in sens:
        if s["type"] == "framepos" and s["objtype"] == "site":
            I += [0, s["obj"], s["adr"]]
        elif s["type"] == "framepos":
            I += [1, midx[s["obj"]], s["adr"]]
        elif s["type"] == "jointpos":
            I += [2, lay.jnt_qposadr[s["obj"]], s["adr"]]
        elif s["type"] == "framezaxis":
            I += [3, midx[s["obj"]], s["adr"]]
        else:
            I += [4, 0, s["adr"]]  # geom distance: not produced by the engine yet (fr3_pick, next round)
    # ---- cooperative kernel (16 lanes per rollout, one lane per finger link): per-lane list of geoms to broad-phase.
    # lane l tests the geoms of moving body 1+l; static geoms are dealt greedily to the least-loaded lanes.
    if NM - 1 <= 16:
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
    ntp = 9 if desc["task"] == "leap_cube" else 22
    return _pack(TASK_KIND[desc["task"]], lay, ntp, F, I)


def geom_order(desc: dict) -> list[str]:
    """Names of the collision geoms in the order the engine tests them (for diagnostics)."""
    st = engine_structure(desc)
    others = [g for g in desc["geoms"] if g["body"] != st["free"] and g["type"] in ("box", "sphere")]
    others.sort(key=lambda g: (-1 if st["is_static"][g["body"]] else st["midx"][g["body"]]))
    return [g["name"] for g in others]

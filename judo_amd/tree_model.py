"""Model image of a floating-base robot on a ground plane for the cooperative tree kernel (`csrc/jh_engine_v4.hip`).

Structure accepted (the Spot model, judo/models/xml/spot_primitive/robot.xml): one free base body; serial chains of hinge joints hanging
off the base (4 legs x 3, one arm x 7), one body per joint, joint anchors at the body origins; position servos with ctrl and force
ranges; friction loss and limits on every joint; sphere / capsule / box collision geoms against ONE static plane and against each other -- the
robot-robot geom pairs MuJoCo's static filters leave (same body, parent-child, the <exclude> pairs of spot_primitive/contact.xml:4-14; 287 for Spot) are
listed in the image; pyramidal cones.  Contact parameters are mixed on the host: the plane has the higher priority in the shipped model, so its
friction / solref / solimp apply to plane contacts (mj_contactParam); the robot geoms all carry the same solref / solimp / priority (checked), so a
robot-robot pair takes the larger of the two frictions and the stored solref / solimp.

Float image F / int image I (little-endian fp32 / int32), see the enums at the top of jh_engine_v4.hip.
"""

from __future__ import annotations

import numpy as np

from judo_amd.models import MINVAL, clamp_solimp, inverse_weights, layout, quat_to_mat, solref_to_kb

TH_F, TH_I = 32, 16           # header sizes
TD_F, TD_I = 56, 4            # per joint dof
TG_F, TG_I = 28, 2            # per geom
TS_F, TS_I = 16, 4            # per sensor
SOLVER_TOL, SOLVER_MAX_ITER, SOLVER_LS_TOL = 1e-4, 20, 1e-2
GTYPE = {"sphere": 2, "capsule": 3, "box": 6}
MAX_JOINTS, MAX_GEOMS, MAX_DEPTH = 25, 31, 7


def tree_structure(desc: dict) -> dict:
    """Base body, joint bodies in dof order, per-joint (parent joint, chain start, depth)."""
    lay = layout(desc)
    bodies, joints = desc["bodies"], desc["joints"]
    free = [j for j, jn in enumerate(joints) if jn["type"] == "free"]
    if len(free) != 1 or lay.jnt_dofadr[free[0]] != 0:
        raise NotImplementedError("tree kernel: exactly one free joint, first in the dof order")
    base = joints[free[0]]["body"]
    if bodies[base]["parent"] != 0 and any(len(lay.body_joints[b]) for b in [bodies[base]["parent"]]):
        raise NotImplementedError("tree kernel: the free body must hang off the world")
    hinges = [j for j, jn in enumerate(joints) if jn["type"] != "free"]
    body_of = {}
    info = []
    for k, j in enumerate(hinges):
        jn = joints[j]
        if jn["type"] != "hinge" or np.abs(jn["pos"]).max() > 0 or lay.jnt_dofadr[j] != 6 + k or len(lay.body_joints[jn["body"]]) != 1:
            raise NotImplementedError("tree kernel: one hinge per body, anchored at the body origin, dofs in joint order")
        body_of[jn["body"]] = k
    for k, j in enumerate(hinges):
        b = joints[j]["body"]
        p = bodies[b]["parent"]
        if p == base:
            info.append(dict(parent=-1, start=k, depth=0))
        elif p in body_of:
            pk = body_of[p]
            info.append(dict(parent=pk, start=info[pk]["start"], depth=info[pk]["depth"] + 1))
            if pk != k - 1:
                raise NotImplementedError("tree kernel: chains must be contiguous in the dof order")
        else:
            raise NotImplementedError("tree kernel: every jointed body must hang off the base or off another jointed body")
    if len(hinges) > MAX_JOINTS or max(i["depth"] for i in info) >= MAX_DEPTH:
        raise NotImplementedError("tree kernel: at most 25 joints in chains of at most 7")
    return dict(layout=lay, base=base, hinges=hinges, body_of=body_of, info=info)


def robot_pairs(desc: dict, robot_geoms: list) -> list[tuple[int, int]]:
    """Robot-robot geom pairs (indices into `robot_geoms`, g1 < g2) after MuJoCo's static filters: not on the same body, not parent and child, not excluded.
    The contype / conaffinity test passes for every pair by construction: `tools/compile_mjcf.py` refuses a collision geom whose masks are not 1 / 1, and
    `tree_structure` above has one joint per body (no welded bodies: body == weld id)."""
    bodies, geoms = desc["bodies"], desc["geoms"]
    excl = {tuple(sorted(e)) for e in desc.get("excludes", [])}
    pairs = []
    for i, ga in enumerate(robot_geoms):
        for j in range(i + 1, len(robot_geoms)):
            gb = robot_geoms[j]
            b1, b2 = ga["body"], gb["body"]
            if b1 == b2 or tuple(sorted((b1, b2))) in excl or bodies[b1]["parent"] == b2 or bodies[b2]["parent"] == b1:
                continue
            pairs.append((i, j))
    return pairs


def bounding_radius(g: dict) -> float:
    size = list(g["size"]) + [0, 0, 0]
    return {"sphere": size[0], "capsule": size[0] + size[1], "box": float(np.linalg.norm(size[:3]))}[g["type"]]


def pack_tree_model(desc: dict) -> tuple[np.ndarray, np.ndarray]:
    st = tree_structure(desc)
    lay, base, hinges, body_of, info = st["layout"], st["base"], st["hinges"], st["body_of"], st["info"]
    bodies, joints, geoms, o = desc["bodies"], desc["joints"], desc["geoms"], desc["option"]
    if o["cone"] != "pyramidal" or o["integrator"] != "implicitfast":
        raise NotImplementedError("tree kernel: pyramidal cones, implicitfast integrator")
    dofw, bodyw = inverse_weights(desc)
    nj = len(hinges)
    planes = [g for g in geoms if g["type"] == "plane"]
    if len(planes) != 1 or len(lay.body_joints[planes[0]["body"]]) != 0:
        raise NotImplementedError("tree kernel: exactly one static plane")
    plane = planes[0]
    pb = bodies[plane["body"]]
    Rp = quat_to_mat(pb["quat"]) @ quat_to_mat(plane["quat"])
    ppos = np.array(pb["pos"]) + quat_to_mat(pb["quat"]) @ np.array(plane["pos"])
    robot_geoms = [g for g in geoms if g is not plane]
    if len(robot_geoms) > MAX_GEOMS:
        raise NotImplementedError("tree kernel: too many collision geoms")
    act_of = {a["joint"]: a for a in desc["actuators"]}
    if len(act_of) != len(desc["actuators"]) or any(a["gear"] != 1.0 for a in desc["actuators"]):
        raise NotImplementedError("tree kernel: at most one unit-gear position servo per joint")

    sensors = desc.get("sensors", [])
    pairs = robot_pairs(desc, robot_geoms)
    if pairs:
        ref = robot_geoms[0]
        for g in robot_geoms:  # one set of mixing inputs for every robot-robot pair (see the module docstring)
            if list(g["solref"]) != list(ref["solref"]) or list(g["solimp"]) != list(ref["solimp"]) or g.get("priority", 0) != ref.get("priority", 0) or g.get("solmix", 1.0) != ref.get("solmix", 1.0):
                raise NotImplementedError("tree kernel: robot geoms with different solref / solimp / priority / solmix")
        if np.abs(bodyw[plane["body"]]).max() != 0:
            raise NotImplementedError("tree kernel: the plane must sit on a static body (zero inverse weight)")
    F = np.zeros(TH_F + nj * TD_F + len(robot_geoms) * TG_F + len(sensors) * TS_F, dtype=np.float32)
    I = np.zeros(TH_I + nj * TD_I + len(robot_geoms) * TG_I + len(sensors) * TS_I + len(pairs), dtype=np.int32)
    bb = bodies[base]
    F[0:8] = [o["timestep"], o["impratio"], SOLVER_TOL, SOLVER_MAX_ITER, SOLVER_LS_TOL, *o["gravity"]]
    F[8:14] = [*ppos, *Rp[:, 2]]                       # plane point, plane normal
    F[14] = bb["mass"]
    F[15:18] = bb["ipos"]
    F[18:27] = quat_to_mat(bb["iquat"]).reshape(-1)
    F[27:30] = bb["inertia"]
    I[0:4] = [nj, len(robot_geoms), lay.nq, lay.nv]
    for k, j in enumerate(hinges):
        jn, b = joints[j], bodies[joints[j]["body"]]
        d = 6 + k
        f = F[TH_F + k * TD_F: TH_F + (k + 1) * TD_F]
        f[0:3] = b["pos"]
        f[3:12] = quat_to_mat(b["quat"]).reshape(-1)
        f[12:15] = jn["axis"]
        f[15] = b["mass"]
        f[16:19] = b["ipos"]
        f[19:28] = quat_to_mat(b["iquat"]).reshape(-1)
        f[28:31] = b["inertia"]
        f[31], f[32] = jn["damping"], jn["armature"]
        fl = jn["frictionloss"]
        si = clamp_solimp(jn["solimpfriction"])
        imp0 = si[0] if not (si[0] == si[1] or si[2] <= MINVAL) else 0.5 * (si[0] + si[1])
        _, fB = solref_to_kb(jn["solreffriction"], jn["solimpfriction"], o["timestep"])
        fR = max(MINVAL, (1 - imp0) / imp0 * dofw[d])
        f[33:36] = [fl, fB, 1.0 / fR if fl > 0 else 0.0]
        f[36] = dofw[d]
        rng = jn["range"]
        lK, lB = solref_to_kb(jn["solreflimit"], jn["solimplimit"], o["timestep"])
        f[37:42] = [float(rng is not None), *(rng or (0, 0)), lK, lB]
        f[42:47] = clamp_solimp(jn["solimplimit"])
        a = act_of.get(j)
        if a is not None:
            cr, fr = a["ctrlrange"], a["forcerange"]
            f[47:55] = [a["kp"], a["kv"], float(cr is not None), *(cr or (0, 0)), float(fr is not None), *(fr or (0, 0))]
        if jn["actuatorfrcrange"] is not None:
            raise NotImplementedError("tree kernel: joint-level actuatorfrcrange")
        I[TH_I + k * TD_I: TH_I + (k + 1) * TD_I] = [info[k]["parent"], info[k]["start"], info[k]["depth"], 1 if a is not None else 0]
    og_f, og_i = TH_F + nj * TD_F, TH_I + nj * TD_I
    pri_p = plane.get("priority", 0)
    for gi, g in enumerate(robot_geoms):
        if g["type"] not in GTYPE or g["condim"] != 3 or g["margin"] != 0 or g["gap"] != 0:
            raise NotImplementedError(f"tree kernel: geom {g['name']} ({g['type']})")
        gb = g["body"]
        owner = -1 if gb == base else body_of.get(gb)
        if owner is None:
            raise NotImplementedError("tree kernel: collision geoms must sit on the base or on a jointed body")
        pri_g = g.get("priority", 0)
        if pri_p > pri_g:
            mu, solref, solimp = plane["friction"][0], plane["solref"], plane["solimp"]
        elif pri_g > pri_p:
            mu, solref, solimp = g["friction"][0], g["solref"], g["solimp"]
        else:
            mu = max(plane["friction"][0], g["friction"][0])
            solref = [0.5 * (plane["solref"][i] + g["solref"][i]) for i in range(2)]
            solimp = [0.5 * (plane["solimp"][i] + g["solimp"][i]) for i in range(5)]
        cK, cB = solref_to_kb(solref, solimp, o["timestep"])
        size = (list(g["size"]) + [0, 0, 0])[:3]
        f = F[og_f + gi * TG_F: og_f + (gi + 1) * TG_F]
        f[0:3] = size
        f[3:6] = g["pos"]
        f[6:15] = quat_to_mat(g["quat"]).reshape(-1)
        f[15] = max(1e-5, mu)
        f[16:18] = [cK, cB]
        f[18:23] = clamp_solimp(solimp)
        f[23] = bodyw[gb][0] + bodyw[plane["body"]][0]   # diagApprox: translational inverse weights of both bodies
        f[24] = bounding_radius(g)                       # mj_collideGeoms' bounding-sphere filter of the robot-robot pairs
        f[25] = max(1e-5, g["friction"][0])              # the geom's own friction (robot-robot: the larger of the two)
        if pairs:  # a robot-robot pair mixes two equal sets: solref / solimp as stored require the plane's to equal the robot's own, or the plane not to win
            own_k, own_b = solref_to_kb(g["solref"], g["solimp"], o["timestep"])
            if not (np.isclose(own_k, cK) and np.isclose(own_b, cB) and np.allclose(clamp_solimp(g["solimp"]), clamp_solimp(solimp))):
                raise NotImplementedError("tree kernel: robot-robot pairs need the robot geoms' own solref / solimp to equal the plane-mixed ones")
        I[og_i + gi * TG_I: og_i + (gi + 1) * TG_I] = [owner, GTYPE[g["type"]]]
    # sensors: site positions (optionally in the frame of a world-fixed reference site) and site frame axes, as the Spot models declare them
    os_f, os_i = og_f + len(robot_geoms) * TG_F, og_i + len(robot_geoms) * TG_I
    nsd = 0

    def site_owner(si: int):
        st = desc["sites"][si]
        if list(st["quat"]) != [1.0, 0.0, 0.0, 0.0]:
            raise NotImplementedError("tree kernel: sites with their own rotation")
        b = st["body"]
        if b == base:
            return -1, np.array(st["pos"], dtype=np.float64), np.eye(3)
        if b in body_of:
            return body_of[b], np.array(st["pos"], dtype=np.float64), np.eye(3)
        if len(lay.body_joints[b]) == 0 and (bodies[b]["parent"] == 0 or b == 0):   # world-fixed: fold the body pose into the site
            Rw = quat_to_mat(bodies[b]["quat"]) if b != 0 else np.eye(3)
            pw = (np.array(bodies[b]["pos"]) if b != 0 else np.zeros(3)) + Rw @ np.array(st["pos"])
            return -2, pw, Rw
        raise NotImplementedError("tree kernel: sensor site on an unsupported body")

    for k, sn in enumerate(sensors):
        kind = {"framepos": 0, "framexaxis": 1, "frameyaxis": 2, "framezaxis": 3}.get(sn["type"])
        if kind is None or sn.get("objtype") != "site" or sn["dim"] != 3 or sn["adr"] != nsd:
            raise NotImplementedError(f"tree kernel: sensor {sn['name']} ({sn['type']})")
        owner, lp, Rw = site_owner(sn["obj"])
        f = F[os_f + k * TS_F: os_f + (k + 1) * TS_F]
        f[0:3] = lp
        has_ref = 0
        if sn.get("reftype") is not None:
            if kind != 0 or sn["reftype"] != "site":
                raise NotImplementedError("tree kernel: reference frames for position sensors on sites only")
            ro, rp, rR = site_owner(sn["ref"])
            if ro != -2:
                raise NotImplementedError("tree kernel: the reference site must be world-fixed")
            f[3:6], f[6:15], has_ref = rp, rR.reshape(-1), 1
        if owner == -2 and kind != 0:      # axis of a world-fixed site: a constant
            f[0:3] = Rw[:, kind - 1]
        I[os_i + k * TS_I: os_i + (k + 1) * TS_I] = [kind, owner, sn["adr"], has_ref]
        nsd += 3
    I[4:6] = [len(sensors), nsd]
    op = os_i + len(sensors) * TS_I
    I[6:8] = [len(pairs), op]
    for k, (a, b) in enumerate(pairs):
        I[op + k] = a | (b << 8)
    return F, I


def pack_tree_blob(desc: dict) -> bytes:
    """Header (magic, counts) + float image + int image, the form jh_tree_create takes."""
    F, I = pack_tree_model(desc)
    return np.array([0x34564A54, F.size, I.size, 0], dtype=np.uint32).tobytes() + F.tobytes() + I.tobytes()

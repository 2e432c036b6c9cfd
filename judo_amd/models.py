"""Model constants for the rollout kernels (host side).

`load_description(task)` reads `judo_amd/models/<task>.json` -- the numbers of the reference's MJCF
(judo/models/xml/{cartpole,cylinder_push,leap_cube,fr3_pick}.xml) transcribed
into this build's schema by tools/compile_mjcf.py.  `pack_model(desc)` derives the constants the kernels need
(what MuJoCo's compiler would precompute: inverse weights at the reference pose `qpos0`, solref -> (K, B) with
the refsafe clamp, candidate collision pairs) and serialises them into the blob `jh_model_create` uploads.

This is the product's own host code: numpy only, no oracle, no MuJoCo.
"""

from __future__ import annotations

import json
import os
import struct
from dataclasses import dataclass

import numpy as np

MODEL_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "models")
TASK_KIND = {"cartpole": 0, "cylinder_push": 1, "leap_cube": 2, "fr3_pick": 3}
BLOB_MAGIC = 0x314D484A
MINVAL = 1e-15
MINMU = 1e-5


def load_description(task: str) -> dict:
    path = os.path.join(MODEL_DIR, task + ".json")
    if not os.path.exists(path):
        raise ValueError(f"unknown task {task!r}: no model description at {path}")
    with open(path) as f:
        return json.load(f)


# ----------------------------------------------------------------------------------------- small rigid-body helpers
def quat_to_mat(q) -> np.ndarray:
    w, x, y, z = q
    return np.array(
        [
            [1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
            [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
            [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)],
        ]
    )


def quat_mul(a, b) -> np.ndarray:
    return np.array(
        [
            a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3],
            a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2],
            a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1],
            a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0],
        ]
    )


@dataclass
class Layout:
    """Index bookkeeping of a description (MuJoCo ordering: joints/dofs in body order)."""

    nq: int
    nv: int
    nu: int
    ns: int
    jnt_qposadr: list[int]
    jnt_dofadr: list[int]
    dof_body: list[int]
    dof_jnt: list[int]
    body_joints: list[list[int]]


def layout(desc: dict) -> Layout:
    nq = nv = 0
    qa, da, dof_body, dof_jnt = [], [], [], []
    body_joints: list[list[int]] = [[] for _ in desc["bodies"]]
    for j, jn in enumerate(desc["joints"]):
        qa.append(nq)
        da.append(nv)
        n_q, n_v = (7, 6) if jn["type"] == "free" else (1, 1)
        dof_body += [jn["body"]] * n_v
        dof_jnt += [j] * n_v
        body_joints[jn["body"]].append(j)
        nq += n_q
        nv += n_v
    return Layout(nq, nv, len(desc["actuators"]), desc["nsensordata"], qa, da, dof_body, dof_jnt, body_joints)


def qpos0(desc: dict) -> np.ndarray:
    lay = layout(desc)
    q = np.zeros(lay.nq)
    for j, jn in enumerate(desc["joints"]):
        if jn["type"] == "free":
            b = desc["bodies"][jn["body"]]
            q[lay.jnt_qposadr[j] : lay.jnt_qposadr[j] + 7] = list(b["pos"]) + list(b["quat"])
    return q


def _kinematics(desc: dict, q: np.ndarray):
    """Body poses, inertial frames and per-dof motion axes (angular, linear-at-origin) in world coordinates."""
    lay = layout(desc)
    nb = len(desc["bodies"])
    xpos = np.zeros((nb, 3))
    xquat = np.tile([1.0, 0, 0, 0], (nb, 1))
    S = np.zeros((lay.nv, 6))
    for b in range(1, nb):
        body = desc["bodies"][b]
        p = body["parent"]
        joints = lay.body_joints[b]
        if len(joints) == 1 and desc["joints"][joints[0]]["type"] == "free":
            j = joints[0]
            qa, da = lay.jnt_qposadr[j], lay.jnt_dofadr[j]
            pos, quat = q[qa : qa + 3].copy(), q[qa + 3 : qa + 7] / np.linalg.norm(q[qa + 3 : qa + 7])
            R = quat_to_mat(quat)
            for k in range(3):
                S[da + k, 3 + k] = 1.0
                S[da + 3 + k, :3] = R[:, k]
                S[da + 3 + k, 3:] = np.cross(pos, R[:, k])
        else:
            pos = xpos[p] + quat_to_mat(xquat[p]) @ np.array(body["pos"])
            quat = quat_mul(xquat[p], body["quat"])
            for j in joints:
                jn = desc["joints"][j]
                R = quat_to_mat(quat)
                axis = R @ np.array(jn["axis"])
                anchor = pos + R @ np.array(jn["pos"])
                val = q[lay.jnt_qposadr[j]]
                da = lay.jnt_dofadr[j]
                if jn["type"] == "slide":
                    pos = pos + axis * val
                    S[da, 3:] = axis
                else:
                    ql = np.concatenate([[np.cos(val / 2)], np.sin(val / 2) * np.array(jn["axis"])])
                    quat = quat_mul(quat, ql)
                    pos = anchor - quat_to_mat(quat) @ np.array(jn["pos"])
                    S[da, :3] = axis
                    S[da, 3:] = np.cross(anchor, axis)
        xpos[b], xquat[b] = pos, quat / np.linalg.norm(quat)
    return lay, xpos, xquat, S


def _spatial_inertia(mass: float, com: np.ndarray, Ic: np.ndarray) -> np.ndarray:
    cx = np.array([[0, -com[2], com[1]], [com[2], 0, -com[0]], [-com[1], com[0], 0]])
    I = np.zeros((6, 6))
    I[:3, :3] = Ic + mass * (com @ com * np.eye(3) - np.outer(com, com))
    I[:3, 3:] = mass * cx
    I[3:, :3] = mass * cx.T
    I[3:, 3:] = mass * np.eye(3)
    return I


def mass_matrix(desc: dict, q: np.ndarray) -> tuple[np.ndarray, dict]:
    """Joint-space inertia (CRB sum over bodies, + armature) at configuration q."""
    lay, xpos, xquat, S = _kinematics(desc, q)
    nb = len(desc["bodies"])
    # which dofs move each body: its own and its ancestors'
    body_dofs: list[list[int]] = [[] for _ in range(nb)]
    for b in range(1, nb):
        own = [d for d in range(lay.nv) if lay.dof_body[d] == b]
        body_dofs[b] = body_dofs[desc["bodies"][b]["parent"]] + own
    M = np.zeros((lay.nv, lay.nv))
    coms = np.zeros((nb, 3))
    for b in range(1, nb):
        body = desc["bodies"][b]
        R = quat_to_mat(xquat[b])
        com = xpos[b] + R @ np.array(body["ipos"])
        coms[b] = com
        if body["mass"] <= 0:
            continue
        Ri = R @ quat_to_mat(body["iquat"])
        I6 = _spatial_inertia(body["mass"], com, Ri @ np.diag(body["inertia"]) @ Ri.T)
        Sb = S[body_dofs[b]]
        M[np.ix_(body_dofs[b], body_dofs[b])] += Sb @ I6 @ Sb.T
    for d in range(lay.nv):
        M[d, d] += desc["joints"][lay.dof_jnt[d]]["armature"]
    return M, dict(layout=lay, xpos=xpos, xquat=xquat, S=S, body_dofs=body_dofs, coms=coms)


def inverse_weights(desc: dict) -> tuple[np.ndarray, np.ndarray]:
    """(dof_invweight0 (nv,), body_invweight0 (nbody,2)) at qpos0 -- MuJoCo's per-dof / per-body inverse inertia
    measures that scale every constraint regulariser R = (1-d)/d * diagApprox."""
    M, aux = mass_matrix(desc, qpos0(desc))
    lay, S = aux["layout"], aux["S"]
    Minv = np.linalg.inv(M)
    dofw = np.diag(Minv).copy()
    for j, jn in enumerate(desc["joints"]):
        if jn["type"] == "free":
            da = lay.jnt_dofadr[j]
            dofw[da : da + 3] = dofw[da : da + 3].mean()
            dofw[da + 3 : da + 6] = dofw[da + 3 : da + 6].mean()
    nb = len(desc["bodies"])
    bodyw = np.zeros((nb, 2))
    for b in range(1, nb):
        dofs = aux["body_dofs"][b]
        if not dofs:
            continue
        Sb = S[dofs]
        Jr = Sb[:, :3].T  # angular velocity Jacobian
        Jp = np.cross(Sb[:, :3], aux["coms"][b]).T + Sb[:, 3:].T  # velocity of the centre of mass
        Mi = Minv[np.ix_(dofs, dofs)]
        bodyw[b] = (np.trace(Jp @ Mi @ Jp.T) / 3, np.trace(Jr @ Mi @ Jr.T) / 3)
    return dofw, bodyw


def solref_to_kb(solref, solimp, timestep: float) -> tuple[float, float]:
    """(K, B) of the reference acceleration aref = -B*vel - K*imp*pos (MuJoCo solref semantics, refsafe clamp)."""
    dmax = min(0.9999, max(0.0001, solimp[1]))
    if solref[0] > 0:
        tc = max(solref[0], 2 * timestep)
        return 1 / max(MINVAL, dmax * dmax * tc * tc * solref[1] * solref[1]), 2 / max(MINVAL, dmax * tc)
    return -solref[0] / max(MINVAL, dmax * dmax), -solref[1] / max(MINVAL, dmax)


def clamp_solimp(solimp) -> list[float]:
    return [min(0.9999, max(0.0001, solimp[0])), min(0.9999, max(0.0001, solimp[1])), max(0.0, solimp[2]),
            min(0.9999, max(0.0001, solimp[3])), max(1.0, solimp[4])]


def actuator_ctrlrange(desc: dict) -> np.ndarray:
    """(nu, 2) control bounds, +-inf where the actuator is not ctrl-limited (judo/tasks/base.py:98-103)."""
    out = np.zeros((len(desc["actuators"]), 2))
    for i, a in enumerate(desc["actuators"]):
        out[i] = a["ctrlrange"] if a["ctrlrange"] is not None else (-np.inf, np.inf)
    return out


# ----------------------------------------------------------------------------------------- blob packing
def _pack(kind: int, lay: Layout, ntaskparam: int, floats, ints) -> bytes:
    f = np.asarray(floats, dtype=np.float32)
    i = np.asarray(ints, dtype=np.int32)
    head = struct.pack("<16I", BLOB_MAGIC, 1, kind, lay.nq, lay.nv, lay.nu, lay.ns, ntaskparam, f.size, i.size, 0, 0, 0, 0, 0, 0)
    return head + f.tobytes() + i.tobytes()


def _pack_cartpole(desc: dict) -> bytes:
    lay = layout(desc)
    o = desc["option"]
    cart, pole = desc["bodies"][1], desc["bodies"][2]
    jx, jth = desc["joints"]
    act = desc["actuators"][0]
    dofw, _ = inverse_weights(desc)
    K, B = solref_to_kb(jx["solreflimit"], jx["solimplimit"], o["timestep"])
    site_tip = next(s for s in desc["sites"] if s["name"] == "trace_pole")
    P = [
        o["timestep"], -o["gravity"][2], cart["mass"], pole["mass"], pole["ipos"][2], pole["inertia"][0], jx["damping"], jth["damping"],
        act["kp"], act["kv"], *(act["ctrlrange"] or (0, 0)), float(act["ctrlrange"] is not None),
        *(act["forcerange"] or (0, 0)), float(act["forcerange"] is not None),
        *(jx["range"] or (0, 0)), float(jx["range"] is not None), K, B, *clamp_solimp(jx["solimplimit"]), dofw[0], site_tip["pos"][2],
    ]
    return _pack(TASK_KIND["cartpole"], lay, 6, P, [])


def _pack_cylinder(desc: dict) -> bytes:
    lay = layout(desc)
    o = desc["option"]
    pusher, cart = desc["bodies"][2], desc["bodies"][3]
    jp, jc = desc["joints"][0], desc["joints"][2]
    act = desc["actuators"][0]
    gp = next(g for g in desc["geoms"] if g["name"] == "pusher")
    gc = next(g for g in desc["geoms"] if g["name"] == "cart")
    _, bodyw = inverse_weights(desc)
    solref = [0.5 * (a + b) for a, b in zip(gp["solref"], gc["solref"])]
    solimp = [0.5 * (a + b) for a, b in zip(gp["solimp"], gc["solimp"])]
    K, B = solref_to_kb(solref, solimp, o["timestep"])
    mu = max(MINMU, max(gp["friction"][0], gc["friction"][0]))
    margin = max(gp["margin"], gc["margin"]) - max(gp["gap"], gc["gap"])
    site = next(s for s in desc["sites"] if s["name"] == "pusher_site")
    P = [
        o["timestep"], pusher["mass"], cart["mass"], jp["damping"], jc["damping"], act["kp"], act["kv"],
        *(act["ctrlrange"] or (0, 0)), float(act["ctrlrange"] is not None), *(act["forcerange"] or (0, 0)), float(act["forcerange"] is not None),
        gp["size"][0] + gc["size"][0], K, B, *clamp_solimp(solimp), bodyw[2][0] + bodyw[3][0], mu, site["pos"][2], margin,
    ]
    return _pack(TASK_KIND["cylinder_push"], lay, 6, P, [])


def pack_model(desc: dict) -> bytes:
    task = desc.get("family", desc["task"])
    if task == "cartpole":
        return _pack_cartpole(desc)
    if task == "cylinder_push":
        return _pack_cylinder(desc)
    from judo_amd.engine_model import pack_engine_model  # articulated-body engine (leap_cube, fr3_pick)

    return pack_engine_model(desc)

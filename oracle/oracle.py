"""ctypes front-end of the ORACLE (`libjudo_oracle.so`, built by `oracle/Makefile`).

TEST INFRASTRUCTURE ONLY.  May be imported by `tests/`, `__graft_entry__.smoke()` and the
`cpu_baseline` leg of `bench.py` -- never by `judo_amd/` (the product path fails loudly without
its HIP library instead of falling back to this).

`load_model(task)` reads the build's model description (`judo_amd/models/<task>.json`, numbers
transcribed from the reference MJCF by `tools/compile_mjcf.py`) and instantiates the C engine's
`jo_model` through its builder API; the candidate collision pairs are chosen here with MuJoCo's
filter rules (same body, parent-child unless the parent is static, explicit excludes, both static).
"""

from __future__ import annotations

import ctypes as C
import json
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_MODELS = os.path.join(os.path.dirname(_HERE), "judo_amd", "models")
_LIB = None

JNT = {"free": 0, "slide": 2, "hinge": 3}
GEOM = {"plane": 0, "sphere": 2, "capsule": 3, "cylinder": 5, "box": 6}
SENS = {"framepos_site": 0, "framepos_body": 1, "jointpos": 2, "framezaxis_body": 3, "distance": 4, "framexaxis_site": 5, "frameyaxis_site": 6, "framezaxis_site": 7, "framequat_body": 8}

dp = C.POINTER(C.c_double)
ip = C.POINTER(C.c_int)


def _d(a) -> "C._Pointer":
    return np.ascontiguousarray(a, dtype=np.float64).ctypes.data_as(dp)


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (seconds)."""
    exp = os.environ.get("JUDO_ORACLE_EXPERIMENTS") == "1"  # tools/proto, tools/diag only: the library with the round-4 solver experiments compiled in (-DJO_EXPERIMENTS)
    so = os.path.join(_HERE, "libjudo_oracle_exp.so" if exp else "libjudo_oracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("jo_engine.c", "jo_plan.c", "jo_engine.h")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"] + (["exp"] if exp else []))
    return so


def lib() -> C.CDLL:
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        L = _LIB
        L.jo_model_new.restype = C.c_void_p
        L.jo_model_new.argtypes = [C.c_double, C.c_int, C.c_int, C.c_double, dp, C.c_int]
        L.jo_model_free.argtypes = [C.c_void_p]
        L.jo_add_body.argtypes = [C.c_void_p, C.c_int, dp, dp, C.c_double, dp, dp, dp]
        L.jo_add_joint.argtypes = [C.c_void_p, C.c_int, C.c_int, dp, dp, C.c_double, C.c_double, C.c_double, C.c_int, dp, C.c_double, C.c_int, dp, dp, dp, dp, dp]
        L.jo_add_geom.argtypes = [C.c_void_p, C.c_int, C.c_int, dp, dp, dp, dp, dp, dp, C.c_double, C.c_double, C.c_int]
        L.jo_add_pair.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.jo_set_geom_priority.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.jo_add_site.argtypes = [C.c_void_p, C.c_int, dp]
        L.jo_add_actuator.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_int, dp, C.c_int, dp]
        L.jo_add_sensor.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double]
        L.jo_add_equality_joint.argtypes = [C.c_void_p, C.c_int, C.c_int, dp, dp, dp]
        L.jo_model_finalize.argtypes = [C.c_void_p]
        L.jo_model_dims.argtypes = [C.c_void_p, ip]
        L.jo_model_get_invweight0.argtypes = [C.c_void_p, dp, dp]
        L.jo_model_get_qpos0.argtypes = [C.c_void_p, dp]
        L.jo_data_new.restype = C.c_void_p
        L.jo_data_free.argtypes = [C.c_void_p]
        L.jo_mass_matrix.argtypes = [C.c_void_p, C.c_void_p, dp]
        L.jo_energy.restype = C.c_double
        L.jo_energy.argtypes = [C.c_void_p, C.c_void_p, dp, dp]
        L.jo_forward_probe.argtypes = [C.c_void_p, dp, dp, dp, dp, dp, dp, dp, dp, ip, dp, dp]
        L.jo_body_pose.argtypes = [C.c_void_p, dp, C.c_int, dp, dp]
        L.jo_integrate_pos.argtypes = [C.c_void_p, dp, dp, dp]
        L.jo_pair_contact_counts.argtypes = [C.c_void_p, dp, C.c_int, C.POINTER(C.c_long)]
        L.jo_collide_shapes.argtypes = [C.c_int, dp, dp, dp, C.c_int, dp, dp, dp, C.c_double, dp]
        L.jo_collide_shapes.restype = C.c_int
        L.jo_export_problem.argtypes = [C.c_void_p, dp, dp, dp, dp, C.c_int, C.c_int, dp, dp, dp, dp, dp, dp, ip, ip, ip, ip, dp, dp, dp, ip]
        L.jo_export_problem.restype = C.c_int
        L.jo_rollout_batch.argtypes = [C.c_void_p, dp, C.c_int, dp, C.c_int, C.c_int, dp, dp, C.c_int]
        L.jo_spline_weights.argtypes = [C.c_int, C.c_int, dp, C.c_int, dp, dp]
        L.jo_spline_eval.argtypes = [dp, dp, C.c_int, C.c_int, C.c_int, C.c_int, dp]
        L.jo_mppi_sigma.argtypes = [C.c_double, C.c_int, C.c_double, C.c_int, C.c_int, dp]
        L.jo_cem_sigma_ramp.argtypes = [dp, C.c_int, C.c_double, C.c_double, C.c_double, C.c_int, C.c_int]
        L.jo_cem_pre_optimization.argtypes = [dp, C.c_int, dp, C.c_int, dp, C.c_int, dp]
        L.jo_sample_knots.argtypes = [dp, dp, dp, C.c_int, C.c_int, C.c_int, dp]
        L.jo_clip_knots.argtypes = [dp, C.c_int, C.c_int, C.c_int, dp, dp]
        L.jo_mppi_update.argtypes = [dp, dp, C.c_int, C.c_int, C.c_int, C.c_double, dp]
        L.jo_cem_update.argtypes = [dp, dp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, dp, dp, ip]
        L.jo_ps_update.argtypes = [dp, dp, C.c_int, C.c_int, C.c_int, dp]
        L.jo_reward_cartpole.argtypes = [dp, dp, C.c_int, C.c_int, dp, dp]
        L.jo_reward_cylinder.argtypes = [dp, C.c_int, C.c_int, dp, dp]
        L.jo_reward_leap.argtypes = [dp, C.c_int, C.c_int, C.c_int, dp, dp]
        L.jo_reward_fr3.argtypes = [dp, dp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, dp, ip, dp]
    return _LIB


def load_description(task: str) -> dict:
    with open(os.path.join(_MODELS, task + ".json")) as f:
        return json.load(f)


def collision_pairs(desc: dict, scope: str = "all") -> list[tuple[int, int]]:
    """Candidate geom pairs after MuJoCo's static filters (same welded body, parent-child unless the parent is the world, explicit excludes)
    that have a supported narrow phase.

    scope "all" (default): every such pair -- for leap_cube that includes the hand's self-collision (finger-finger, finger-palm), which
    jh_engine_v5.hip models, for fr3_pick the arm links against each other and against the gripper; scope "cube": leap_cube only, the cube's contacts alone
    (what jh_engine.hip / jh_engine_v2.hip model); scope "kernel": fr3_pick only, the subset k_fr3_v6 models (arm links against table and cube only)."""
    bodies, geoms = desc["bodies"], desc["geoms"]
    nb = len(bodies)
    njnt_body = [0] * nb
    for j in desc["joints"]:
        njnt_body[j["body"]] += 1

    def weld(b: int) -> int:  # static bodies are welded to the world
        while b > 0 and njnt_body[b] == 0:
            b = bodies[b]["parent"]
        return b

    def weld_parent(b: int) -> int:
        w = weld(b)
        return weld(bodies[w]["parent"]) if w > 0 else 0

    excl = {tuple(sorted(e)) for e in desc["excludes"]}
    supported = {("box", "box"), ("box", "sphere"), ("sphere", "box"), ("sphere", "sphere"), ("cylinder", "cylinder"), ("box", "capsule"), ("capsule", "box"),
                 ("plane", "sphere"), ("plane", "capsule"), ("plane", "box"), ("sphere", "plane"), ("capsule", "plane"), ("box", "plane")}
    spot = desc.get("family", desc["task"]) == "spot"
    if desc.get("family", desc["task"]) == "leap_cube":  # caltech_leap_cube's fingertip cylinders (caltech_leap_components/leap_rh.xml:131,175,219,259): MuJoCo's general convex collider
        supported = supported | {("cylinder", "box"), ("box", "cylinder"), ("cylinder", "sphere"), ("sphere", "cylinder")}
    if spot:  # the robot against itself (judo/models/xml/spot_primitive/contact.xml:4-14 lists the 11 body pairs it excludes): capsule-capsule, sphere-capsule on top
        supported = supported | {("capsule", "capsule"), ("sphere", "capsule"), ("capsule", "sphere")}
    if desc.get("family", desc["task"]) == "fr3_pick":  # link against link (fr3_components/fr3.xml:11,16,23,30,37,46,60,70: no exclude between non-adjacent links)
        supported = supported | {("capsule", "capsule")}
    pairs = []
    for g1 in range(len(geoms)):
        for g2 in range(g1 + 1, len(geoms)):
            b1, b2 = geoms[g1]["body"], geoms[g2]["body"]
            w1, w2 = weld(b1), weld(b2)
            if w1 == w2:
                continue  # same (welded) body, incl. both static
            if tuple(sorted((b1, b2))) in excl:
                continue
            # parent-child filter, not applied when the parent is the (welded) world
            if (weld_parent(b1) == w2 and w2 != 0) or (weld_parent(b2) == w1 and w1 != 0):
                continue
            if (geoms[g1]["type"], geoms[g2]["type"]) not in supported:
                continue
            if {geoms[g1]["type"], geoms[g2]["type"]} == {"capsule", "box"} and not spot and scope == "kernel":
                # fr3_pick, scope "kernel": the pair set k_fr3_v6 models (judo_amd/engine_model.py::generic_pairs) -- the arm links' capsules (stand-ins for collision meshes
                # that are absent from the reference repository) against static geometry (table) and the free body (cube) only.  The DEFAULT scope is what the MJCF says:
                # every pair MuJoCo's static filters leave (fr3_components/fr3.xml:11-99: eleven collision geoms with default contype, no <exclude>), link against
                # link and link against hand / finger boxes included; tests/test_oracle.py::test_fr3_link_pairs_never_touch_on_the_measured_workloads holds the two
                # sets to identical trajectories on the BASELINE workload and the shipped 64 x 250 configuration
                ob = b2 if geoms[g1]["type"] == "capsule" else b1
                free = any(j["body"] == ob and j["type"] == "free" for j in desc["joints"])
                if not (weld(ob) == 0 or free):
                    continue
            if (geoms[g1]["type"], geoms[g2]["type"]) == ("capsule", "capsule") and scope == "kernel":
                continue
            if desc.get("family", desc["task"]) == "leap_cube" and scope in ("cube", "task"):
                cube = next(i for i, g in enumerate(geoms) if g["name"] == "cube")
                if cube not in (g1, g2):
                    continue
            pairs.append((g1, g2))
    return pairs


class Model:
    """Owns a C `jo_model`."""

    def __init__(self, task: str, desc: dict | None = None, pairs: list[tuple[int, int]] | None = None, scope: str = "all") -> None:
        L = lib()
        self.task = task
        self.desc = d = desc if desc is not None else load_description(task)
        o = d["option"]
        integ = {"euler": 0, "implicitfast": 3}[o["integrator"]]
        cone = {"pyramidal": 0, "elliptic": 1}[o["cone"]]
        self.ptr = L.jo_model_new(o["timestep"], integ, cone, o["impratio"], _d(o["gravity"]), int(o["contact"]))
        for b in d["bodies"][1:]:
            r = L.jo_add_body(self.ptr, b["parent"], _d(b["pos"]), _d(b["quat"]), b["mass"], _d(b["ipos"]), _d(b["iquat"]), _d(b["inertia"]))
            assert r >= 0
        for j in d["joints"]:
            rng = j["range"] or [0.0, 0.0]
            fr = j["actuatorfrcrange"] or [0.0, 0.0]
            r = L.jo_add_joint(self.ptr, j["body"], JNT[j["type"]], _d(j["pos"]), _d(j["axis"]), j["damping"], j["armature"], j["frictionloss"],
                               int(j["range"] is not None), _d(rng), j["margin"], int(j["actuatorfrcrange"] is not None), _d(fr),
                               _d(j["solreflimit"]), _d(j["solimplimit"]), _d(j["solreffriction"]), _d(j["solimpfriction"]))
            assert r >= 0, (j["name"], r)
        for g in d["geoms"]:
            size = (list(g["size"]) + [0, 0, 0])[:3]
            r = L.jo_add_geom(self.ptr, g["body"], GEOM[g["type"]], _d(size), _d(g["pos"]), _d(g["quat"]), _d(g["friction"]), _d(g["solref"]), _d(g["solimp"]), g["margin"], g["gap"], g["condim"])
            assert r >= 0
            if g.get("priority", 0):
                assert L.jo_set_geom_priority(self.ptr, r, int(g["priority"])) == 0
        self.pairs = pairs if pairs is not None else collision_pairs(d, scope)
        for g1, g2 in self.pairs:
            assert L.jo_add_pair(self.ptr, g1, g2) >= 0
        for s in d["sites"]:
            assert L.jo_add_site(self.ptr, s["body"], _d(s["pos"])) >= 0
        for a in d["actuators"]:
            r = L.jo_add_actuator(self.ptr, a["joint"], a["kp"], a["kv"], int(a["ctrlrange"] is not None), _d(a["ctrlrange"] or [0, 0]),
                                  int(a["forcerange"] is not None), _d(a["forcerange"] or [0, 0]))
            assert r >= 0
        for s in d["sensors"]:
            if s["type"] == "framepos":
                ref = -1
                if s.get("reftype") is not None:  # position in the frame of a reference site
                    assert s["reftype"] == "site" and list(d["sites"][s["ref"]]["quat"]) == [1.0, 0.0, 0.0, 0.0]
                    ref = s["ref"]
                r = L.jo_add_sensor(self.ptr, SENS["framepos_site" if s["objtype"] == "site" else "framepos_body"], s["obj"], ref, 0.0)
            elif s["type"] in ("framexaxis", "frameyaxis", "framezaxis") and s.get("objtype") == "site":
                assert list(d["sites"][s["obj"]]["quat"]) == [1.0, 0.0, 0.0, 0.0] and s.get("reftype") is None
                r = L.jo_add_sensor(self.ptr, SENS[s["type"] + "_site"], s["obj"], -1, 0.0)
            elif s["type"] == "framequat":
                assert s["objtype"] == "body" and s.get("reftype") in (None, "body")
                r = L.jo_add_sensor(self.ptr, SENS["framequat_body"], s["obj"], s["ref"] if s.get("reftype") else -1, 0.0)
            elif s["type"] == "jointpos":
                r = L.jo_add_sensor(self.ptr, SENS["jointpos"], s["obj"], -1, 0.0)
            elif s["type"] == "framezaxis":
                r = L.jo_add_sensor(self.ptr, SENS["framezaxis_body"], s["obj"], -1, 0.0)
            else:
                r = L.jo_add_sensor(self.ptr, SENS["distance"], s["body1"], s["body2"], s["cutoff"])
            assert r >= 0
        for e in d["equalities"]:
            assert L.jo_add_equality_joint(self.ptr, e["joint1"], e["joint2"], _d(e["polycoef"]), _d(e["solref"]), _d(e["solimp"])) >= 0
        assert L.jo_model_finalize(self.ptr) == 0
        dims = (C.c_int * 7)()
        L.jo_model_dims(self.ptr, dims)
        self.nq, self.nv, self.nu, self.ns, self.nbody, self.ngeom, self.npair = list(dims)
        self.nx = self.nq + self.nv
        self.dt = o["timestep"]

    def __del__(self) -> None:
        try:
            lib().jo_model_free(self.ptr)
        except Exception:
            pass

    # ---- diagnostics
    def invweight0(self) -> tuple[np.ndarray, np.ndarray]:
        dw, bw = np.zeros(self.nv), np.zeros(2 * self.nbody)
        lib().jo_model_get_invweight0(self.ptr, _d(dw), _d(bw))
        return dw, bw.reshape(-1, 2)

    def qpos0(self) -> np.ndarray:
        q = np.zeros(self.nq)
        lib().jo_model_get_qpos0(self.ptr, _d(q))
        return q

    def mass_matrix(self, qpos: np.ndarray) -> np.ndarray:
        L = lib()
        d = L.jo_data_new()
        buf = np.zeros(8192 // 8)  # qpos is the first field of jo_data
        C.memmove(d, np.ascontiguousarray(qpos, dtype=np.float64).ctypes.data, 8 * self.nq)
        M = np.zeros((self.nv, self.nv))
        L.jo_mass_matrix(self.ptr, d, _d(M))
        L.jo_data_free(d)
        del buf
        return M

    def forward(self, qpos, qvel, ctrl) -> dict:
        L = lib()
        out = {k: np.zeros(self.nv) for k in ("qacc", "qacc_smooth", "qfrc_bias", "qfrc_constraint")}
        sens = np.zeros(max(self.ns, 1))
        info = (C.c_int * 4)()
        cons = np.zeros((160, 16))  # JO_MAXCON rows
        stats = np.zeros(3)
        ctrl = np.zeros(max(self.nu, 1)) if self.nu == 0 else np.ascontiguousarray(ctrl, dtype=np.float64)
        n = L.jo_forward_probe(self.ptr, _d(qpos), _d(qvel), _d(ctrl), _d(out["qacc"]), _d(out["qacc_smooth"]), _d(out["qfrc_bias"]),
                               _d(out["qfrc_constraint"]), _d(sens), info, _d(cons), _d(stats))
        out.update(sensordata=sens[: self.ns], ncon=info[0], nefc=info[1], solver_iter=info[2], con_overflow=info[3], contacts=cons[:n], solver_cost=stats[0], solver_gradnorm=stats[1], trace_M=stats[2])
        return out

    # ---- kinematics in isolation (tests/test_oracle_independent.py)
    def body_pose(self, qpos, body: int) -> tuple[np.ndarray, np.ndarray]:
        pos, mat = np.zeros(3), np.zeros(9)
        lib().jo_body_pose(C.c_void_p(self.ptr), _d(np.ascontiguousarray(qpos, dtype=np.float64)), int(body), _d(pos), _d(mat))
        return pos, mat.reshape(3, 3)

    def point_world(self, qpos, body: int, local) -> np.ndarray:
        pos, mat = self.body_pose(qpos, body)
        return pos + mat @ np.asarray(local, dtype=np.float64)

    def body_local(self, qpos, body: int, world_point) -> np.ndarray:
        pos, mat = self.body_pose(qpos, body)
        return mat.T @ (np.asarray(world_point, dtype=np.float64) - pos)

    def integrate_pos(self, qpos, dq) -> np.ndarray:
        out = np.zeros(self.nq)
        lib().jo_integrate_pos(C.c_void_p(self.ptr), _d(np.ascontiguousarray(qpos, dtype=np.float64)), _d(np.ascontiguousarray(dq, dtype=np.float64)), _d(out))
        return out

    def pair_contact_counts(self, qpos: np.ndarray) -> np.ndarray:
        """Contacts per candidate pair (in `self.pairs` order) summed over a batch of configurations (N, nq): kinematics + collision only."""
        q = np.ascontiguousarray(np.atleast_2d(qpos)[:, : self.nq], dtype=np.float64)
        out = np.zeros(len(self.pairs), dtype=np.int64)
        lib().jo_pair_contact_counts(C.c_void_p(self.ptr), _d(q), int(q.shape[0]), out.ctypes.data_as(C.POINTER(C.c_long)))
        return out

    def problem(self, qpos, qvel, ctrl, qacc_warmstart=None) -> dict:
        """The assembled constraint problem of one forward pass (`jo_export_problem`) and the oracle's own solution `qacc`."""
        L = lib()
        nv, me, mc = self.nv, 1024, 256
        M, a0, qacc = np.zeros((nv, nv)), np.zeros(nv), np.zeros(nv)
        J, aref, R, fl = np.zeros((me, nv)), np.zeros(me), np.zeros(me), np.zeros(me)
        tp, rid = np.zeros(me, dtype=np.int32), np.zeros(me, dtype=np.int32)
        cadr, cdim, cmu, cfr = np.zeros(mc, dtype=np.int32), np.zeros(mc, dtype=np.int32), np.zeros(mc), np.zeros((mc, 5))
        dims = (C.c_int * 3)()
        ipt = lambda a: a.ctypes.data_as(ip)
        ctrl = np.zeros(max(self.nu, 1)) if self.nu == 0 else np.ascontiguousarray(ctrl, dtype=np.float64)
        ws = None if qacc_warmstart is None else _d(np.ascontiguousarray(qacc_warmstart, dtype=np.float64))
        ne = L.jo_export_problem(C.c_void_p(self.ptr), _d(np.ascontiguousarray(qpos, dtype=np.float64)), _d(np.ascontiguousarray(qvel, dtype=np.float64)), _d(ctrl), ws, me, mc,
                                 _d(M), _d(a0), _d(J), _d(aref), _d(R), _d(fl), ipt(tp), ipt(rid), ipt(cadr), ipt(cdim), _d(cmu), _d(cfr), _d(qacc), C.cast(dims, ip))
        assert ne >= 0, "jo_export_problem: buffers too small"
        nc = dims[1]
        return dict(M=M, qacc_smooth=a0, J=J[:ne].copy(), aref=aref[:ne].copy(), R=R[:ne].copy(), frictionloss=fl[:ne].copy(), type=tp[:ne].copy(), id=rid[:ne].copy(),
                    con_adr=cadr[:nc].copy(), con_dim=cdim[:nc].copy(), con_mu=cmu[:nc].copy(), con_friction=cfr[:nc].copy(), qacc=qacc, cone=int(dims[2]), ncon=nc)

    def rollout(self, x0: np.ndarray, controls: np.ndarray, nthread: int | None = None) -> tuple[np.ndarray, np.ndarray]:
        """controls (N,H,nu), x0 (nx,) or (N,nx) -> states (N,H,nx), sensors (N,H,ns)  [RolloutBackend.rollout semantics]."""
        controls = np.ascontiguousarray(controls, dtype=np.float64)
        N, H, nu = controls.shape
        assert nu == self.nu
        x0 = np.ascontiguousarray(x0, dtype=np.float64)
        batched = int(x0.ndim == 2)
        states = np.zeros((N, H, self.nx))
        sensors = np.zeros((N, H, self.ns))
        lib().jo_rollout_batch(self.ptr, _d(x0), batched, _d(controls), N, H, _d(states), _d(sensors), nthread or os.cpu_count() or 1)
        return states, sensors


# ---------------------------------------------------------------- one narrow-phase routine in isolation
def supports_pair(kind_a: str, kind_b: str) -> bool:
    z3, q = np.zeros(3), np.array([1.0, 0, 0, 0])
    far = np.array([10.0, 0, 0])
    return lib().jo_collide_shapes(GEOM[kind_a], _d(np.ones(3)), _d(z3), _d(q), GEOM[kind_b], _d(np.ones(3)), _d(far), _d(q), 0.0, _d(np.zeros(56))) >= 0


def collide_pair(kind_a, size_a, pos_a, quat_a, kind_b, size_b, pos_b, quat_b, margin: float = 0.0):
    """Contacts of two free-standing shapes as [(dist, pos, normal a->b)]."""
    out = np.zeros(56)
    sa, sb = (list(np.atleast_1d(size_a)) + [0, 0, 0])[:3], (list(np.atleast_1d(size_b)) + [0, 0, 0])[:3]
    n = lib().jo_collide_shapes(GEOM[kind_a], _d(sa), _d(pos_a), _d(quat_a), GEOM[kind_b], _d(sb), _d(pos_b), _d(quat_b), float(margin), _d(out))
    assert n >= 0, f"no collision routine for {kind_a}-{kind_b}"
    return [(out[7 * i], out[7 * i + 1 : 7 * i + 4].copy(), out[7 * i + 4 : 7 * i + 7].copy()) for i in range(n)]


# ---------------------------------------------------------------- the optimizers' device noise stream, restated (jh_noise_normal)
def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Philox4x32-10 (Salmon et al., SC'11; Random123) on uint32 arrays: the counter-based generator behind `jh_noise_normal`."""
    c0, c1, c2, c3 = (np.asarray(c, dtype=np.uint64) for c in (c0, c1, c2, c3))
    k0, k1 = np.uint64(k0), np.uint64(k1)
    M0, M1, W0, W1, LO = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57), np.uint64(0x9E3779B9), np.uint64(0xBB67AE85), np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0, p1 = M0 * c0, M1 * c2
        c0, c1, c2, c3 = (p1 >> np.uint64(32)) ^ c1 ^ k0, p1 & LO, (p0 >> np.uint64(32)) ^ c3 ^ k1, p0 & LO
        k0, k1 = (k0 + W0) & LO, (k1 + W1) & LO
    return tuple(c.astype(np.uint32) for c in (c0, c1, c2, c3))


def noise_normal(seed: int, draw: int, rows: int, n_total: int) -> np.ndarray:
    """(rows, n_total) standard normals of draw number `draw`: block b of four rollouts of row r = Philox(counter (b, r, draw, 0), key = seed), two Box-Muller
    pairs on the top 24 bits of each word -- the definition `jh_noise_normal` implements (fp64 here: the kernel's fp32 transcendentals differ in the last bits)."""
    nb = (n_total + 3) // 4
    b, r = np.meshgrid(np.arange(nb, dtype=np.uint64), np.arange(rows, dtype=np.uint64))
    x = philox4x32_10(b, r, np.full_like(b, draw), np.zeros_like(b), seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    u = [((xi >> np.uint32(8)).astype(np.float64) + 0.5) * 2.0 ** -24 for xi in x]
    z = np.empty((rows, nb, 4))
    for h in range(2):
        rad, th = np.sqrt(-2.0 * np.log(u[2 * h])), 2.0 * np.pi * u[2 * h + 1]
        z[:, :, 2 * h], z[:, :, 2 * h + 1] = rad * np.cos(th), rad * np.sin(th)
    return z.reshape(rows, 4 * nb)[:, :n_total]


# ---------------------------------------------------------------- plan-path primitives (numpy in / numpy out)
def spline_weights(kind: str | int, knot_times, query_times) -> np.ndarray:
    k = {"zero": 0, "linear": 1, "cubic": 3}.get(kind, kind)
    t = np.ascontiguousarray(knot_times, dtype=np.float64)
    q = np.ascontiguousarray(query_times, dtype=np.float64)
    W = np.zeros((len(q), len(t)))
    assert lib().jo_spline_weights(int(k), len(t), _d(t), len(q), _d(q), _d(W)) == 0
    return W


def spline_eval(W, knots) -> np.ndarray:
    knots = np.ascontiguousarray(knots, dtype=np.float64)
    N, K, nu = knots.shape
    H = W.shape[0]
    U = np.zeros((N, H, nu))
    lib().jo_spline_eval(_d(W), _d(knots), N, H, K, nu, _d(U))
    return U


def mppi_sigma(sigma, use_ramp, noise_ramp, K, nu) -> np.ndarray:
    out = np.zeros((K, nu))
    lib().jo_mppi_sigma(float(sigma), int(use_ramp), float(noise_ramp), K, nu, _d(out))
    return out


def cem_sigma_ramp(sigma, use_ramp, noise_ramp, smin, smax) -> np.ndarray:
    s = np.array(sigma, dtype=np.float64, order="C")
    lib().jo_cem_sigma_ramp(_d(s), int(use_ramp), float(noise_ramp), float(smin), float(smax), s.shape[0], s.shape[1])
    return s


def cem_pre_optimization(sigma, old_times, new_times) -> np.ndarray:
    sigma = np.ascontiguousarray(sigma, dtype=np.float64)
    out = np.zeros((len(new_times), sigma.shape[1]))
    lib().jo_cem_pre_optimization(_d(sigma), sigma.shape[0], _d(old_times), len(new_times), _d(new_times), sigma.shape[1], _d(out))
    return out


def sample_knots(nominal, noise, sigma) -> np.ndarray:
    noise = np.ascontiguousarray(noise, dtype=np.float64)
    Nm1, K, nu = noise.shape
    out = np.zeros((Nm1 + 1, K, nu))
    lib().jo_sample_knots(_d(nominal), _d(noise), _d(np.broadcast_to(sigma, (K, nu))), Nm1 + 1, K, nu, _d(out))
    return out


def clip_knots(knots, lo, hi) -> np.ndarray:
    k = np.array(knots, dtype=np.float64, order="C")
    N, K, nu = k.shape
    lib().jo_clip_knots(_d(k), N, K, nu, _d(lo), _d(hi))
    return k


def mppi_update(knots, rewards, temperature) -> np.ndarray:
    knots = np.ascontiguousarray(knots, dtype=np.float64)
    N, K, nu = knots.shape
    out = np.zeros((K, nu))
    lib().jo_mppi_update(_d(knots), _d(rewards), N, K, nu, float(temperature), _d(out))
    return out


def cem_update(knots, rewards, num_elites, smin, smax):
    knots = np.ascontiguousarray(knots, dtype=np.float64)
    N, K, nu = knots.shape
    out, sig = np.zeros((K, nu)), np.zeros((K, nu))
    idx = (C.c_int * num_elites)()
    lib().jo_cem_update(_d(knots), _d(rewards), N, K, nu, num_elites, float(smin), float(smax), _d(out), _d(sig), idx)
    return out, sig, np.array(list(idx))


def ps_update(knots, rewards) -> np.ndarray:
    knots = np.ascontiguousarray(knots, dtype=np.float64)
    N, K, nu = knots.shape
    out = np.zeros((K, nu))
    lib().jo_ps_update(_d(knots), _d(rewards), N, K, nu, _d(out))
    return out


def reward_cartpole(states, controls, w=(10.0, 10.0, 0.1, 0.1, 0.01, 0.1)) -> np.ndarray:
    states = np.ascontiguousarray(states, dtype=np.float64)
    N, H, _ = states.shape
    out = np.zeros(N)
    lib().jo_reward_cartpole(_d(states), _d(controls), N, H, _d(w), _d(out))
    return out


def reward_cylinder(states, p=(0.5, 0.0, 0.1, 0.25, 0.0, 0.0)) -> np.ndarray:
    states = np.ascontiguousarray(states, dtype=np.float64)
    N, H, _ = states.shape
    out = np.zeros(N)
    lib().jo_reward_cylinder(_d(states), N, H, _d(p), _d(out))
    return out


def reward_leap(states, goal_quat=(1.0, 0.0, 0.0, 0.0), w_pos=100.0, w_rot=0.1, goal_pos=(0.0, 0.03, 0.1)) -> np.ndarray:
    states = np.ascontiguousarray(states, dtype=np.float64)
    N, H, nx = states.shape
    out = np.zeros(N)
    p = np.array([w_pos, w_rot, *goal_pos, *goal_quat], dtype=np.float64)
    lib().jo_reward_leap(_d(states), N, H, nx, _d(p), _d(out))
    return out


FR3_DEFAULT_P = (1.0, 10.0, 1.0, 10.0, 1.0, 1.0, 0.25, 0.1, 0.005, 2.0, 0.6, 0.4, 0.3)
FR3_ARM_HOME = (0, -0.7854, 0.0, -2.3562, 0.0, 1.5708, 0.7854, 0.04, 0.04)
FR3_SADR = (2, 3, 4, 11, 5)  # left_finger_table, right_finger_table, obj_table, grasp_site, ee_z


def reward_fr3(states, sensors, phase: int, p=FR3_DEFAULT_P, arm_home=FR3_ARM_HOME, sadr=FR3_SADR, nq=16, nv=15) -> np.ndarray:
    states = np.ascontiguousarray(states, dtype=np.float64)
    sensors = np.ascontiguousarray(sensors, dtype=np.float64)
    N, H, _ = states.shape
    out = np.zeros(N)
    pp = np.array([*p, *arm_home], dtype=np.float64)
    sa = (C.c_int * 5)(*sadr)
    lib().jo_reward_fr3(_d(states), _d(sensors), N, H, nq, nv, sensors.shape[-1], int(phase), _d(pp), sa, _d(out))
    return out


# ------------------------------------------------------------------------------------------------ action normalisers + the plan-step harness
class OracleNormalizer:
    """The three action normalisers (judo/utils/normalization.py:75-213) restated in numpy: "none" (:75-92), "min_max" over the finite
    ctrlranges (:94-137), "running" with the batch form of Welford's update and the asymmetric normalize / denormalize pair (:138-213)."""

    def __init__(self, kind: str, dim: int, lo=None, hi=None) -> None:
        self.kind, self.dim = kind, dim
        if kind == "min_max":
            self.lo, self.hi = np.asarray(lo, dtype=np.float64), np.asarray(hi, dtype=np.float64)
            self.dims = np.where((self.lo != -np.inf) & (self.hi != np.inf))[0]
        elif kind == "running":
            self.count, self.mean, self.std, self.M2 = 0, np.zeros(dim), np.ones(dim), np.zeros(dim)
            self.min_std, self.max_std, self.eps = 1e-5, 1e3, 1e-6
        elif kind != "none":
            raise ValueError(kind)

    def normalize(self, x):
        x = np.asarray(x, dtype=np.float64)
        if self.kind == "none":
            return x
        if self.kind == "min_max":
            out, d = x.copy(), self.dims
            out[..., d] = 2 * (x[..., d] - self.lo[d]) / (self.hi[d] - self.lo[d]) - 1
            return out
        return (x - self.mean) / (self.std + self.eps)

    def denormalize(self, x):
        x = np.asarray(x, dtype=np.float64)
        if self.kind == "none":
            return x
        if self.kind == "min_max":
            out, d = x.copy(), self.dims
            out[..., d] = (x[..., d] + 1) * (self.hi[d] - self.lo[d]) / 2 + self.lo[d]
            return out
        return x * self.std + self.mean

    def update(self, x) -> None:
        if self.kind != "running":
            return
        x = np.asarray(x, dtype=np.float64)
        axes = tuple(range(x.ndim - 1))
        self.count += int(np.prod(x.shape[:-1]))
        delta = x - self.mean
        self.mean = self.mean + delta.sum(axis=axes) / self.count
        self.M2 = np.maximum(self.M2 + (delta * (x - self.mean)).sum(axis=axes), 0)
        self.std = np.clip(np.sqrt(self.M2 / self.count), self.min_std, self.max_std)


def spline_resample(kind: str, old_times, old_knots, new_times) -> np.ndarray:
    """`nominal = prev_spline(new_times)` (judo/controller/controller.py:220-221): the previous plan's interpolant (hold-ends) at the shifted knot times."""
    return spline_eval(spline_weights(kind, old_times, new_times), np.asarray(old_knots, dtype=np.float64)[None])[0]


def trace_segments(sensors, rewards, adrs, max_num_traces: int) -> np.ndarray:
    """`Controller.update_traces` (judo/controller/controller.py:323-363): the min(max_num_traces, N) best rollouts, best first (equal rewards: the
    higher index first, what argsort(...)[-E:][::-1] gives for a stable sort), per elite every trace sensor's (H-1) segments [p_h, p_{h+1}]."""
    sensors, rewards = np.asarray(sensors, dtype=np.float64), np.asarray(rewards, dtype=np.float64)
    E = min(int(max_num_traces), len(rewards))
    elite = np.argsort(rewards, kind="stable")[len(rewards) - E :][::-1] if E > 0 else np.zeros(0, dtype=int)
    out = []
    for e in elite:
        for a in adrs:
            p = sensors[e, :, a : a + 3]
            out.append(np.stack([p[:-1], p[1:]], axis=1))
    return np.concatenate(out, axis=0) if out else np.zeros((0, 2, 3))

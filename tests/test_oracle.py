"""CPU tests of the ORACLE: (1) pinned against the golden vectors generated from the reference's numpy code,
(2) known-answer physics tests of the fp64 engine (no upstream test pins any rollout value)."""

import copy
import os

import numpy as np
import pytest

from oracle import oracle as O
from tests.conftest import GOLDEN


# ------------------------------------------------------------------------------------------------ golden: plan primitives
def test_spline_weights_match_reference():
    g = np.load(os.path.join(GOLDEN, "spline.npz"))
    keys = [k[: -len("_cfg")] for k in g.files if k.endswith("_cfg")]
    assert len(keys) >= 18
    for key in keys:
        kind, K, H, dt, hor, t0 = g[key + "_cfg"]
        kt, q = t0 + np.linspace(0, hor, int(K)), t0 + dt * np.arange(int(H))
        W = O.spline_weights(int(kind), kt, q)
        np.testing.assert_allclose(W, g[key + "_W"], atol=1e-12)
        np.testing.assert_allclose(W.sum(1), 1.0, atol=1e-12)
        np.testing.assert_allclose(O.spline_eval(W, g[key + "_knots"]), g[key + "_U"], atol=1e-12)
        np.testing.assert_allclose(O.spline_weights(int(kind), kt, g[key + "_shift_times"]) @ g[key + "_knots"][0], g[key + "_shift_knots"], atol=1e-12)
        far = O.spline_weights(int(kind), kt, np.array([t0 - 1.0, t0 + hor + 2.0])) @ g[key + "_knots"][0]
        np.testing.assert_allclose(far, g[key + "_far"], atol=1e-14)  # hold first / last knot outside the span


def test_sampling_matches_reference():
    g = np.load(os.path.join(GOLDEN, "optimizers.npz"))
    n = 0
    for key in sorted({k[: -len("_params")] for k in g.files if k.endswith("_params") and ("_mppi_" in k or "_ps_" in k)}):
        N, K, nu, ramp, nr, sig = g[key + "_params"]
        s = O.mppi_sigma(sig, ramp, nr, int(K), int(nu))
        out = O.sample_knots(g[key + "_nominal"], g[key + "_noise"], s)
        np.testing.assert_allclose(out, g[key + "_out"], atol=1e-15)
        np.testing.assert_array_equal(out[0], g[key + "_nominal"])
        n += 1
    assert n == 24
    for key in sorted({k[: -len("_params")] for k in g.files if k.endswith("_params") and "_cem_" in k}):
        N, K, nu, ramp, nr, smin, smax = g[key + "_params"]
        sig = g[key + "_sigma0"]
        for call in range(2):  # the ramp multiplies the sigma STATE on every call
            sig = O.cem_sigma_ramp(sig, ramp, nr, smin, smax)
            np.testing.assert_allclose(sig, g[f"{key}_call{call}_sigma_after"], atol=1e-15)
            np.testing.assert_allclose(O.sample_knots(g[f"{key}_call{call}_nominal"], g[f"{key}_call{call}_noise"], sig), g[f"{key}_call{call}_out"], atol=1e-15)
        so = O.cem_pre_optimization(g[key + "_prek_sigma_in"], g[key + "_prek_old_times"], g[key + "_prek_new_times"])
        np.testing.assert_allclose(so, g[key + "_prek_sigma_out"], atol=1e-14)


def test_updates_match_reference():
    g = np.load(os.path.join(GOLDEN, "optimizers.npz"))
    keys = sorted({k[: -len("_knots")] for k in g.files if k.startswith("update_") and k.endswith("_knots")})
    assert len(keys) == 16
    for key in keys:
        kn, rw = g[key + "_knots"], g[key + "_rewards"]
        for lam in (0.05, 0.0025):
            np.testing.assert_allclose(O.mppi_update(kn, rw, lam), g[f"{key}_mppi_{lam}"], atol=1e-13)
        np.testing.assert_allclose(O.ps_update(kn, rw), g[key + "_ps"], atol=0)
        for k in (2, 3):
            nom, sg, idx = O.cem_update(kn, rw, k, 0.01, 0.3)
            ref_idx = g[f"{key}_cem{k}_elite_idx"]
            assert np.allclose(np.sort(rw[idx]), np.sort(rw[ref_idx]))  # same elite rewards even when ties reorder indices
            if set(idx) == set(ref_idx):
                np.testing.assert_allclose(nom, g[f"{key}_cem{k}_nominal"], atol=1e-14)
                np.testing.assert_allclose(sg, g[f"{key}_cem{k}_sigma"], atol=1e-14)


def test_rewards_match_reference():
    g = np.load(os.path.join(GOLDEN, "rewards.npz"))
    np.testing.assert_allclose(O.reward_cartpole(g["cartpole_states"], g["cartpole_controls"]), g["cartpole_reward"], rtol=1e-13)
    for i in (0, 1):
        np.testing.assert_allclose(O.reward_cylinder(g[f"cylinder{i}_states"], (0.5, 0.0, 0.1, 0.25, *g[f"cylinder{i}_goal"])), g[f"cylinder{i}_reward"], rtol=1e-13)
    for i in (0, 1, 2):  # includes quat == goal, antipodal quat, |v| < 1e-6, angle > pi
        np.testing.assert_allclose(O.reward_leap(g[f"leap{i}_states"], g[f"leap{i}_goal_quat"]), g[f"leap{i}_reward"], rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(O.reward_leap(g["leap0_states"]), g["leap_default_reward"], rtol=1e-12)
    for ph, name in enumerate(["LIFT", "MOVE", "PLACE", "HOMING"]):
        np.testing.assert_allclose(O.reward_fr3(g[f"fr3_{name}_states"], g[f"fr3_{name}_sensors"], ph), g[f"fr3_{name}_reward"], rtol=1e-12)


# ------------------------------------------------------------------------------------------------ known-answer physics
def _variant(task, **edits):
    d = copy.deepcopy(O.load_description(task))
    for fn in edits.values():
        fn(d)
    return O.Model(task, d)


def test_cartpole_mass_matrix_and_bias_closed_form():
    m = O.Model("cartpole")
    d = m.desc
    mp, mc, l, I = d["bodies"][2]["mass"], d["bodies"][1]["mass"], d["bodies"][2]["ipos"][2], d["bodies"][2]["inertia"][0]
    # capsule inertia: cylinder + hemispherical caps (mass split by volume)
    r, L = 0.045, 1.0
    m_cyl = mp * (np.pi * r * r * L) / (np.pi * r * r * L + 4 / 3 * np.pi * r**3)
    m_sph = mp - m_cyl
    assert I == pytest.approx(m_cyl * (3 * r * r + L * L) / 12 + m_sph * (0.4 * r * r + L * L / 4 + 3 * r * L / 8), rel=1e-12)
    for th in (0.0, 0.7, 2.5, np.pi):
        M = m.mass_matrix(np.array([0.3, th]))
        np.testing.assert_allclose(M, [[mc + mp, mp * l * np.cos(th)], [mp * l * np.cos(th), I + mp * l * l]], atol=1e-14)
        out = m.forward(np.array([0.3, th]), np.array([0.2, -0.5]), np.array([0.3]))
        np.testing.assert_allclose(out["qfrc_bias"], [-mp * l * np.sin(th) * 0.25, -mp * 9.81 * l * np.sin(th)], atol=1e-13)


def test_cartpole_energy_conservation_and_pendulum_period():
    def free(d):
        d["joints"][0]["damping"] = 0.0
        d["joints"][0]["range"] = None
        d["actuators"] = []
        d["option"]["timestep"] = 0.0005

    m = _variant("cartpole", a=free)
    x0 = np.array([0.0, 2.0, 0.0, 0.0])
    st, _ = m.rollout(x0, np.zeros((1, 4000, 0)))

    def energy(x):
        mp, mc, l, I = 0.1, 1.0, 0.5, m.desc["bodies"][2]["inertia"][0]
        q, th, qd, thd = x
        M = np.array([[mc + mp, mp * l * np.cos(th)], [mp * l * np.cos(th), I + mp * l * l]])
        v = np.array([qd, thd])
        return 0.5 * v @ M @ v + mp * 9.81 * l * np.cos(th)

    assert abs(energy(st[0, -1]) - energy(x0)) < 2e-3 * abs(energy(x0))  # symplectic Euler at h = 0.5 ms
    # heavy cart -> fixed-pivot physical pendulum: small oscillations about the hanging pose
    def heavy(d):
        free(d)
        d["bodies"][1]["mass"] = 1e9

    mh = _variant("cartpole", a=heavy)
    st, _ = mh.rollout(np.array([0.0, np.pi + 0.01, 0.0, 0.0]), np.zeros((1, 8000, 0)))
    th = st[0, :, 1] - np.pi
    zc = np.where(np.diff(np.sign(th)) != 0)[0]
    period = 2 * np.mean(np.diff(zc)) * 0.0005
    I = mh.desc["bodies"][2]["inertia"][0]
    assert period == pytest.approx(2 * np.pi * np.sqrt((I + 0.1 * 0.25) / (0.1 * 9.81 * 0.5)), rel=2e-3)


def test_cartpole_servo_force_clamp_and_joint_limit():
    m = O.Model("cartpole")
    # kp*(u - x) = 100*1.8 is clamped to +10 N: first-step cart acceleration from rest with the pole hanging
    out = m.forward(np.array([0.0, np.pi]), np.zeros(2), np.array([5.0]))  # ctrl also clamped to 1.8
    M = m.mass_matrix(np.array([0.0, np.pi]))
    np.testing.assert_allclose(out["qacc_smooth"], np.linalg.solve(M, [10.0, 0.0]), rtol=1e-12)
    # cart pushed against its +1.8 limit: the soft limit holds it within a few mm
    st, _ = m.rollout(np.array([1.75, np.pi, 2.0, 0.0]), np.full((1, 100, 1), 1.8))
    assert st[0, :, 0].max() < 1.8 + 0.08 and st[0, -1, 0] < 1.81
    out = m.forward(np.array([1.85, np.pi]), np.array([0.5, 0.0]), np.array([1.8]))
    assert out["nefc"] == 1 and out["qfrc_constraint"][0] < 0  # limit force pushes back


def test_cylinder_push_free_motion_closed_form_and_contact_momentum():
    m = O.Model("cylinder_push")
    # no contact: implicit-damping Euler of a PD-driven unit mass, per axis
    x = np.array([1.0, 0.0, 2.0, 2.0, 0.3, -0.2, 0.1, 0.0])
    u = np.array([0.5, 0.4])
    st, _ = m.rollout(x, np.tile(u, (1, 30, 1)))
    h, p, v = 0.02, x[:2].copy(), x[4:6].copy()
    for t in range(30):
        a = (10 * (u - p) - 4 * v) / (1 + h * 4)
        v = v + h * a
        p = p + h * v
        np.testing.assert_allclose(st[0, t, :2], p, atol=1e-13)
        np.testing.assert_allclose(st[0, t, 4:6], v, atol=1e-13)
    # head-on contact with damping and the servo removed: momentum conserved, cylinders separate
    def bare(d):
        for j in d["joints"]:
            j["damping"] = 0.0
        d["actuators"] = []

    mb = _variant("cylinder_push", a=bare)
    x = np.array([0.0, 0.0, 0.6, 0.0, 1.0, 0.0, 0.0, 0.0])
    st, _ = mb.rollout(x, np.zeros((1, 60, 0)))
    mom = st[0, :, 4] + st[0, :, 6]
    np.testing.assert_allclose(mom, 1.0, atol=1e-9)
    assert st[0, -1, 6] > 0.3 and st[0, -1, 6] > st[0, -1, 4]  # the cart was pushed away
    assert (np.linalg.norm(st[0, :, 2:4] - st[0, :, 0:2], axis=1) > 0.5 - 0.025).all()  # penetration stays within one step of approach (1 m/s * 0.02 s)


def test_leap_free_fall_then_rest_on_palm():
    m = O.Model("leap_cube")
    q = np.array([0.0, 0.03, 0.1, 1, 0, 0, 0, 0.5, -0.75, 0.75, 0.25, 0.5, 0, 0.75, 0.25, 0.5, 0.75, 0.75, 0.25, 0.65, 0.9, 0.75, 0.6])
    x0 = np.concatenate([q, np.zeros(22)])
    U = np.tile(q[7:], (1, 150, 1))
    st, se = m.rollout(x0, U)
    h = 0.01
    for k in range(1, 6):  # ballistic phase (no contact yet): semi-implicit Euler
        assert st[0, k - 1, 2] == pytest.approx(0.1 - 9.81 * h * h * k * (k + 1) / 2, abs=1e-12)
        assert st[0, k - 1, 25] == pytest.approx(-9.81 * h * k, abs=1e-12)
    # after 1.5 s the cube rests on the palm: small velocity, constraint force balances its weight
    assert np.abs(st[0, -1, 23:26]).max() < 0.02 and np.abs(st[0, -1, 26:29]).max() < 0.5  # settled up to a slow rocking on the tilted palm
    out = m.forward(st[0, -1, :23], st[0, -1, 23:], U[0, -1])
    assert out["ncon"] >= 3
    assert out["qfrc_constraint"][2] == pytest.approx(0.108 * 9.81, rel=0.15)
    # sensors: first 16 = joint positions of the state the step started from, then site positions
    np.testing.assert_allclose(se[0, 1, :16], st[0, 0, 7:23], atol=1e-14)
    np.testing.assert_allclose(se[0, 1, 16:19], st[0, 0, 0:3], atol=1e-14)


def test_leap_joint_limits_and_friction_loss():
    def nograv(d):
        d["option"]["gravity"] = [0.0, 0.0, 0.0]

    m = _variant("leap_cube", a=nograv)
    q = np.array([0.0, 0.03, 0.3, 1, 0, 0, 0, 0.5, -0.75, 0.75, 0.25, 0.5, 0, 0.75, 0.25, 0.5, 0.75, 0.75, 0.25, 0.65, 0.9, 0.75, 0.6])
    x0 = np.concatenate([q, np.zeros(22)])
    # dof friction loss is a SOFT constraint: below saturation it acts as a damper of strength D*B, so a servo torque
    # of 0.0006 Nm (< frictionloss 0.001) makes the joint creep at v = tau / (D*B + damping + kv)
    U = np.tile(q[7:], (1, 50, 1))
    U[0, :, 0] += 0.002  # kp = 0.3 -> 0.0006 Nm
    st, _ = m.rollout(x0, U)
    dofw, _ = m.invweight0()
    R = (1 - 0.9) / 0.9 * dofw[6]
    B = 2 / (0.95 * 0.02)
    v_expected = 0.3 * (0.002 - (st[0, -1, 7] - q[7])) / (1 / R * B + 0.03 + 0.1)
    assert st[0, -1, 29] == pytest.approx(v_expected, rel=0.05)
    # far above saturation the friction force is exactly +-frictionloss
    # (at rest the row's reference acceleration is 0, a 0.3 Nm servo torque drives the joint forward: force = -frictionloss)
    big = q[7:].copy()
    big[0] += 1.0
    out = m.forward(q, np.zeros(22), big)
    assert out["qfrc_constraint"][6] == pytest.approx(-0.001, rel=1e-6)
    # command far outside the joint range (ctrl is clamped to the range, which equals the joint range): limit holds
    U2 = np.tile(q[7:], (1, 200, 1))
    U2[0, :, 3] = 5.0
    st, _ = m.rollout(x0, U2)
    hi = m.desc["joints"][4]["range"][1]
    assert st[0, :, 10].max() < hi + 0.02 and st[0, -1, 10] == pytest.approx(hi, abs=0.02)


def test_box_box_contacts_face_and_edge():
    desc = {
        "task": "boxes", "option": {"timestep": 0.01, "integrator": "implicitfast", "cone": "elliptic", "impratio": 1.0, "gravity": [0, 0, -9.81], "contact": True},
        "bodies": [dict(name="world", parent=-1, pos=[0, 0, 0], quat=[1, 0, 0, 0], mass=0, ipos=[0, 0, 0], iquat=[1, 0, 0, 0], inertia=[0, 0, 0]),
                   dict(name="a", parent=0, pos=[0, 0, 0.149], quat=[1, 0, 0, 0], mass=1.0, ipos=[0, 0, 0], iquat=[1, 0, 0, 0], inertia=[0.01, 0.01, 0.01]),
                   dict(name="floor", parent=0, pos=[0, 0, 0], quat=[1, 0, 0, 0], mass=0, ipos=[0, 0, 0], iquat=[1, 0, 0, 0], inertia=[0, 0, 0])],
        "joints": [dict(name="f", body=1, type="free", pos=[0, 0, 0], axis=[0, 0, 1], damping=0, armature=0, frictionloss=0, stiffness=0, ref=0, margin=0, range=None,
                        actuatorfrcrange=None, solreflimit=[0.02, 1], solimplimit=[0.9, 0.95, 0.001, 0.5, 2], solreffriction=[0.02, 1], solimpfriction=[0.9, 0.95, 0.001, 0.5, 2])],
        "geoms": [dict(name="ga", body=1, type="box", size=[0.05, 0.05, 0.05], pos=[0, 0, 0], quat=[1, 0, 0, 0], friction=[1, 0.005, 0.0001], solref=[0.02, 1],
                       solimp=[0.9, 0.95, 0.001, 0.5, 2], margin=0, gap=0, condim=3),
                  dict(name="gf", body=2, type="box", size=[1, 1, 0.1], pos=[0, 0, 0], quat=[1, 0, 0, 0], friction=[1, 0.005, 0.0001], solref=[0.02, 1],
                       solimp=[0.9, 0.95, 0.001, 0.5, 2], margin=0, gap=0, condim=3)],
        "sites": [], "actuators": [], "sensors": [], "excludes": [], "equalities": [], "nsensordata": 0,
    }
    m = O.Model("boxes", desc, pairs=[(0, 1)])
    out = m.forward(np.array([0, 0, 0.149, 1, 0, 0, 0.0]), np.zeros(6), np.zeros(0))
    assert out["ncon"] == 4  # face-face: the four corners of the small box
    np.testing.assert_allclose(out["contacts"][:, 0], -0.001, atol=1e-12)
    np.testing.assert_allclose(np.abs(out["contacts"][:, 4:7]), [[0, 0, 1]] * 4, atol=1e-12)
    assert out["contacts"][0, 6] < 0  # normal points from geom 1 (box a) to geom 2 (floor): downwards
    np.testing.assert_allclose(sorted(out["contacts"][:, 1]), [-0.05, -0.05, 0.05, 0.05], atol=1e-12)
    # resting: total normal force ~ weight
    st, _ = m.rollout(np.array([0, 0, 0.149, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0.0]), np.zeros((1, 100, 0)))
    o2 = m.forward(st[0, -1, :7], st[0, -1, 7:], np.zeros(0))
    assert o2["qfrc_constraint"][2] == pytest.approx(9.81, rel=0.02)
    # edge-edge: box rotated 45deg about x AND placed across the floor's top edge
    c = np.cos(np.pi / 8)
    s = np.sin(np.pi / 8)
    q = np.array([1.0 + 0.03, 0.0, 0.1 + 0.0707 - 0.05, c, s, 0, 0])  # hanging over the x = 1 edge of the floor
    o3 = m.forward(q, np.zeros(6), np.zeros(0))
    assert o3["ncon"] >= 1


def test_spot_policy_step_matches_golden():
    """Policy half of the Spot policy rollout (system_class.cpp:125-238): the vectorised oracle against the vectors written by the
    independent scalar restatement in tools/extract_spot_policy.py (which also verifies the ONNX graph: Gemm/Elu, alpha = beta = 1, transB = 1)."""
    from oracle import policy as P

    g = np.load(os.path.join(GOLDEN, "spot_policy.npz"))
    Ws, bs = P.load_actor()
    assert [w.shape for w in Ws] == [(512, 84), (256, 512), (128, 256), (12, 128)]
    np.testing.assert_allclose(P.actor(Ws, bs, g["obs"]), g["actions"], rtol=1e-12, atol=1e-12)
    nq, nv, bq, bv, lq, lv = (int(x) for x in g["step_layout"])
    obs, ctrl, out = P.policy_step(Ws, bs, g["step_qpos"], g["step_qvel"], g["step_command"], g["step_prev"], base_qpos=bq, base_qvel=bv, leg_qpos=lq, leg_qvel=lv)
    np.testing.assert_allclose(obs, g["step_obs"], rtol=0, atol=1e-14)
    np.testing.assert_allclose(out, g["step_out"], rtol=0, atol=0)
    np.testing.assert_allclose(ctrl, g["step_ctrl"], rtol=0, atol=1e-15)
    # hand-checkable pieces: identity orientation leaves the base velocity alone and gravity points down; the leg override is exclusive
    q = np.zeros((1, nq)); q[0, 3] = 1.0; v = np.zeros((1, nv)); v[0, :3] = [0.3, -0.2, 0.1]
    o = P.observation(q, v, np.zeros((1, 25)), np.zeros((1, 12)), base_qpos=bq, base_qvel=bv, leg_qpos=lq, leg_qvel=lv)
    np.testing.assert_allclose(o[0, :3], [0.3, -0.2, 0.1]); np.testing.assert_allclose(o[0, 6:9], [0, 0, -1])
    both = g["step_command"][3].copy()[None]; assert np.any(both[0, 10:13]) and np.any(both[0, 13:16])
    _, c, _ = P.policy_step(Ws, bs, g["step_qpos"][3:4], g["step_qvel"][3:4], both, g["step_prev"][3:4], base_qpos=bq, base_qvel=bv, leg_qpos=lq, leg_qvel=lv)
    np.testing.assert_allclose(c[0, 0:3], both[0, 10:13]); assert not np.allclose(c[0, 3:6], both[0, 13:16])  # FL wins, FR keeps the policy's targets


def test_spot_walks_under_the_extracted_policy():
    """End-to-end anchor for the policy step AND the floating-base engine: the Spot model (spot_primitive/robot.xml: free base, 12 leg +
    7 arm hinges, sphere / capsule / box geoms on a ground plane, position servos with force ranges) driven by the extracted ONNX actor through
    System::rollout's loop (one policy step, two physics substeps).  A locomotion policy trained elsewhere only tracks velocity commands if the
    observation layout, the joint permutations, the 0.2 action scale and the contact physics are all right: a wrong permutation makes it fall."""
    from oracle import policy as P

    om = P.spot_model()
    assert (om.nq, om.nv, om.nu) == (26, 25, 19)
    Ws, bs = P.load_actor()
    x0 = P.spot_reset_state()
    T = 150  # 3 s at the 50 Hz policy rate
    for vel, tol in (([0.0, 0.0, 0.0], 0.03), ([0.5, 0.0, 0.0], 0.1), ([0.0, 0.3, 0.0], 0.08), ([0.0, 0.0, 0.5], 0.12)):
        cmds = np.tile(P.DEFAULT_POLICY_COMMAND, (T, 1)); cmds[:, 0:3] = vel
        st, out = P.policy_rollout(om, Ws, bs, x0, cmds)
        assert np.isfinite(st).all()
        assert st[:, 2].min() > 0.42 and st[:, 2].max() < 0.6          # stays at standing height (commanded 0.52)
        up = 1 - 2 * (st[:, 4] ** 2 + st[:, 5] ** 2)                   # z component of the body z axis
        assert up.min() > 0.97                                         # stays upright
        yaw = 2 * np.arctan2(st[:, 6], st[:, 3])
        got = np.array([*(st[-1, :2] - st[49, :2]) / (100 * 0.02), (yaw[-1] - yaw[49]) / (100 * 0.02)])  # mean over the last 2 s, world frame
        if vel[2] == 0:  # heading stays ~0: world frame = body frame
            np.testing.assert_allclose(got, vel, atol=tol)
        else:
            assert abs(got[2] - vel[2]) < tol and np.abs(st[-1, :2]).max() < 0.3


def test_philox_known_answers_and_normal_moments():
    """The optimizers' device noise stream (jh_noise_normal) is Philox4x32-10 + Box-Muller: the oracle's restatement against the Random123 known-answer
    vectors (kat_vectors: zeros, all ones, digits of pi) and the moments of what comes out."""
    from scipy import stats

    assert [int(v[0]) for v in O.philox4x32_10([0], [0], [0], [0], 0, 0)] == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]
    assert [int(v[0]) for v in O.philox4x32_10([0xFFFFFFFF], [0xFFFFFFFF], [0xFFFFFFFF], [0xFFFFFFFF], 0xFFFFFFFF, 0xFFFFFFFF)] == [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]
    assert [int(v[0]) for v in O.philox4x32_10([0x243F6A88], [0x85A308D3], [0x13198A2E], [0x03707344], 0xA4093822, 0x299F31D0)] == [0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]
    z = O.noise_normal(1234, 3, 64, 16384)
    assert abs(z.mean()) < 4e-3 and abs(z.std() - 1) < 4e-3 and abs(stats.kurtosis(z.ravel())) < 2e-2 and abs(stats.skew(z.ravel())) < 1e-2
    assert stats.kstest(z.ravel()[:100000], "norm").pvalue > 1e-3
    # rows and draws are independent streams
    z2 = O.noise_normal(1234, 4, 64, 16384)
    assert abs(np.corrcoef(z.ravel(), z2.ravel())[0, 1]) < 5e-3 and abs(np.corrcoef(z[0], z[1])[0, 1]) < 3e-2


def test_caltech_fingertip_cylinders_are_collided_as_cylinders():
    """The oracle model of caltech_leap_cube holds the MJCF's fingertip cylinders (judo/models/xml/caltech_leap_components/leap_rh.xml:131,175,219,259) and collides them
    through the general convex routine, as MuJoCo does for cylinder-box; the leap kernel's sphere stand-in (judo_amd/engine_model.py::kernel_stand_ins) is NOT applied
    to it.  With the stand-in applied to a copy, some rollouts differ (the cylinder's rim reaches where the sphere does not) and the rest are bit-identical."""
    from judo_amd.engine_model import kernel_stand_ins

    desc = O.load_description("caltech_leap_cube")
    cyl = [i for i, g in enumerate(desc["geoms"]) if g["type"] == "cylinder"]
    assert len(cyl) == 4 and all(desc["geoms"][i]["size"][:2] == [0.014, 0.007] for i in cyl) and not any("substitute_for_cylinder" in g for g in desc["geoms"])
    om = O.Model("caltech_leap_cube")
    kinds = {tuple(sorted((desc["geoms"][a]["type"], desc["geoms"][b]["type"]))) for a, b in om.pairs}
    assert ("box", "cylinder") in kinds and ("cylinder", "sphere") in kinds
    cube = next(i for i, g in enumerate(desc["geoms"]) if g["name"] == "cube")
    assert all((min(cube, i), max(cube, i)) in set(om.pairs) for i in cyl)
    os_ = O.Model("caltech_leap_cube", desc=kernel_stand_ins(desc))
    assert len(os_.pairs) == len(om.pairs)
    from judo_amd.models import qpos0

    rng = np.random.default_rng(4)
    N, H = 64, 48
    home = np.array([a["ctrlrange"] for a in desc["actuators"]]).mean(axis=1) * 0 + 0.5
    U = home[None, None] + 0.4 * np.repeat(rng.standard_normal((N, 4, 16)), H // 4, axis=1)
    x0 = np.concatenate([qpos0(desc), np.zeros(22)])
    rc, _ = om.rollout(x0, U)
    rs, _ = os_.rollout(x0, U)
    differ = np.abs(rc - rs).reshape(N, -1).max(axis=1) > 0
    assert np.isfinite(rc).all() and 0 < differ.sum() < N


def test_fr3_link_pairs_never_touch_on_the_baseline_workload():
    """fr3_components/fr3.xml:11-99 gives the arm eleven collision geoms with default contype and no <exclude>: MuJoCo collides link against link and link against the
    gripper's boxes.  The oracle's default model has all 190 pairs; k_fr3_v6 models 78 of them (links against table and cube only: a stated deviation, the links'
    hulls being capsule stand-ins for meshes the reference repository does not hold).  On the BASELINE workload -- rollouts from QPOS_HOME with the CEM's first
    sigma, 0.155 ramped and clipped to 0.3 rad (judo/optimizers/cem.py:23-27,69-72), H = 40 -- and at TWICE that noise no left-out pair ever produces a contact: the two
    pair sets give bit-identical trajectories.  (At four times the noise the hand reaches links 0, 1 and 5: the deviation is real, it is outside what the planner samples.)"""
    from judo_amd.tasks import FR3Pick

    t = FR3Pick()
    desc = O.load_description("fr3_pick")
    full, sub = O.Model("fr3_pick"), O.Model("fr3_pick", scope="kernel")
    extra = [p for p in full.pairs if p not in set(sub.pairs)]
    assert len(full.pairs) == 190 and len(extra) == 112
    x0 = t.default_state()
    assert full.pair_contact_counts(x0[None, :16])[[full.pairs.index(p) for p in extra]].sum() == 0  # the stand-in hulls do not overlap at the home pose
    lo, hi = np.array([a["ctrlrange"] for a in desc["actuators"]]).T
    K, H, N = 4, 40, 96
    sigma_k = np.clip(0.155 * np.linspace(1, 4, K), 0.01, 0.3)
    W = O.spline_weights("linear", np.linspace(0, 1.0, K), 0.004 * np.arange(H))
    for mult in (1.0, 2.0):
        rng = np.random.default_rng(7)
        knots = np.clip(t.reset_command[None, None] + mult * sigma_k[None, :, None] * rng.standard_normal((N, K, 8)), lo, hi)
        U = O.spline_eval(W, knots)
        rf, _ = full.rollout(x0, U)
        rk, _ = sub.rollout(x0, U)
        assert np.array_equal(rf, rk), mult
        counts = full.pair_contact_counts(rf.reshape(-1, rf.shape[-1])[:, :16])
        assert counts[[full.pairs.index(p) for p in extra]].sum() == 0 and counts.sum() > 0

"""TEST INFRASTRUCTURE (oracle): vectorised numpy restatement of the policy half of the Spot policy rollout.

Reference: mujoco_extensions/system/system_class.cpp
  :103-123  joint orderings (Eigen permutation matrices) and default joint positions
  :125-206  System::setObservation   (84-d observation)
  :209-238  System::policyInference  (float32 cast, actor, 0.2 scale, leg re-ordering, arm pass-through, leg override)
and judo/models/policies/spot_locomotion.onnx (Gemm/Elu actor 84-512-256-128-12; weights extracted by tools/extract_spot_policy.py,
which also checks the graph).  Pinned by tests/golden/spot_policy.npz, produced by an independent scalar restatement in that tool;
onnxruntime is not installed, so the ONNX operator definitions are the ground truth for the actor.  Only tests/ may import this module.
"""

import os

import numpy as np

M2O = np.array([1, 6, 11, 2, 7, 12, 3, 8, 13, 4, 9, 14, 0, 5, 10, 15, 16, 17, 18])   # mujoco_to_orbit.indices()
O2M_LEGS = np.array([0, 3, 6, 9, 1, 4, 7, 10, 2, 5, 8, 11])                         # orbit_to_mujoco_legs.indices()
DEFAULT_JOINT_POS = np.array([0.12, 0.5, -1, -0.12, 0.5, -1, 0.12, 0.5, -1, -0.12, 0.5, -1, 0, -0.9, 1.8, 0, -0.9, 0, -1.54])
WEIGHTS = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "judo_amd", "models", "spot_locomotion_policy.npz")


def load_actor(path: str = WEIGHTS):
    w = np.load(path)
    return [w[f"W{i}"].astype(np.float64) for i in range(4)], [w[f"b{i}"].astype(np.float64) for i in range(4)]


def actor(Ws, bs, obs):
    x = np.asarray(obs, dtype=np.float64)
    for i, (W, b) in enumerate(zip(Ws, bs)):
        x = x @ W.T + b
        if i < 3:
            x = np.where(x > 0, x, np.expm1(np.minimum(x, 0)))
    return x


def _rot(v, q):  # mju_rotVecQuat, batched
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y), 2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
                  2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], axis=1).reshape(-1, 3, 3)
    return np.einsum("nij,nj->ni", R, v)


def observation(qpos, qvel, command, prev_out, base_qpos=0, base_qvel=0, leg_qpos=7, leg_qvel=6):
    qpos, qvel, command, prev_out = (np.asarray(a, dtype=np.float64) for a in (qpos, qvel, command, prev_out))
    inv = qpos[:, base_qpos + 3:base_qpos + 7] * np.array([1.0, -1.0, -1.0, -1.0])
    lin = _rot(qvel[:, base_qvel:base_qvel + 3], inv)
    grav = _rot(np.tile([0.0, 0.0, -1.0], (len(qpos), 1)), inv)
    jp, jv = np.zeros((len(qpos), 19)), np.zeros((len(qpos), 19))
    jp[:, M2O] = qpos[:, leg_qpos:leg_qpos + 19] - DEFAULT_JOINT_POS   # (P v)[idx[i]] = v[i]
    jv[:, M2O] = qvel[:, leg_qvel:leg_qvel + 19]
    return np.concatenate([lin, qvel[:, base_qvel + 3:base_qvel + 6], grav, command[:, 0:3], command[:, 3:10], command[:, 10:22], command[:, 22:25], jp, jv, prev_out], axis=1)


def policy_step(Ws, bs, qpos, qvel, command, prev_out, **layout):
    obs = observation(qpos, qvel, command, prev_out, **layout)
    out = actor(Ws, bs, obs.astype(np.float32)).astype(np.float32).astype(np.float64)
    ctrl = np.zeros((len(obs), 19))
    legs = np.zeros((len(obs), 12)); legs[:, O2M_LEGS] = 0.2 * out
    ctrl[:, :12] = DEFAULT_JOINT_POS[:12] + legs
    ctrl[:, 12:] = obs[:, 12:19]
    lj = obs[:, 19:31].reshape(-1, 4, 3)
    nz = (lj ** 2).sum(-1) > 0
    first = np.where(nz.any(1), nz.argmax(1), -1)
    for leg in range(4):
        m = first == leg
        ctrl[m, 3 * leg:3 * leg + 3] = lj[m, leg]
    return obs, ctrl, out


# ------------------------------------------------------------------------------------------------ the rollout around the policy step
LEGS_STANDING_POS_RL = np.array([0.12, 0.5, -1.0, -0.12, 0.5, -1.0, 0.12, 0.5, -1.0, -0.12, 0.5, -1.0])  # judo/tasks/spot/spot_constants.py:72-88
ARM_STOWED_POS = np.array([0, -3.11, 3.13, 1.56, 0, -1.56, 0.0])                                         # :90
STANDING_HEIGHT = 0.52                                                                                    # :95
DEFAULT_POLICY_COMMAND = np.concatenate([[0, 0, 0], ARM_STOWED_POS, np.zeros(12), [0, 0, STANDING_HEIGHT]])  # judo/tasks/spot/spot_base.py:159-161


def spot_model(self_collision: bool = False):
    """The Spot model (judo/models/xml/spot_primitive/robot.xml) in the oracle engine.  Default scope (what jh_engine_v4.hip models): robot geoms against the
    ground plane (sphere / capsule / box vs plane).  `self_collision=True` adds the robot's own pairs after MuJoCo's static filters and the 11 excludes of
    `spot_primitive/contact.xml:4-14` (capsule-capsule, sphere-capsule, box-capsule, box-box, box-sphere, sphere-sphere): what `k_tree_v4<true>`
    (jh_engine_v4.hip, the kernel's default since round 5) models and `tests/test_gpu_spot.py::test_robot_self_collision_matches_oracle` compares.  The model's sensors (relative frame positions, frame axes) are not evaluated."""
    from oracle import oracle as O

    desc = O.load_description("spot")
    if self_collision:
        return O.Model("spot", desc=desc, pairs=O.collision_pairs(desc, scope="all"))
    plane = next(i for i, g in enumerate(desc["geoms"]) if g["type"] == "plane")
    pairs = [(min(plane, i), max(plane, i)) for i, g in enumerate(desc["geoms"]) if i != plane]
    return O.Model("spot", desc=desc, pairs=pairs)


def spot_reset_state(arm=ARM_STOWED_POS):
    """SpotBase.reset_pose (spot_base.py:421-435) with zero velocities: (nq + nv,) = (26 + 25,)."""
    return np.concatenate([[0, 0, STANDING_HEIGHT, 1, 0, 0, 0], LEGS_STANDING_POS_RL, arm, np.zeros(25)])


def policy_rollout(om, Ws, bs, state, commands, physics_substeps=2, last_policy_output=None, with_sensors=False):
    """System::rollout (system_class.cpp:277-331) without the wall-clock cutoff: per command row one policy step, then
    `physics_substeps` engine steps with that control held; the state is recorded at the end of the substeps."""
    nq = om.nq
    x = np.asarray(state, dtype=np.float64).copy()
    out = np.zeros(12) if last_policy_output is None else np.asarray(last_policy_output, dtype=np.float64)
    states = np.zeros((len(commands), om.nx))
    sensors = np.zeros((len(commands), om.ns))
    for i, cmd in enumerate(np.asarray(commands, dtype=np.float64)):
        _, ctrl, o = policy_step(Ws, bs, x[None, :nq], x[None, nq:], cmd[None], out[None])
        out = o[0]
        st, se = om.rollout(x, np.repeat(ctrl, physics_substeps, axis=0)[None], nthread=1)
        x = st[0, -1]
        states[i] = x
        sensors[i] = se[0, -1]   # mjData.sensordata after the last mj_step of the row: from that step's forward pass (system_class.cpp:312-314)
    return (states, sensors, out) if with_sensors else (states, out)

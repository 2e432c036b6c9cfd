"""GPU parity tests for fr3_pick (BASELINE config 4: CEM, 7-DoF arm + gripper + cube + table) on the cooperative arm kernel
(default; the generic one-lane kernel is cross-checked at the end), against the fp64 oracle.  Collision scope on both sides: the box geoms (cube, table, hand and
finger boxes) among each other, and the capsule stand-ins for the arm links' collision meshes against table and cube."""

import os

import numpy as np
import pytest

from tests.conftest import bounded

pytestmark = pytest.mark.gpu

from tests.conftest import GOLDEN  # noqa: E402


def _controls(N, H, seed, sigma=0.1):
    from judo_amd.tasks import FR3Pick
    from oracle import oracle as O

    t = FR3Pick()
    rng = np.random.default_rng(seed)
    K = 4
    W = O.spline_weights("linear", np.linspace(0, 0.004 * H, K), 0.004 * np.arange(H))
    knots = t.reset_command[None, None] + sigma * np.linspace(0.25, 1.0, K)[None, :, None] * rng.standard_normal((N, K, 8))
    knots[0] = t.reset_command
    r = t.actuator_ctrlrange
    knots = np.clip(knots, r[:, 0], r[:, 1])
    return O.Model("fr3_pick"), t, knots, O.spline_eval(W, knots)


def test_fr3_reward_kernel_matches_reference_golden(gpu):
    from judo_amd.tasks import FR3Pick, Phase

    g = np.load(os.path.join(GOLDEN, "rewards.npz"))
    t = FR3Pick()
    for ph in Phase:
        t.phase = ph
        out = t.reward(g[f"fr3_{ph.name}_states"], g[f"fr3_{ph.name}_sensors"], None)
        np.testing.assert_allclose(out, g[f"fr3_{ph.name}_reward"], rtol=6e-7, atol=6e-7)


@pytest.mark.parametrize("x0_kind", ["home", "grasping"])
def test_fr3_rollout_backend_matches_oracle(gpu, x0_kind):
    from judo_amd.rollout_backend import GpuRolloutBackend

    N, H = 128, 40
    om, task, knots, U = _controls(N, H, seed=1)
    x0 = task.default_state()
    if x0_kind == "grasping":  # fingers around the cube, arm low: table / finger / cube contacts and the finger equality all active
        x0 = x0.copy()
        x0[7:14] = [0.0, 0.55, 0.0, -2.05, 0.0, 2.6, 0.785]
        x0[14:16] = [0.03, 0.025]
    rs, rsens = om.rollout(x0, U)
    be = GpuRolloutBackend("fr3_pick", N)
    gs, gsens, _ = be.rollout(x0, U)
    assert gs.shape == (N, H, 31) and gsens.shape == (N, H, 14) and np.isfinite(gs).all()
    # observed (tools/diag/fr3_parity_margins.py): first step 9e-7, all entries median 3e-9, joint / cube positions at the 40-step horizon 95th percentile 7e-7,
    # sensors median 2e-8, 99th percentile 2e-5 (the box-box distance sensors)
    np.testing.assert_allclose(gs[:, 0], rs[:, 0], atol=6e-6)  # one step: servo gains of 4500 on fp32 positions
    e = np.abs(gs - rs)
    assert bounded("np.median(e)", np.median(e), 2e-8) and bounded("np.percentile(e[:, -1, :16], 95)", np.percentile(e[:, -1, :16], 95), 5e-6)
    es = np.abs(gsens - rsens)
    assert bounded("np.median(es)", np.median(es), 1.5e-7) and bounded("np.percentile(es, 99)", np.percentile(es, 99), 7e-5)
    st = be.model.stats()
    # closing the empty gripper slams the two pad stacks together: for 1-2 steps the oracle sees up to 72 contacts (36 box pairs); the kernel holds 96
    # finger-finger + 32 other contacts per rollout since round 3, so nothing is dropped any more
    assert st["contact_overflow"] == 0


def test_fr3_closed_empty_gripper_keeps_every_pad_contact(gpu):
    """a gripper closed on nothing puts up to 92 pad-against-pad contacts between the two fingers (the reference's pad boxes,
    fr3_components/fr3.xml:84-116 + params_and_default.xml:43-55; 36 box pairs).  Single steps from jammed finger states (the oracle counts the contacts and
    the test insists that the case is really reached) and the closing motion itself (command 0 on the gripper: the stacks slam together around step 17-18):
    nothing may be dropped and the step must agree with the oracle as tightly as everywhere else."""
    from judo_amd.rollout_backend import GpuRolloutBackend
    from judo_amd.tasks import FR3Pick
    from oracle import oracle as O

    om, task = O.Model("fr3_pick"), FR3Pick()
    rng = np.random.default_rng(0)
    N = 192
    x0 = np.tile(task.default_state(), (N, 1))
    x0[:, 14:16] = rng.uniform(-0.003, 0.0005, (N, 2))         # both slides past the closed position: the pad stacks interpenetrate
    x0[:, 16 + 13 : 16 + 15] = rng.uniform(-0.3, 0.1, (N, 2))  # still closing / bouncing back
    U = np.tile(task.reset_command, (N, 1, 1))
    ncon = np.array([om.forward(x[:16], x[16:], task.reset_command)["ncon"] for x in x0])
    assert ncon.max() >= 88 and (ncon > 48).sum() >= N // 4, (ncon.max(), (ncon > 48).sum())  # round 2 kept 48 of them
    rs, _ = om.rollout(x0, U)
    be = GpuRolloutBackend("fr3_pick", N)
    gs, _, _ = be.rollout(x0, U)
    st = be.model.stats()
    assert st["contact_overflow"] == 0 and st["newton_cap_hits"] == 0
    # the finger velocities after the step are what up to 90 stiff rows decide: compare them relative to the step's own velocity change.
    # Reachable states first: total interpenetration below the pad thickness of 4 mm (a gripper closing at its fastest gains 1.2 mm per step).
    deep = x0[:, 14] + x0[:, 15] < -0.0039
    assert deep.sum() >= 10 and (~deep & (ncon > 48)).sum() >= 20
    dv = np.abs(rs[:, 0, 16 + 13 : 16 + 15] - x0[:, 16 + 13 : 16 + 15]).max(axis=1) + 1e-3
    ev = np.abs(gs[:, 0, 16 + 13 : 16 + 15] - rs[:, 0, 16 + 13 : 16 + 15]).max(axis=1) / dv
    # observed: median 4e-6 .. 2.4e-5 by contact count, max 9e-5 (the solve ends on the fp32 resolution of the finger accelerations)
    assert bounded("np.median(ev[~deep])", np.median(ev[~deep]), 5e-6) and bounded("ev[~deep].max()", ev[~deep].max(), 5e-4), (np.median(ev[~deep]), ev[~deep].max())
    np.testing.assert_allclose(gs[~deep, 0, :16], rs[~deep, 0, :16], atol=7e-7)
    # Beyond 4 mm the pad boxes have been pushed THROUGH one another: contacts with opposite normals fight each other, the constraint cost at the optimum is
    # ~6e7 (5e2 just below 4 mm) and the finger accelerations of ~50 m/s^2 are the difference of row forces ~1e6.  The problem itself is ill-conditioned there:
    # the fp64 oracle's own answer moves by 1-4 m/s^2 when its inputs are merely rounded to fp32 (measured below), so that is the yardstick, not 1e-5.
    # (An independent solver confirms the oracle's minimiser to 1e-8 on these states, tests/test_oracle_independent.py.)  Nothing is dropped there either.
    spread = 0.0
    for i in np.nonzero(deep)[0][:6]:
        s32, _ = om.rollout(x0[i].astype(np.float32).astype(np.float64), U[:1])
        spread = max(spread, np.abs(s32[0, 0, 16 + 13 : 16 + 15] - rs[i, 0, 16 + 13 : 16 + 15]).max())
    assert spread > 1e-3   # m/s after one 4 ms step: the oracle against itself
    assert np.abs(gs[deep, 0, 16 + 13 : 16 + 15] - rs[deep, 0, 16 + 13 : 16 + 15]).max() < max(4 * spread, 0.02)
    # the closing motion from the home pose, gripper commanded shut: the servo drives the fingers together at up to 0.95 m/s = 3.8 mm per step and finger, so the
    # fastest ones tunnel into the regime above in the step of the impact; parity is asked for up to the step in which a rollout gets there
    H, M = 40, 32
    u = task.reset_command.copy(); u[7] = 0.0
    U2 = np.tile(u, (M, H, 1))
    U2[:, :, 7] = rng.uniform(-0.02, 0.01, (M, 1))  # shut, each rollout a little differently
    rs2, _ = om.rollout(task.default_state(), U2)
    be2 = GpuRolloutBackend("fr3_pick", M)
    gs2, _, _ = be2.rollout(task.default_state(), U2)
    st2 = be2.model.stats()
    assert st2["contact_overflow"] == 0 and np.isfinite(gs2).all()
    pen = rs2[:, :, 14] + rs2[:, :, 15]
    assert (pen.min(axis=1) < -5e-4).sum() >= M // 2   # the stacks did slam together
    first_deep = np.where((pen < -0.0039).any(axis=1), (pen < -0.0039).argmax(axis=1), H)  # the state AFTER that step is deep: that step and all before it are well-posed
    ok = np.arange(H)[None, :] <= first_deep[:, None]
    assert ok.sum() > 0.5 * M * H and (first_deep < H).sum() >= 4
    e = np.abs(gs2 - rs2)
    assert bounded("np.median(e[ok])", np.median(e[ok]), 5e-9) and bounded("e[:, :, 14:16][ok].max()", e[:, :, 14:16][ok].max(), 5e-6) and bounded("np.percentile(e[:, :, 16 + 13 :][ok], 99)", np.percentile(e[:, :, 16 + 13 :][ok], 99), 7e-5), (np.median(e[ok]), e[:, :, 14:16][ok].max(), np.percentile(e[:, :, 16 + 13 :][ok], 99))


@pytest.mark.parametrize("phase,seed", [(0, 22), (1, 19), (2, 12), (3, 17)])
def test_fr3_plan_step_cem_matches_oracle(gpu, phase, seed):
    """One CEM plan step per phase against the oracle, cost by cost and then the refit.

    Where the cost error sits (tools/diag/fr3_cost_terms.py evaluates FR3Pick.reward's terms, judo/tasks/fr3_pick.py:225-311, on the kernel's and the oracle's states): NOT in
    the discrete finger-touch count (no flag differs in 4 x 10 240 rollout-steps; the fingers are 0.4 m above the table) but in the finger joint positions -- `gripper-open^2` and,
    in HOMING, the home-pose norm -- of the rollouts whose sampled gripper command shuts the fingers: the two pad stacks slam into each other at up to 0.95 m/s and end up
    millimetres inside one another, the regime in which the fp64 oracle's own answer moves by metres per second squared when its inputs are rounded to fp32
    (test_fr3_finger_pads_... above).  So the rollouts are split on the ORACLE's trajectory: `slam` = the stacks overlap by more than 0.5 mm at some step (55-60 % of a CEM
    sample at sigma 0.155-0.3), where the cost is held to 5 x the observed 95th percentile / maximum (9.4e-3 / 4.6e-2 over 24 seeds, profiles/r06_fr3_cost_terms.txt),
    and the rest, where it is held to 5 x the observed 1.0e-4 / 1.7e-4.  The seeds are those whose third and fourth best oracle rewards lie further apart than 20 x the
    elites' cost error (asserted): the CEM refit is then compared with the ORACLE's nominal unconditionally (tools/diag/fr3_elite_gap_seeds.py lists gaps per seed)."""
    import torch

    from judo_amd.controller import make_controller
    from judo_amd.tasks import Phase
    from oracle import oracle as O
    from tests.harness import oracle_plan_step

    N = 256
    rng = np.random.default_rng(seed)
    ctrl = make_controller("fr3_pick", "cem")
    ctrl.optimizer.config.num_rollouts = N
    ctrl.controller_cfg.horizon = 40 * ctrl.task.dt
    ctrl.reset()
    x0 = ctrl.task.default_state()
    if phase == 1:
        x0[2] = 0.05  # cube in the air -> MOVE
    elif phase == 2:
        x0[0:3] = [0.6, 0.4, 0.05]  # above the goal -> PLACE
    elif phase == 3:
        x0[0:3] = [0.6, 0.4, 0.02]  # on the table at the goal -> HOMING
    ctrl.current_state = x0
    noise = rng.standard_normal((N - 1, 4, 8)).astype(np.float32)
    ctrl.optimizer.injected_noise = noise
    ctrl.keep_candidates = True
    nominal0 = ctrl.nominal_knots.copy()
    sigma0 = ctrl.optimizer.sigma.copy()
    ctrl.update_action()
    torch.cuda.synchronize()
    assert ctrl.task.phase == phase == Phase(phase).value
    ref = oracle_plan_step(O.Model("fr3_pick"), ctrl, nominal0, noise, "cem", sigma0)  # the oracle's default model: every pair the MJCF leaves, link against link included
    cand = ctrl.candidate_knots_device.permute(2, 0, 1).cpu().numpy()
    np.testing.assert_allclose(cand, ref["knots"], rtol=4e-7, atol=4e-7)
    costs = -ctrl.rewards_local
    d = np.abs(costs + ref["rewards"])
    slam = (ref["states"][:, :, 14] + ref["states"][:, :, 15]).min(axis=1) < -5e-4
    assert 0.3 * N < slam.sum() < 0.8 * N
    assert bounded("cost error, finger stacks apart: median", np.median(d[~slam]), 2e-4) and bounded("cost error, finger stacks apart: p95", np.percentile(d[~slam], 95), 5e-4)
    assert bounded("cost error, finger stacks apart: max", d[~slam].max(), 1e-3)
    assert bounded("cost error, finger stacks slammed together: p95", np.percentile(d[slam], 95), 0.05) and bounded("cost error, finger stacks slammed together: max", d[slam].max(), 0.25)
    exp_nom, exp_sig, _ = O.cem_update(ref["knots"], -costs.astype(np.float64), 3, ctrl.optimizer.sigma_min, ctrl.optimizer.sigma_max)
    np.testing.assert_allclose(ctrl.nominal_knots, exp_nom, rtol=7e-7, atol=7e-8)
    np.testing.assert_allclose(ctrl.optimizer.sigma, exp_sig, rtol=5e-6, atol=5e-8)
    # the elite set IS the oracle's: the gap between the third and the fourth best reward clears the rollout error of the best four by a factor of 20
    order = np.argsort(-ref["rewards"])
    gap = ref["rewards"][order[2]] - ref["rewards"][order[3]]
    assert gap > 20 * d[order[:4]].max(), (gap, d[order[:4]].max())
    assert set(np.argsort(costs)[:3]) == set(order[:3])
    np.testing.assert_allclose(ctrl.nominal_knots, ref["nominal"], atol=7e-7)
    np.testing.assert_allclose(ctrl.optimizer.sigma, ref["sigma"], rtol=5e-6, atol=5e-8)


def test_fr3_full_size_sampled_rollouts_match_oracle(gpu):
    """BASELINE size (32 768 x 40, CEM, device noise): 256 of the plan step's own rollouts replayed through the fp64 oracle from the Philox columns the kernel read
    (VERDICT round 3, item 2), with the cost tolerances of the 256-rollout plan-step test; nominal and sigma against an exact elite refit on the GPU's own
    32 768 costs and candidates."""
    import torch

    from judo_amd.controller import make_controller
    from oracle import oracle as O
    from tests.conftest import record_margin
    from tests.harness import oracle_plan_step

    N, M = 32768, 256
    ctrl = make_controller("fr3_pick", "cem")
    ctrl.optimizer.config.num_rollouts = N
    ctrl.controller_cfg.horizon = 40 * ctrl.task.dt
    ctrl.reset()
    ctrl.current_state = ctrl.task.default_state()
    ctrl.optimizer.seed(12)
    ctrl.prefetch_noise = False
    ctrl.keep_candidates = True
    nominal0 = ctrl.nominal_knots.copy()
    sigma0 = ctrl.optimizer.sigma.copy()
    ctrl.update_action()
    torch.cuda.synchronize()
    costs = ctrl.costs_device.cpu().numpy().astype(np.float64)
    noise = ctrl.optimizer.last_noise
    cand = ctrl.candidate_knots_device.permute(2, 0, 1).cpu().numpy().astype(np.float64)
    assert costs.shape == (N,) and np.isfinite(costs).all() and cand.shape == (N, 4, 8)
    idx = np.concatenate([[0], np.sort(np.random.default_rng(6).choice(np.arange(1, N), M - 1, replace=False))])
    inj = noise[:, :, torch.as_tensor(idx[1:], device=noise.device)].permute(2, 0, 1).cpu().numpy()
    ref = oracle_plan_step(O.Model("fr3_pick"), ctrl, nominal0, inj, "cem", sigma0)
    np.testing.assert_allclose(cand[idx], ref["knots"], rtol=4e-7, atol=4e-7)
    d = np.abs(costs[idx] + ref["rewards"])
    record_margin("fr3_full_size_sampled", cost_median=np.median(d), cost_p95=np.percentile(d, 95), cost_max=d.max())
    slam = (ref["states"][:, :, 14] + ref["states"][:, :, 15]).min(axis=1) < -5e-4  # the finger stacks slammed into each other: see test_fr3_plan_step_cem_matches_oracle
    assert bounded("full size, finger stacks apart: median", np.median(d[~slam]), 2e-4) and bounded("full size, finger stacks apart: p95", np.percentile(d[~slam], 95), 5e-4)
    assert bounded("full size, finger stacks slammed together: p95", np.percentile(d[slam], 95), 0.05) and bounded("full size, finger stacks slammed together: max", d[slam].max(), 0.25)
    exp_nom, exp_sig, _ = O.cem_update(cand, -costs, 3, ctrl.optimizer.sigma_min, ctrl.optimizer.sigma_max)
    np.testing.assert_allclose(ctrl.nominal_knots, exp_nom, rtol=3e-7, atol=3e-8)
    np.testing.assert_allclose(ctrl.optimizer.sigma, exp_sig, rtol=5e-7, atol=5e-9)
    st = ctrl.solver_stats()
    assert st["contact_overflow"] < 1e-4 * st["steps"], st


def test_fr3_two_kernel_generations_agree(gpu):
    """The cooperative kernel (16 lanes per rollout, dense row-per-lane Hessian) and the one-lane-per-rollout generic kernel are
    independent implementations of the same step: on identical inputs their rollouts and sensors agree to solver tolerance over
    the first steps (contact-rich trajectories decorrelate later, each still tracking the oracle as the tests above require)."""
    from judo_amd.rollout_backend import GpuRolloutBackend

    N, H = 96, 40
    om, task, knots, U = _controls(N, H, seed=4)
    x0 = task.default_state().copy()
    x0[7:14] = [0.0, 0.55, 0.0, -2.05, 0.0, 2.6, 0.785]
    x0[14:16] = [0.03, 0.025]
    b2 = GpuRolloutBackend("fr3_pick", N)
    s2, y2, _ = b2.rollout(x0, U)
    b1 = GpuRolloutBackend("fr3_pick", N)
    b1.model.set_kernel(1)
    s1, y1, _ = b1.rollout(x0, U)
    np.testing.assert_allclose(s2[:, :3], s1[:, :3], atol=3e-6)
    np.testing.assert_allclose(y2[:, :3], y1[:, :3], atol=4e-7)
    e = np.abs(s2 - s1)
    assert bounded("np.median(e)", np.median(e), 2e-6) and bounded("np.percentile(e[:, -1, :3], 90)", np.percentile(e[:, -1, :3], 90), 5e-3)
    # generation 2 (jh_engine_v3.hip: contact Jacobian in LDS, row-per-lane assembly, one wave per SIMD) against the default generation 3
    # (jh_engine_v6.hip: matrix-free columns, Hessian by float atomics, two waves per SIMD): the same step up to summation order
    b3 = GpuRolloutBackend("fr3_pick", N)
    b3.model.set_kernel(2)
    s3, y3, _ = b3.rollout(x0, U)
    np.testing.assert_allclose(s3[:, :3], s2[:, :3], atol=1e-6)
    e = np.abs(s3 - s2)
    assert bounded("np.median(e)", np.median(e), 5e-7) and bounded("np.percentile(e[:, -1, :3], 90)", np.percentile(e[:, -1, :3], 90), 2e-3), (np.median(e), np.percentile(e[:, -1, :3], 90))


def test_fr3_arm_links_collide_with_table_and_cube(gpu):
    """VERDICT round 2, "missing" 2: the arm links of fr3_components/fr3.xml:11-81 collide (capsule stand-ins for the absent collision meshes) with the table
    and with the cube.  Arm poses folded down onto the table, the cube under the forearm: single steps and short rollouts of both GPU kernels against the
    oracle, which the test first asks to confirm that link contacts are really there."""
    from judo_amd.rollout_backend import GpuRolloutBackend
    from judo_amd.tasks import FR3Pick
    from oracle import oracle as O

    # scope "kernel": the pair subset k_fr3_v6 models.  These poses fold the arm onto the table on purpose -- far outside anything the planner samples -- and the
    # oracle's default model (every pair the MJCF leaves: link against link, link against gripper) would add contacts the kernel does not have; this test is about the
    # link-table / link-cube arithmetic.  (How far the default model is from these poses is recorded below; where the kernel actually goes, the left-out pairs never
    # touch: test_fr3_link_pairs_never_touch_where_the_kernel_goes.)
    om, task = O.Model("fr3_pick", scope="kernel"), FR3Pick()
    names = [g["name"] for g in om.desc["geoms"]]
    assert sum("link" in names[a] or "link" in names[b] for a, b in om.pairs) == 15  # table x links 1..7, cube x links 0..7
    rng = np.random.default_rng(0)
    N, H = 160, 12
    x0 = np.tile(task.default_state(), (N, 1))
    x0[:, 7:14] = np.array([0.0, 1.3, 0.0, -1.2, 0.0, 2.0, 0.785]) + 0.25 * rng.standard_normal((N, 7))
    x0[:, 0:2] = np.array([0.45, 0.0]) + 0.1 * rng.standard_normal((N, 2))
    x0[:, 16:] = 0.2 * rng.standard_normal((N, 15))
    U = np.repeat(x0[:, None, 7:15], H, axis=1)  # hold the pose
    fw = [om.forward(x[:16], x[16:], x[7:15]) for x in x0]
    link_con = np.array([sum("link" in names[int(r[13])] or "link" in names[int(r[14])] for r in f["contacts"]) for f in fw])  # (a contact lists the lower geom type first)
    ncon = np.array([f["ncon"] for f in fw])
    assert (link_con > 0).sum() > N // 2 and link_con.max() >= 4
    # (the hand box and the twelve finger boxes come down on the table with the links; generation 1 holds 32 contacts outside the gripper, generation 3 up to 96 --
    # `test_fr3_general_contacts_beyond_the_lds_pool` below is about those)
    ok32 = (ncon <= 32) & (link_con > 0)
    ok = (ncon <= 96) & (link_con > 0)
    assert ok32.sum() >= N // 4
    rs, _ = om.rollout(x0, U)
    be = GpuRolloutBackend("fr3_pick", N)
    gs, _, _ = be.rollout(x0, U)
    assert np.isfinite(gs).all()
    # one step: velocities relative to the step's own velocity scale (links dug 1-4 cm into the table are thrown out at several rad/s)
    sc = np.maximum(1.0, np.abs(rs[:, 0, 16:]).max(axis=1, keepdims=True))
    e1 = (np.abs(gs[:, 0, 16:] - rs[:, 0, 16:]) / sc).max(axis=1)
    assert bounded("np.median(e1[ok])", np.median(e1[ok]), 7e-6) and bounded("e1[ok].max()", e1[ok].max(), 0.001), (np.median(e1[ok]), e1[ok].max())
    np.testing.assert_allclose(gs[ok, 0, :16], rs[ok, 0, :16], atol=2e-5)
    assert be.model.stats()["contact_overflow"] <= 64 * H * int((ncon > 90).sum() + 1)  # nothing is dropped below the capacity
    few = ok & (ncon <= 20)
    eH = np.abs(gs[few, -1, :16] - rs[few, -1, :16]).max(axis=1)
    assert few.sum() >= 10 and bounded("np.median(eH)", np.median(eH), 7e-6) and bounded("np.percentile(eH, 75)", np.percentile(eH, 75), 1.5e-5), (few.sum(), np.median(eH), np.percentile(eH, 75))
    # the generic one-lane kernel: an independent second implementation of the same pair list
    b1 = GpuRolloutBackend("fr3_pick", N)
    b1.model.set_kernel(1)
    g1, _, _ = b1.rollout(x0, U[:, :2])
    e = (np.abs(g1[:, 0, 16:] - gs[:, 0, 16:]) / sc).max(axis=1)
    assert bounded("np.median(e[ok32])", np.median(e[ok32]), 5e-6) and bounded("e[ok32].max()", e[ok32].max(), 0.001), (np.median(e[ok32]), e[ok32].max())


def test_fr3_general_contacts_beyond_the_lds_pool(gpu):
    """The reference's SHIPPED fr3_pick configuration (64 rollouts, 1 s = 250 steps, judo/optimizers/overrides.py, judo/controller/overrides.py) presses the gripper onto
    the table all the time: ten pad boxes with 4-point manifolds are 40-60 contacts outside the gripper, where the kernel's LDS pool holds 32 (0.3-0.4 contacts dropped per
    rollout-step before round 3).  Generation 3 keeps up to 96: the rest lives in a row of global memory and the wave runs the six-slot copy of the solver.  States of that
    very workload with more than 32 such contacts, single steps against the oracle; and the workload itself no longer drops anything."""
    import torch

    from judo_amd.controller import make_controller
    from judo_amd.rollout_backend import GpuRolloutBackend
    from oracle import oracle as O

    ctrl = make_controller("fr3_pick", "cem")
    ctrl.solver_warnings = False
    ctrl.reset()
    ctrl.current_state = ctrl.task.default_state()
    ctrl.optimizer.seed(3)
    assert (ctrl.optimizer.num_rollouts, ctrl.num_timesteps) == (64, 250)
    ctrl.solver_stats()
    xs, t = [], 0.0
    for step in range(100):
        ctrl.force_materialize = step >= 30 and step % 5 == 0  # (the fused kernel otherwise: it is the one whose counters are asserted below)
        ctrl.time = t
        ctrl.update_action()
        t += 0.05
        if ctrl.force_materialize:
            torch.cuda.synchronize()
            st = ctrl.last_rollout[0].cpu().numpy().astype(np.float64)
            xs.append(st[::4, ::5].reshape(-1, st.shape[-1]))
    stats = ctrl.solver_stats()
    assert stats["contact_overflow"] < 1e-4 * stats["steps"], stats  # (0.33 per step in round 2's kernel)
    xs = np.concatenate(xs)
    om = O.Model("fr3_pick")
    gbody = [g["body"] for g in om.desc["geoms"]]
    fing = {i for i, b in enumerate(om.desc["bodies"]) if "finger" in b["name"]}
    gen, ff = np.zeros(len(xs), int), np.zeros(len(xs), int)
    for i, x in enumerate(xs):
        for r in om.forward(x[:16], x[16:], x[7:15])["contacts"]:
            if gbody[int(r[13])] in fing and gbody[int(r[14])] in fing:
                ff[i] += 1
            else:
                gen[i] += 1
    sel = (gen > 32) & (gen <= 96) & (ff <= 96)
    assert sel.sum() >= 10, (sel.sum(), np.bincount(np.minimum(gen // 8, 12)))
    x = xs[sel]
    U = x[:, None, 7:15]
    ref, _ = om.rollout(x, U)
    be = GpuRolloutBackend("fr3_pick", len(x))
    be.model.stats()
    g, _, _ = be.rollout(x, U)
    assert be.model.stats()["contact_overflow"] == 0
    sc = np.maximum(1.0, np.abs(ref[:, 0, 16:]).max(axis=1, keepdims=True))
    e = (np.abs(g[:, 0, 16:] - ref[:, 0, 16:]) / sc).max(axis=1)
    # (a gripper pressed flat onto the table with 60-90 contacts is a stiff, nearly rank-deficient solve: the worst of these states sits at 2e-3 of its velocity scale)
    assert bounded("np.median(e)", np.median(e), 3e-6) and bounded("e.max()", e.max(), 0.005), (np.median(e), e.max())
    np.testing.assert_allclose(g[:, 0, :16], ref[:, 0, :16], atol=5e-5)


def test_fr3_link_pairs_never_touch_where_the_kernel_goes(gpu):
    """The 112 geom pairs the MJCF collides and k_fr3_v6 leaves out (link against link, link against the gripper's boxes: fr3_components/fr3.xml:11-99, no <exclude>) --
    a stated deviation -- counted on the states the KERNEL's rollouts visit: (a) a 2 048-rollout sample of the BASELINE plan step (32 768 x H 40, CEM, device noise, from
    QPOS_HOME) and (b) the reference's shipped configuration (64 rollouts x 250 steps, judo/optimizers/overrides.py) in closed loop over 8 plan steps with the kernel as
    the plant.  Kinematics + the oracle's narrow phase on every visited configuration: no left-out pair produces a contact, so on these workloads the oracle with the
    kernel's subset IS the oracle with every pair (and the parity tests above compare against the latter)."""
    import torch

    from judo_amd.controller import make_controller
    from judo_amd.rollout_backend import GpuRolloutBackend
    from oracle import oracle as O

    full, sub = O.Model("fr3_pick"), O.Model("fr3_pick", scope="kernel")
    extra = np.array([i for i, p in enumerate(full.pairs) if p not in set(sub.pairs)])
    assert len(extra) == 112

    def visited_states(ctrl, sample):
        cand = ctrl.candidate_knots_device.permute(2, 0, 1).cpu().numpy().astype(np.float64)[sample]
        W = O.spline_weights(ctrl.spline_order, ctrl.spline_timesteps, ctrl.rollout_times)
        U = O.spline_eval(W, cand)
        be = GpuRolloutBackend("fr3_pick", len(sample))
        states, _, _ = be.rollout(ctrl.current_state, U)
        return np.asarray(states, dtype=np.float64), U

    # (a) BASELINE plan step
    ctrl = make_controller("fr3_pick", "cem")
    ctrl.optimizer.config.num_rollouts = 32768
    ctrl.controller_cfg.horizon = 40 * ctrl.task.dt
    ctrl.reset()
    ctrl.current_state = ctrl.task.default_state()
    ctrl.optimizer.seed(12)
    ctrl.keep_candidates = True
    ctrl.update_action()
    torch.cuda.synchronize()
    sample = np.sort(np.random.default_rng(3).choice(32768, 2048, replace=False))
    states, _ = visited_states(ctrl, sample)
    counts = full.pair_contact_counts(states.reshape(-1, states.shape[-1]))
    assert counts.sum() > 0 and counts[extra].sum() == 0, [full.pairs[i] for i in extra[counts[extra] > 0]]

    # (b) shipped configuration, closed loop, the kernel as the plant
    ctrl = make_controller("fr3_pick", "cem")
    assert ctrl.optimizer.config.num_rollouts == 64 and ctrl.num_timesteps == 250
    ctrl.reset()
    ctrl.current_state = ctrl.task.default_state()
    ctrl.optimizer.seed(5)
    ctrl.keep_candidates = True
    plant = GpuRolloutBackend("fr3_pick", 1)
    nsub = int(round(0.05 / ctrl.task.dt))
    total = 0
    for step in range(8):
        ctrl.time = 0.05 * step
        ctrl.update_action()
        torch.cuda.synchronize()
        states, _ = visited_states(ctrl, np.arange(64))
        counts = full.pair_contact_counts(states.reshape(-1, states.shape[-1]))
        total += int(counts.sum())
        assert counts[extra].sum() == 0, (step, [full.pairs[i] for i in extra[counts[extra] > 0]])
        u = np.stack([ctrl.action(ctrl.time + k * ctrl.task.dt) for k in range(nsub)])[None]
        xs, _, _ = plant.rollout(ctrl.current_state, u)
        ctrl.current_state = np.asarray(xs[0, -1], dtype=np.float64)
    assert total > 0

"""Edge sizes of the two cooperative engine kernels through the whole plan step: a single rollout, a ragged last wave (N not a
multiple of the 4 rollouts a wave holds), a one-step horizon, the maximum knot count the kernels keep in registers / LDS (K = 8), and the
argument errors the C ABI reports.  Oracle = the fp64 engine on the same candidates."""

import os

import numpy as np
import pytest

from tests.conftest import bounded

pytestmark = pytest.mark.gpu


def _plan(task, opt, N, H, K, seed):
    import torch
    from judo_amd.controller import make_controller
    from oracle import oracle as O
    from tests.harness import oracle_plan_step

    ctrl = make_controller(task, opt)
    ctrl.optimizer.config.num_rollouts = N
    ctrl.controller_cfg.horizon = H * ctrl.task.dt
    ctrl.reset()
    ctrl.current_state = ctrl.task.default_state()
    ctrl.system_metadata = {"goal_quat": np.array([0.0, 1.0, 0.0, 0.0])} if task == "leap_cube" else {}
    nu = ctrl.task.nu
    sigma0 = None
    if opt == "cem":
        sigma0 = np.full((K, nu), ctrl.optimizer.sigma[0, 0])  # constant in time: the node-count change re-interpolates it to itself
    # the node count changes between two plan steps, as a GUI slider would (the previous plan is resampled onto the new knot grid)
    ctrl.optimizer.config.num_nodes = K
    nominal0 = np.atleast_2d(ctrl.spline(ctrl.time + ctrl.spline_timesteps))
    noise = np.random.default_rng(seed).standard_normal((max(N - 1, 0), K, nu)).astype(np.float32)
    ctrl.optimizer.injected_noise = noise
    ctrl.keep_candidates = True
    ctrl.update_action()
    torch.cuda.synchronize()
    ref = oracle_plan_step(O.Model(task), ctrl, nominal0, noise, opt, sigma0)
    return ctrl, ref


@pytest.mark.parametrize("task,opt,N,H,K", [
    ("leap_cube", "mppi", 1, 16, 4), ("leap_cube", "ps", 5, 1, 4), ("leap_cube", "mppi", 7, 9, 8), ("leap_cube", "cem", 3, 12, 4),
    ("fr3_pick", "cem", 3, 10, 4), ("fr3_pick", "mppi", 1, 1, 4), ("fr3_pick", "ps", 6, 7, 8), ("fr3_pick", "cem", 9, 25, 5),
])
def test_small_and_ragged_plan_steps_match_oracle(gpu, task, opt, N, H, K):
    ctrl, ref = _plan(task, opt, N, H, K, seed=N * 100 + H)
    assert ctrl.num_timesteps == H and ctrl.optimizer.num_nodes == K
    cand = ctrl.candidate_knots_device.permute(2, 0, 1).cpu().numpy()
    np.testing.assert_allclose(cand, ref["knots"], rtol=4e-7, atol=4e-7)
    costs = -ctrl.rewards_local
    assert costs.shape == (N,) and np.isfinite(costs).all()
    # few steps from the home pose: fp32 vs fp64 engine, costs are sums / means of O(1) terms
    np.testing.assert_allclose(costs, -ref["rewards"], rtol=2e-6, atol=2e-5)
    if N == 1:  # the only sample is the unperturbed nominal: every optimiser returns it
        np.testing.assert_allclose(ctrl.nominal_knots, ref["knots"][0], rtol=0, atol=3e-7)


def test_engine_kernels_reject_more_knots_than_they_hold(gpu):
    """The C entry point refuses K > 8 on the cooperative fr3 kernel (a lane's knots are staged on chip, 8 of them); the controller routes such a plan step
    through the materialise path instead (tests/test_gpu_controller.py), so a live num_nodes edit never raises in the control loop.  The leap kernel reads its
    knots from memory: 9 knots are a fused plan step there."""
    import torch

    from judo_amd import _lib
    from judo_amd.controller import make_controller

    for task, fused in (("leap_cube", True), ("fr3_pick", False)):
        ctrl = make_controller(task, "mppi")
        ctrl.optimizer.config.num_rollouts = 8
        ctrl.optimizer.config.num_nodes = 9
        ctrl.controller_cfg.horizon = 8 * ctrl.task.dt
        ctrl.reset()
        ctrl.current_state = ctrl.task.default_state()
        ctrl.system_metadata = {"goal_quat": np.array([0.0, 1.0, 0.0, 0.0])} if task == "leap_cube" else {}
        assert ctrl.uses_fused_cost == fused
        ctrl.update_action()
        assert np.isfinite(ctrl.nominal_knots).all() and ctrl.nominal_knots.shape == (9, ctrl.nu)
        K, nu, N, H = 9, ctrl.nu, 8, 8
        z = lambda *shape: torch.zeros(shape, dtype=torch.float32, device="cuda")  # noqa: E731
        st = _lib.lib().jh_rollout_cost(ctrl.model.handle, _lib.ptr(z(ctrl.task.nq + ctrl.task.nv)), _lib.ptr(z(K * nu)), _lib.ptr(z(K * nu * N)), N, _lib.ptr(z(K * nu)),
                                        _lib.ptr(z(H * K)), _lib.ptr(z(2 * nu)), _lib.ptr(z(32)), 0, N, 0, H, K, _lib.ptr(z(N)), None, 0)
        if fused:
            _lib.check(st, "jh_rollout_cost")
        else:
            with pytest.raises(ValueError, match="at most 8 knots"):
                _lib.check(st, "jh_rollout_cost")


@pytest.mark.parametrize("task,N,H", [("leap_cube", 130, 48), ("fr3_pick", 130, 40), ("cylinder_push", 200, 64)])
def test_rollouts_are_bit_reproducible_and_independent_of_their_wave_position(gpu, task, N, H):
    """Two identical launches agree bit for bit, and a rollout's trajectory does not depend on which lane row / which wave-mates it
    gets (the same controls shifted by 1..3 positions): what makes the sharded plan step reproduce the single-GPU one exactly."""
    import torch
    from judo_amd.rollout_backend import GpuRolloutBackend
    from judo_amd.tasks import get_registered_tasks

    t = get_registered_tasks()[task][0]()
    x0 = torch.as_tensor(np.asarray(t.default_state(), dtype=np.float32)).cuda()
    g = torch.Generator(device="cuda").manual_seed(5)
    U = (0.3 * torch.randn((N, H, t.nu), device="cuda", generator=g) + torch.as_tensor(np.asarray(t.optimizer_warm_start(), dtype=np.float32)).cuda()).contiguous()
    be = GpuRolloutBackend(task, N)
    s0, y0 = be.rollout_device(x0, U)
    s1, y1 = be.rollout_device(x0, U)
    assert torch.equal(s0, s1) and torch.equal(y0, y1)
    for sh in (1, 2, 3):
        Us = torch.cat([U[:1].expand(sh, -1, -1), U[:-sh]]).contiguous()
        s2, y2 = be.rollout_device(x0, Us)
        assert torch.equal(s2[sh:], s0[:-sh]) and torch.equal(y2[sh:], y0[:-sh])


@pytest.mark.parametrize("task,N,H", [("leap_cube", 37, 48), ("fr3_pick", 37, 40)])
def test_latency_mode_of_small_launches_changes_no_bit(gpu, task, N, H, monkeypatch):
    """A launch that would leave SIMDs idle lets four (or two) rows of a wave compute the same rollout so that the wave does not wait for the slowest of four different
    Newton solves (include/judo_amd.h, jh_latency_shift): same arithmetic, the first row writes -- the results are the bits of the one-rollout-per-row mapping."""
    import torch
    from judo_amd.rollout_backend import GpuRolloutBackend
    from judo_amd.tasks import get_registered_tasks

    t = get_registered_tasks()[task][0]()
    x0 = torch.as_tensor(np.asarray(t.default_state(), dtype=np.float32)).cuda()
    g = torch.Generator(device="cuda").manual_seed(9)
    U = (0.3 * torch.randn((N, H, t.nu), device="cuda", generator=g) + torch.as_tensor(np.asarray(t.optimizer_warm_start(), dtype=np.float32)).cuda()).contiguous()
    be = GpuRolloutBackend(task, N)
    out = {}
    for mode in ("0", "1", "2", None):  # one rollout per row; two copies; four copies; the launcher's own choice (four at this size)
        if mode is None:
            monkeypatch.delenv("JUDO_AMD_LATENCY_SHIFT", raising=False)
        else:
            monkeypatch.setenv("JUDO_AMD_LATENCY_SHIFT", mode)
        be.model.stats()
        s, y = be.rollout_device(x0, U)
        out[mode] = (s.clone(), y.clone(), be.model.stats())
    for mode in ("1", "2", None):
        assert torch.equal(out[mode][0], out["0"][0]) and torch.equal(out[mode][1], out["0"][1])
        # the copies are not counted: the solver statistics describe N rollouts, not N x copies
        assert out[mode][2]["steps"] == out["0"][2]["steps"] == N * H and out[mode][2]["newton_iters"] == out["0"][2]["newton_iters"]


def test_device_noise_is_a_function_of_the_global_rollout_index(gpu):
    """jh_noise_normal (the optimizers' noise: Philox4x32-10 + Box-Muller, include/judo_amd.h): against the oracle's restatement, and shard by shard -- any
    column range, aligned to the generator's blocks of four or not, into a buffer with any row stride, is bit-identical to those columns of the full draw.
    That is what makes a sharded plan step independent of the number of GPUs without every rank drawing all rollouts' noise."""
    import torch

    from judo_amd import _lib
    from judo_amd.device import current_stream_ptr
    from oracle import oracle as O

    L = _lib.lib()
    rows, n_total, seed, draw = 64, 4099, 0x1234_5678_9ABC, 7
    full = torch.empty((rows, n_total), dtype=torch.float32, device="cuda")
    _lib.check(L.jh_noise_normal(seed, draw, rows, 0, n_total, full.data_ptr(), n_total, current_stream_ptr()), "jh_noise_normal")
    ref = O.noise_normal(seed, draw, rows, n_total)
    got = full.cpu().numpy()
    # fp32 log / sincos against fp64: relative to the radius, worst in the tails
    assert bounded("np.abs(got - ref).max()", np.abs(got - ref).max(), 3e-5) and bounded("np.abs(got - ref).mean()", np.abs(got - ref).mean(), 5e-7)
    for off, cnt, ld in ((0, 1, 1), (1, 4098, 4100), (2, 7, 9), (4, 4095, 4095), (1025, 2050, 2051), (4096, 3, 8)):
        buf = torch.full((rows, ld), float("nan"), dtype=torch.float32, device="cuda")
        _lib.check(L.jh_noise_normal(seed, draw, rows, off, cnt, buf.data_ptr(), ld, current_stream_ptr()), "jh_noise_normal")
        assert torch.equal(buf[:, :cnt], full[:, off : off + cnt]), (off, cnt, ld)
        assert torch.isnan(buf[:, cnt:]).all()  # nothing written beyond the shard's columns
    other = torch.empty_like(full)
    _lib.check(L.jh_noise_normal(seed, draw + 1, rows, 0, n_total, other.data_ptr(), n_total, current_stream_ptr()), "jh_noise_normal")
    assert abs(float(torch.corrcoef(torch.stack([full.flatten(), other.flatten()]))[0, 1])) < 1e-2
    with pytest.raises(ValueError):
        _lib.check(L.jh_noise_normal(seed, draw, rows, 0, 8, full.data_ptr(), 4, current_stream_ptr()), "jh_noise_normal")


def test_product_library_ships_one_kernel_generation(gpu):
    """The product library alone (a fresh process that never loads tests/libjudo_amd_xcheck.so) knows generation 3 only: selecting a cross-check generation is an
    error that says where those kernels live, and the default path runs."""
    import subprocess
    import sys

    code = (
        "import numpy as np\n"
        "from judo_amd.device import GpuModel\n"
        "from judo_amd import _lib\n"
        "from judo_amd.rollout_backend import GpuRolloutBackend\n"
        "for task in ('leap_cube', 'fr3_pick'):\n"
        "    m = GpuModel(task)\n"
        "    for gen in (1, 2):\n"
        "        try:\n"
        "            m.set_kernel(gen)\n"
        "            raise SystemExit(f'{task}: generation {gen} accepted without the test build')\n"
        "        except _lib.JudoAmdError as e:\n"
        "            assert 'libjudo_amd_xcheck' in str(e), e\n"
        "    m.set_kernel(3)\n"
        "    be = GpuRolloutBackend(task, 4)\n"
        "    from judo_amd.tasks import get_registered_tasks\n"
        "    t = get_registered_tasks()[task][0]()\n"
        "    s, y, _ = be.rollout(t.default_state(), np.tile(t.optimizer_warm_start(), (4, 3, 1)))\n"
        "    assert np.isfinite(s).all()\n"
        "print('ok')\n"
    )
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), (r.stdout[-500:], r.stderr[-1500:])
    # and nothing of the product imports the test build
    for dirpath, _, files in os.walk(os.path.join(root, "judo_amd")):
        for f in files:
            if f.endswith(".py"):
                assert "xcheck" not in open(os.path.join(dirpath, f)).read().replace("jh_register_xcheck", ""), f

"""Policy step of the Spot policy rollout on the GPU (jh_policy_step: observation, MFMA actor, control mapping) against the oracle
restatement and the golden vectors.  The actor is fp32 on both sides (the reference casts the observation to float for ONNX
inference); MFMA f32 accumulates in a different order than numpy, hence 1e-4-relative tolerances on O(10) activations."""

import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from tests.conftest import GOLDEN  # noqa: E402


def test_policy_step_matches_golden_and_oracle(gpu):
    import torch
    from judo_amd.policy import SpotLocomotionPolicy, SpotStateLayout
    from oracle import policy as P

    g = np.load(os.path.join(GOLDEN, "spot_policy.npz"))
    nq, nv, bq, bv, lq, lv = (int(x) for x in g["step_layout"])
    lay = SpotStateLayout(nq, nv, bq, bv, lq, lv)
    pol = SpotLocomotionPolicy()
    states = np.concatenate([g["step_qpos"], g["step_qvel"]], axis=1)
    ctrl, out = pol.step(states, g["step_command"], g["step_prev"], lay)
    np.testing.assert_allclose(pol.last_observation.cpu().numpy(), g["step_obs"], rtol=0, atol=2e-6)   # fp32 rotation of O(1) vectors
    np.testing.assert_allclose(out, g["step_out"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(ctrl, g["step_ctrl"], rtol=7e-6, atol=3.5e-6)
    # ragged batch sizes around the 32-row tiles and either side of the switches between the three launch shapes (512 and 2 048 rollouts), torch tensors in / out:
    Ws, bs = P.load_actor()
    rng = np.random.default_rng(3)
    for N in (1, 127, 129, 512, 513, 1000, 2048, 2049, 5000):
        qpos = rng.standard_normal((N, nq)) * 0.3
        qpos[:, 3:7] = rng.standard_normal((N, 4)); qpos[:, 3:7] /= np.linalg.norm(qpos[:, 3:7], axis=1, keepdims=True)
        qvel, cmd, prev = rng.standard_normal((N, nv)), rng.standard_normal((N, 25)) * 0.3, rng.standard_normal((N, 12))
        cmd[:, 10:22] = 0
        obs_ref, ctrl_ref, out_ref = P.policy_step(Ws, bs, qpos, qvel, cmd, prev, base_qpos=bq, base_qvel=bv, leg_qpos=lq, leg_qvel=lv)
        st = torch.as_tensor(np.concatenate([qpos, qvel], 1), dtype=torch.float32, device="cuda")
        c2, o2 = pol.step(st, torch.as_tensor(cmd, dtype=torch.float32, device="cuda"), torch.as_tensor(prev, dtype=torch.float32, device="cuda"), lay)
        assert isinstance(c2, torch.Tensor) and c2.shape == (N, 19) and o2.shape == (N, 12)
        np.testing.assert_allclose(o2.cpu().numpy(), out_ref, rtol=5e-5, atol=5e-5)
        np.testing.assert_allclose(c2.cpu().numpy(), ctrl_ref, rtol=1.5e-5, atol=7.5e-6)
    with pytest.raises(ValueError):
        pol.step(states[:, :-1], g["step_command"], g["step_prev"], lay)
    with pytest.raises(ValueError):
        pol.step(states, g["step_command"][:, :24], g["step_prev"], lay)


def test_policy_step_throughput_sanity(gpu):
    """65 536 rollouts (the headline batch): one policy step is 27.5 GFLOP of f32 GEMM; it has to run at MFMA speed, not VALU speed."""
    import torch
    from judo_amd.policy import SpotLocomotionPolicy, SpotStateLayout

    N = 65536
    pol = SpotLocomotionPolicy()
    lay = SpotStateLayout(33, 31)
    g = torch.Generator(device="cuda").manual_seed(0)
    st = torch.randn((N, 64), device="cuda", generator=g) * 0.3
    st[:, 3:7] = torch.nn.functional.normalize(torch.randn((N, 4), device="cuda", generator=g), dim=1)
    cmd, prev = torch.randn((N, 25), device="cuda", generator=g) * 0.3, torch.randn((N, 12), device="cuda", generator=g)
    for _ in range(2):
        pol.step(st, cmd, prev, lay)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        c, o = pol.step(st, cmd, prev, lay)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    tflops = 2.0 * N * (84 * 512 + 512 * 256 + 256 * 128 + 128 * 12) / (ms * 1e-3) / 1e12
    print(f"policy step N={N}: {ms:.3f} ms, {tflops:.1f} TFLOP/s f32 (MFMA f32 peak 157)")
    assert torch.isfinite(c).all() and tflops > 15.0

"""Action normalisers (judo_amd/normalization.py) and their use in the plan step: the behaviours the reference's own test file pins
(/root/reference/tests/test_controller/test_action_normalization.py:15-215), restated on the build's classes.  The unit tests run on the CPU; the ones that
run a plan step need the GPU (the product has no CPU path)."""

import numpy as np
import pytest

from judo_amd.normalization import IdentityNormalizer, MinMaxNormalizer, RunningMeanStdNormalizer, make_normalizer


def test_identity_passes_values_through():
    rng = np.random.default_rng(0)
    n = IdentityNormalizer(3)
    x = rng.standard_normal((10, 3))
    np.testing.assert_array_equal(n.normalize(x), x)
    np.testing.assert_array_equal(n.denormalize(n.normalize(x)), x)


def test_min_max_maps_the_range_onto_minus_one_plus_one():
    rng = np.random.default_rng(1)
    ends = rng.standard_normal((3, 2))
    lo, hi = ends.min(axis=1), ends.max(axis=1)
    n = MinMaxNormalizer(3, lo, hi)
    np.testing.assert_allclose(n.normalize(lo), -np.ones(3), atol=1e-5)
    np.testing.assert_allclose(n.normalize(hi), np.ones(3), atol=1e-5)
    x = rng.uniform(lo, hi, (10, 3))
    z = n.normalize(x)
    np.testing.assert_allclose(n.denormalize(z), x, atol=1e-9)
    assert np.all(z >= -1 - 1e-6) and np.all(z <= 1 + 1e-6)


@pytest.mark.parametrize("shape", [(30, 3), (20, 2, 3)])
def test_running_statistics_accumulate_over_batches(shape):
    """Batches of any leading shape: count, mean and population std after every update equal those of all data seen so far; normalised data is standardised."""
    rng = np.random.default_rng(2)
    data = rng.standard_normal(shape)
    n = RunningMeanStdNormalizer(3)
    parts = 3 if len(shape) == 2 else 2
    step = shape[0] // parts
    lead = tuple(range(len(shape) - 1))
    for i in range(parts):
        n.update(data[i * step : (i + 1) * step])
        seen = data[: (i + 1) * step]
        assert n.count == seen[..., 0].size
        np.testing.assert_allclose(n.mean, seen.mean(axis=lead), atol=1e-6)
        np.testing.assert_allclose(n.std, seen.std(axis=lead), atol=1e-6)
    z = n.normalize(data)
    np.testing.assert_allclose(z.mean(axis=lead), 0, atol=1e-5)
    np.testing.assert_allclose(z.std(axis=lead), 1, atol=1e-5)
    np.testing.assert_allclose(n.denormalize(z), data, atol=1e-5)


def test_unknown_kind_is_refused():
    with pytest.raises(ValueError):
        make_normalizer("standardise", 2)


def _controller(kind=None, **ckw):
    from judo_amd.config import ControllerConfig
    from judo_amd.controller import make_controller

    ctrl = make_controller("cylinder_push", "cem")
    if kind is not None:
        ctrl.controller_cfg = ControllerConfig(action_normalizer=kind, **ckw)
    ctrl.current_state = np.random.default_rng(3).random(ctrl.task.nq + ctrl.task.nv)
    ctrl.time = 0.0
    return ctrl


@pytest.mark.gpu
def test_a_live_change_of_the_kind_rebuilds_the_normaliser(gpu):
    ctrl = _controller()
    assert isinstance(ctrl.action_normalizer, IdentityNormalizer)
    ctrl.controller_cfg.action_normalizer = "min_max"
    ctrl.update_action()
    assert isinstance(ctrl.action_normalizer, MinMaxNormalizer)


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["none", "min_max", "running"])
def test_every_kind_runs_a_plan_step(gpu, kind):
    ctrl = _controller(kind)
    ctrl.update_action()
    assert np.isfinite(ctrl.nominal_knots).all()


@pytest.mark.gpu
def test_min_max_takes_the_actuator_ranges_and_keeps_candidates_inside(gpu):
    ctrl = _controller("min_max", max_opt_iters=1)
    n, r = ctrl.action_normalizer, ctrl.task.actuator_ctrlrange
    assert isinstance(n, MinMaxNormalizer)
    np.testing.assert_allclose(n.min, r[:, 0]); np.testing.assert_allclose(n.max, r[:, 1])
    ctrl.update_action()
    cand = ctrl.candidate_knots
    assert np.all(cand >= r[:, 0] - 1e-6) and np.all(cand <= r[:, 1] + 1e-6)
    z = n.normalize(cand)
    assert np.all(z >= n.normalize(r[:, 0]) - 1e-6) and np.all(z <= n.normalize(r[:, 1]) + 1e-6)


@pytest.mark.gpu
def test_running_kind_is_fed_with_the_candidates_of_the_plan_step(gpu):
    ctrl = _controller("running", max_opt_iters=1)
    n = ctrl.action_normalizer
    assert isinstance(n, RunningMeanStdNormalizer) and n.count == 0
    ctrl.update_action()
    assert n.count == ctrl.optimizer.num_rollouts * ctrl.optimizer.num_nodes
    cand = ctrl.candidate_knots   # (N, K, nu) raw candidates of this plan step
    np.testing.assert_allclose(n.mean, cand.mean(axis=(0, 1)), atol=1.5e-9)
    np.testing.assert_allclose(n.std, cand.std(axis=(0, 1)), atol=1e-7)

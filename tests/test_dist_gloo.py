"""world_size = 2 over gloo on CPU: the sharding arithmetic, the single all-gather of the per-rank record and the
merge math of the multi-GPU plan step (judo_amd/distributed.py).  There is no GPU here, so the shard-local records the
kernels would produce are computed by the oracle; what is under test is the product's exchange path and the
log-sum-exp merge that `jh_shard_merge` implements on the device.  The device merge itself is exercised through the C ABI by the GPU twins of this test:
tests/test_gpu_dist.py::test_shard_records_through_the_c_abi_merge_match_the_one_gpu_update (G records of jh_update_shard -> jh_shard_merge, against jh_update_fused and the
oracle) and ::test_two_ranks_reproduce_the_single_process_plan_step (two processes, the product's launch -> all-gather -> merge path end to end)."""

import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _mppi_record(knots, costs, lam):
    beta = costs.min()
    w = np.exp(-(costs - beta) / lam)
    return np.concatenate([[beta, w.sum()], (w[:, None, None] * knots).sum(0).reshape(-1)])


def _mppi_merge(recs, lam, KU):
    recs = recs.reshape(-1, 2 + KU)
    beta = recs[:, 0].min()
    e = np.exp(-(recs[:, 0] - beta) / lam)
    return (e[:, None] * recs[:, 2:]).sum(0) / (e * recs[:, 1]).sum()


def _worker(rank, world, port, N, K, nu, lam, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from judo_amd.distributed import all_gather_costs, all_gather_records, shard_rollouts, world_info
    from oracle import oracle as O

    assert world_info() == (world, rank)
    rng = np.random.default_rng(0)  # identical data on every rank; each rank touches only its shard
    knots = rng.standard_normal((N, K, nu))
    costs = np.abs(rng.standard_normal(N)) * 0.1
    sh = shard_rollouts(N, world, rank)
    sl = slice(sh.offset, sh.offset + sh.count)
    rec = torch.from_numpy(_mppi_record(knots[sl], costs[sl], lam))
    allrec = all_gather_records(rec).numpy()
    assert allrec.shape == (world * (2 + K * nu),)
    nominal = _mppi_merge(allrec, lam, K * nu).reshape(K, nu)
    ref = O.mppi_update(knots, -costs, lam)
    np.testing.assert_allclose(nominal, ref, rtol=1e-10, atol=1e-12)
    # elite records: k x (cost, global index, knots)
    k = 3
    order = sorted(range(sh.count), key=lambda i: (costs[sl][i], -(sh.offset + i)))[:k]
    erec = np.concatenate([np.concatenate([[costs[sl][i], sh.offset + i], knots[sl][i].reshape(-1)]) for i in order])
    allel = all_gather_records(torch.from_numpy(erec)).numpy().reshape(world * k, 2 + K * nu)
    best = sorted(range(world * k), key=lambda r: (allel[r, 0], -allel[r, 1]))[:k]
    elite = allel[best, 2:].reshape(k, K, nu)
    ref_nom, ref_sig, ref_idx = O.cem_update(knots, -costs, k, 0.01, 0.3)
    assert sorted(int(allel[r, 1]) for r in best) == sorted(ref_idx.tolist())
    np.testing.assert_allclose(elite.mean(0), ref_nom, rtol=1e-12)
    np.testing.assert_allclose(np.clip(elite.std(0), 0.01, 0.3), ref_sig, rtol=1e-12)
    full = all_gather_costs(torch.from_numpy(costs[sl].copy()), sh).numpy()
    np.testing.assert_allclose(full, costs)
    np.save(os.path.join(out_dir, f"nominal_{rank}.npy"), nominal)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("N", [64, 67])
def test_two_rank_exchange_and_merge(tmp_path, N):
    world, port = 2, 29500 + (os.getpid() % 2000) + N
    mp.spawn(_worker, args=(world, port, N, 4, 3, 0.0025, str(tmp_path)), nprocs=world, join=True)
    a, b = np.load(tmp_path / "nominal_0.npy"), np.load(tmp_path / "nominal_1.npy")
    np.testing.assert_array_equal(a, b)  # every rank holds the identical nominal without a broadcast

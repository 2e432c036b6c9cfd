"""The sharded plan step end to end on the GPU: two ranks (two processes sharing cuda:0, gloo rendezvous) run the product path --
shard-local rollout kernels, shard-local update records, one all-gather, the merge kernel -- and must produce the nominal a single
process produces from the same noise.  (The driver's multi-GPU runs use the nccl backend; what differs is only the transport of the
per-rank record.)"""

import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _knots(task):
    return 3 if task.startswith("spot") else 4   # the shipped num_nodes overrides


def _plan(task, opt, N, noise, group=None):
    import torch
    from judo_amd.controller import make_controller

    ctrl = make_controller(task, opt, group=group)
    ctrl.optimizer.config.num_rollouts = N
    ctrl.controller_cfg.horizon = 16 * ctrl.task.dt
    ctrl.reset()
    ctrl.current_state = ctrl.task.default_state()
    ctrl.system_metadata = {"goal_quat": np.array([0.0, 1.0, 0.0, 0.0])} if task == "leap_cube" else {}  # (the task draws a random goal per instance)
    if ctrl.task.uses_locomotion_policy:  # the Spot policy rollout: policy step (MFMA GEMMs) + tree kernel per control step, sharded like everything else
        ctrl.rollout_cutoff_time = None
        ctrl.task.config.goal_position = np.array([1.0, 0.5, 0.52])
    if noise is None:
        ctrl.optimizer.seed(77)   # device noise: every rank draws all rollouts' noise from the same seed and keeps its shard's columns
    else:
        ctrl.optimizer.injected_noise = noise
    ctrl.keep_candidates = True   # the clipped candidates of the shard, written by the rollout kernel with the row stride of the (possibly full-width) noise draw
    ctrl.update_action()
    torch.cuda.synchronize()
    sig = np.asarray(ctrl.optimizer.sigma, dtype=np.float64) if opt == "cem" else np.zeros(1)
    cand = ctrl.candidate_knots_device.permute(2, 0, 1).cpu().numpy()
    assert cand.shape[0] == ctrl.last_shard.count
    tr = ctrl.traces  # the elites of ALL shards: their trace rows travel with the per-rank records (jh_trace_gather + all-gather)
    return ctrl.nominal_knots.copy(), sig, ctrl.last_shard, -ctrl.rewards_local, cand, (np.zeros((0, 2, 3)) if tr is None else tr.copy())


def _worker(rank, world, port, cases, out_dir):
    import torch
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    for i, (task, opt, N, nu, seed) in enumerate(cases):
        noise = None if seed < 0 else np.random.default_rng(seed).standard_normal((N - 1, _knots(task), nu)).astype(np.float32)
        nom, sig, shard, costs, cand, tr = _plan(task, opt, N, noise, group=dist.group.WORLD)
        assert (shard.world, shard.rank) == (world, rank) and shard.count in (N // world, N // world + 1)
        np.savez(os.path.join(out_dir, f"case{i}_rank{rank}.npz"), nom=nom, sig=sig, costs=costs, cand=cand, tr=tr)
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_reproduce_the_single_process_plan_step(gpu, tmp_path):
    import torch.multiprocessing as mp

    cases = [("cartpole", "mppi", 257, 1, 1), ("cylinder_push", "cem", 128, 2, 2), ("leap_cube", "mppi", 130, 16, 3), ("fr3_pick", "cem", 96, 8, 4),
             ("cartpole", "ps", 64, 1, 5), ("spot_navigate", "mppi", 49, 3, 6),
             ("leap_cube", "mppi", 131, 16, -1), ("fr3_pick", "cem", 97, 8, -1)]   # seed < 0: noise drawn on the device, same seed on every rank
    world, port = 2, 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(world, port, cases, str(tmp_path)), nprocs=world, join=True)
    for i, (task, opt, N, nu, seed) in enumerate(cases):
        noise = None if seed < 0 else np.random.default_rng(seed).standard_normal((N - 1, _knots(task), nu)).astype(np.float32)
        nom1, sig1, _, costs1, cand1, tr1 = _plan(task, opt, N, noise)
        r0, r1 = np.load(tmp_path / f"case{i}_rank0.npz"), np.load(tmp_path / f"case{i}_rank1.npz")
        np.testing.assert_array_equal(r0["nom"], r1["nom"])  # identical on every rank without a broadcast
        np.testing.assert_array_equal(r0["sig"], r1["sig"])
        # the shards cover the rollouts of the single-process run: same noise rows -> same costs, rank-major
        costs2 = np.concatenate([r0["costs"], r1["costs"]])
        assert costs2.shape == costs1.shape
        # every kernel is bit-reproducible and independent of where a rollout sits in its wave (test_gpu_edges.py): the costs are identical
        np.testing.assert_array_equal(costs2, costs1)
        # the kept candidates of the two shards are the single process's, rank-major (a sharded device draw is a column view of the full draw:
        # the kernel's row stride is the full width, the buffer must have it too -- ADVICE round 2)
        np.testing.assert_array_equal(np.concatenate([r0["cand"], r1["cand"]]), cand1)
        # traces: the same elites (chosen among both shards' rollouts) and the same trace rows on every rank and in the single-process run
        np.testing.assert_array_equal(r0["tr"], r1["tr"])
        np.testing.assert_array_equal(r0["tr"], tr1)
        assert tr1.shape[0] > 0 or task.startswith("caltech")
        # vs one process only the reduction is regrouped (two block records instead of one): fp32 summation order in the MPPI average
        np.testing.assert_allclose(r0["nom"], nom1, rtol=0, atol=1.5e-6 if opt == "mppi" else 0)  # observed 2.4e-7
        np.testing.assert_allclose(r0["sig"], sig1, rtol=1e-6, atol=1e-7)


def test_bench_two_ranks_sharing_the_gpu(gpu):
    """`python bench.py --gpus 2` end to end on a one-GPU box: the launcher spawns two ranks (test hook: both on cuda:0, gloo rendezvous), the settle / warm-up /
    timed steps run the same number of collectives on both, rank 0 prints one contract line for the two-rank job whose plan matches the one-rank run's."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, JUDO_BENCH_SHARED_GPU="1")
    env.pop("WORLD_SIZE", None)
    lines = {}
    for n in (1, 2):
        cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", str(n), "--task", "leap_cube", "--rollouts", "1024", "--horizon-steps", "8", "--steps", "3", "--warmup", "1",
               "--settle", "0.05", "--no-cpu-baseline", "--no-cube-only"]
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=root)
        assert r.returncode == 0, r.stderr[-2000:]
        js = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        assert len(js) == 1, r.stdout[-2000:]
        lines[n] = json.loads(js[0])
    assert lines[2]["n_gpus"] == 2 and lines[1]["n_gpus"] == 1 and lines[2]["steps"] == 3 and lines[2]["value"] > 0
    assert lines[2]["config"]["parallelism"] == "rollout-shard x2" and lines[2]["config"]["rollouts"] == 1024
    assert "cpu_baseline" not in lines[2] and "roofline" in lines[2]


@pytest.mark.parametrize("mode,N,G,k,E", [(0, 1000, 3, 0, 2), (1, 1000, 3, 3, 2), (1, 257, 8, 1, 1), (0, 8192, 8, 0, 5), (1, 7, 3, 3, 5)])
def test_shard_records_through_the_c_abi_merge_match_the_one_gpu_update(gpu, mode, N, G, k, E):
    """The sharded plan step's update through the C ABI, in one process: jh_update_shard on G contiguous shards of the same costs / noise writes G records, their
    concatenation (what the all-gather delivers, rank-major) goes through jh_shard_merge -- and must give what jh_update_fused gives on the unsharded arrays and what
    the oracle's update gives: elites, sigma and trace records bit for bit (selection and copies), the MPPI average to fp32 summation order."""
    import torch

    from judo_amd import _lib
    from judo_amd.distributed import shard_rollouts
    from oracle import oracle as O

    L = _lib.lib()
    K, nu, row = 4, 3, 10
    KU = K * nu
    rng = np.random.default_rng(100 * mode + N + G)
    dev = gpu
    noise = torch.from_numpy(rng.standard_normal((K, nu, N)).astype(np.float32)).to(dev)
    nominal = torch.from_numpy(rng.standard_normal(KU).astype(np.float32)).to(dev)
    sigma = torch.from_numpy((0.1 + rng.random(KU)).astype(np.float32)).to(dev)
    lohi = torch.from_numpy(np.concatenate([-1.5 * np.ones(nu), 1.5 * np.ones(nu)]).astype(np.float32)).to(dev)
    costs_np = (np.abs(rng.standard_normal(N)) * 0.01).astype(np.float32)
    costs_np[N // 3] = costs_np[N // 2]  # a tie among the candidates
    costs = torch.from_numpy(costs_np).to(dev)
    trace = torch.from_numpy(rng.standard_normal((N, row)).astype(np.float32)).to(dev)  # row-major trace buffer, one row per rollout
    lam, tie = 0.0025, 1
    st = torch.cuda.current_stream().cuda_stream
    # one GPU: jh_update_fused
    scr = torch.zeros(int(L.jh_update_fused_scratch_floats(N, K, nu)), dtype=torch.float32, device=dev)
    out1 = torch.zeros(2 * KU + E * (2 + row), dtype=torch.float32, device=dev)
    p = out1.data_ptr()
    _lib.check(L.jh_update_fused(costs.data_ptr(), None, nominal.data_ptr(), noise.data_ptr(), N, sigma.data_ptr(), lohi.data_ptr(), N, 0, K, nu, mode, lam, k, tie, E, trace.data_ptr(), row, 0,
                                 scr.data_ptr(), p, p + 4 * KU, p + 8 * KU, st), "jh_update_fused")
    # G shards -> G records -> merge
    Lrec = int(L.jh_shard_record_floats(K, nu, mode, k, E, row))
    recs = torch.zeros(G * Lrec, dtype=torch.float32, device=dev)
    for g in range(G):
        sh = shard_rollouts(N, G, g)
        scr_g = torch.zeros(int(L.jh_update_fused_scratch_floats(sh.count, K, nu)), dtype=torch.float32, device=dev)
        _lib.check(L.jh_update_shard(costs.data_ptr() + 4 * sh.offset, None, nominal.data_ptr(), noise.data_ptr() + 4 * sh.offset, N, sigma.data_ptr(), lohi.data_ptr(), sh.count, sh.offset,
                                     K, nu, mode, lam, k, tie, E, trace.data_ptr() + 4 * row * sh.offset, row, 0, scr_g.data_ptr(), recs.data_ptr() + 4 * g * Lrec, st), "jh_update_shard")
        torch.cuda.synchronize()
    outG = torch.zeros_like(out1)
    q = outG.data_ptr()
    _lib.check(L.jh_shard_merge(recs.data_ptr(), G, K, nu, mode, lam, k, tie, E, row, q, q + 4 * KU, q + 8 * KU, st), "jh_shard_merge")
    torch.cuda.synchronize()
    a, b = out1.cpu().numpy(), outG.cpu().numpy()
    # the candidates the kernels see: clip(nominal + sigma * noise), global sample 0 = the nominal
    cand = np.clip(nominal.cpu().numpy()[None] + sigma.cpu().numpy()[None] * noise.cpu().numpy().reshape(KU, N).T, -1.5, 1.5).astype(np.float64)
    cand[0] = np.clip(nominal.cpu().numpy(), -1.5, 1.5)
    cand = cand.reshape(N, K, nu)
    if mode == 0:
        np.testing.assert_allclose(b[:KU], a[:KU], rtol=0, atol=6e-7)  # regrouped fp32 sums
        np.testing.assert_allclose(b[:KU].reshape(K, nu), O.mppi_update(cand, -costs_np.astype(np.float64), lam), rtol=0, atol=1e-6)
    else:
        np.testing.assert_array_equal(b[: 2 * KU], a[: 2 * KU])
        ref_nom, ref_sig, ref_idx = O.cem_update(cand, -costs_np.astype(np.float64), min(k, N), 0.0, np.inf)
        np.testing.assert_allclose(b[:KU].reshape(K, nu), ref_nom, rtol=7e-7, atol=7e-8)
        np.testing.assert_allclose(b[KU : 2 * KU].reshape(K, nu), ref_sig, rtol=2e-6, atol=2e-8)
    # trace records: the E best rollouts (ties: higher index first), their rows, bit for bit -- and the right ones
    np.testing.assert_array_equal(b[2 * KU :].view(np.int32), a[2 * KU :].view(np.int32))
    rec = b[2 * KU :].reshape(E, 2 + row)
    order = sorted(range(N), key=lambda i: (costs_np[i], -i))[:E]
    for e, i in enumerate(order):
        assert rec[e, 1:2].view(np.int32)[0] == i and rec[e, 0] == costs_np[i]
        np.testing.assert_array_equal(rec[e, 2:], trace.cpu().numpy()[i])
    for e in range(len(order), E):  # fewer rollouts than trace elites: empty records
        assert rec[e, 1:2].view(np.int32)[0] == -1 and np.isinf(rec[e, 0])

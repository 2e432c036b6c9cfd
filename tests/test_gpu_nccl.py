"""Multi-GPU: the product's exchange on the `nccl` backend (RCCL over xGMI), one process per GPU.  Needs >= 2 visible GPUs; the 1-GPU
box of the round-end test tier skips the RCCL cases and runs only the launcher's refusal (its gloo counterparts are
tests/test_gpu_dist.py -- two ranks sharing one GPU -- and tests/test_dist_gloo.py on CPU)."""

import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _plan(task, opt, N, group=None):
    import torch
    from judo_amd.controller import make_controller

    ctrl = make_controller(task, opt, group=group)
    ctrl.optimizer.config.num_rollouts = N
    ctrl.controller_cfg.horizon = 16 * ctrl.task.dt
    ctrl.reset()
    ctrl.current_state = ctrl.task.default_state()
    ctrl.system_metadata = {"goal_quat": np.array([0.0, 1.0, 0.0, 0.0])} if task == "leap_cube" else {}
    ctrl.optimizer.seed(77)
    for step in range(2):
        ctrl.time = 0.05 * step
        ctrl.update_action()
    torch.cuda.synchronize()
    sig = np.asarray(ctrl.optimizer.sigma, dtype=np.float64) if opt == "cem" else np.zeros(1)
    return ctrl.nominal_knots.copy(), sig, -ctrl.rewards_local, ctrl.traces.copy()


def _worker(rank, world, port, cases, out_dir):
    import torch
    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    for i, (task, opt, N) in enumerate(cases):
        nom, sig, costs, traces = _plan(task, opt, N, group=dist.group.WORLD)
        np.savez(os.path.join(out_dir, f"case{i}_rank{rank}.npz"), nom=nom, sig=sig, costs=costs, traces=traces)
    dist.barrier()
    dist.destroy_process_group()


def test_rccl_sharded_plan_steps_reproduce_one_gpu(gpu, tmp_path):
    import torch
    import torch.multiprocessing as mp

    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (RCCL refuses two ranks on one device)")
    cases = [("leap_cube", "mppi", 1024), ("fr3_pick", "cem", 512), ("cartpole", "ps", 64), ("cylinder_push", "mppi", 4097)]
    world, port = 2, 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(world, port, cases, str(tmp_path)), nprocs=world, join=True)
    for i, (task, opt, N) in enumerate(cases):
        nom1, sig1, costs1, traces1 = _plan(task, opt, N)
        r0, r1 = np.load(tmp_path / f"case{i}_rank0.npz"), np.load(tmp_path / f"case{i}_rank1.npz")
        np.testing.assert_array_equal(r0["nom"], r1["nom"])  # identical on every rank, no broadcast
        np.testing.assert_array_equal(np.concatenate([r0["costs"], r1["costs"]]), costs1)  # shard-invariant noise, bit-reproducible kernels
        np.testing.assert_allclose(r0["nom"], nom1, rtol=0, atol=5e-6 if opt == "mppi" else 0)
        np.testing.assert_allclose(r0["sig"], sig1, rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(r0["traces"], traces1, rtol=0, atol=1e-6)  # global elites, whichever rank they ran on
        np.testing.assert_array_equal(r0["traces"], r1["traces"])


def _worker_one(rank, port, cases, out_dir):
    import torch
    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    from judo_amd.controller import make_controller

    for i, (task, opt, N) in enumerate(cases):
        ctrl = make_controller(task, opt, group=dist.group.WORLD)
        ctrl.force_shard_path = True  # launch (rollout + this rank's record) -> all_gather_into_tensor on the nccl backend -> merge, with one rank
        ctrl.optimizer.config.num_rollouts = N
        ctrl.controller_cfg.horizon = 16 * ctrl.task.dt
        ctrl.reset()
        ctrl.current_state = ctrl.task.default_state()
        ctrl.system_metadata = {"goal_quat": np.array([0.0, 1.0, 0.0, 0.0])} if task == "leap_cube" else {}
        ctrl.optimizer.seed(77)
        for step in range(2):
            ctrl.time = 0.05 * step
            ctrl.update_action()
        torch.cuda.synchronize()
        assert ctrl._side_stream is not None  # the next iteration's noise went to the second stream in front of the collective
        sig = np.asarray(ctrl.optimizer.sigma, dtype=np.float64) if opt == "cem" else np.zeros(1)
        np.savez(os.path.join(out_dir, f"one{i}.npz"), nom=ctrl.nominal_knots, sig=sig, costs=-ctrl.rewards_local, traces=ctrl.traces)
    dist.barrier()
    dist.destroy_process_group()


def test_rccl_world_size_one_runs_the_sharded_plan_step(gpu, tmp_path):
    """RCCL on the ONE GPU every box has: a process group of one rank on the `nccl` backend, the controller forced onto the sharded path -- `jh_plan_step_shard`, the
    device branch of `all_gather_records` (`all_gather_into_tensor` through RCCL), `jh_plan_merge` over G = 1 records -- reproduces the one-call plan step: the same costs bit
    for bit, the same nominal, sigma and traces.  (Two ranks need two devices: RCCL refuses two ranks on one; that case is the test above.)"""
    import torch.multiprocessing as mp

    cases = [("cartpole", "mppi", 4096), ("cylinder_push", "cem", 1000), ("leap_cube", "mppi", 512), ("fr3_pick", "cem", 256)]
    port = 29700 + (os.getpid() % 2000)
    mp.spawn(_worker_one, args=(port, cases, str(tmp_path)), nprocs=1, join=True)
    for i, (task, opt, N) in enumerate(cases):
        nom1, sig1, costs1, traces1 = _plan(task, opt, N)
        r = np.load(tmp_path / f"one{i}.npz")
        np.testing.assert_array_equal(r["costs"], costs1)
        np.testing.assert_allclose(r["nom"], nom1, rtol=0, atol=5e-6 if opt == "mppi" else 0)
        np.testing.assert_allclose(r["sig"], sig1, rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(r["traces"], traces1, rtol=0, atol=1e-6)


def test_bench_launches_its_own_ranks_or_refuses(gpu):
    """`python bench.py --gpus 2` with no launcher around it: two ranks over RCCL when two GPUs are there (one JSON line, n_gpus 2),
    a non-zero exit and no JSON line when they are not."""
    import torch

    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--rollouts", "4096", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True, timeout=600)
    if torch.cuda.device_count() < 2:
        assert r.returncode != 0 and "only 1 GPU" in r.stderr and not r.stdout.strip()
        return
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["config"]["rollouts"] == 4096 and line["value"] > 0

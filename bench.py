#!/usr/bin/env python3
"""Benchmark of the sampling-MPC plan step on MI355X (driver contract: one JSON line on rank 0).

A "step" is one full plan step (`Controller.update_action`): sample -> clip -> spline -> rollout -> cost -> update,
including the all-gather when several GPUs take part and the device->host copy of the new nominal knots.
Default workload = BASELINE.json's metric configuration: leap_cube MPPI, 65 536 rollouts x H = 64 (K = 4, cubic,
sigma = 0.2 ramp 4, lambda = 0.0025), synthetic standard-normal noise drawn on the device (seed 1234, see --seed; every rank keeps its shard of the same draw), inputs
resident in HBM; the plan time advances 0.05 s per step (control_freq 20 Hz).  With N GPUs the 65 536 rollouts are
sharded (strong scaling: total work fixed) -- one process per GPU, one RCCL all-gather of a 66-float record per step.

Extra objects on the JSON line:
  roofline     dominant kernel = the fused rollout kernel; achieved = algorithmic bytes per launch
               ((4*K*nu noise + 4 cost) bytes x rollouts on this GPU) / its mean duration measured
               with HIP events on the launch stream; peak = 8 TB/s HBM3E.
  cpu_baseline the fp64 oracle (oracle/, kind "port": MuJoCo itself is not installable here) rolling out a bounded
               sample of the same workload on all host cores, rank 0 at N=1 only.
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

WORKLOADS = {
    # task: (optimizer, rollouts, horizon steps)  -- BASELINE.json configs[1..4]
    "cartpole": ("mppi", 4096, 64),
    "cylinder_push": ("mppi", 16384, 64),
    "fr3_pick": ("cem", 32768, 40),
    "leap_cube": ("mppi", 65536, 64),
    # not a BASELINE config: the Spot policy rollout at the headline batch, the shipped 2 s horizon = 100 control steps x 2 physics substeps
    "spot_navigate": ("mppi", 65536, 100),
}
HBM_PEAK_GBS = 8000.0
TRAFFIC_FILE = "r06_traffic.json"


def usable_cpus() -> int:
    """CPUs this process may actually use: the affinity mask capped by the cgroup CPU quota (the GPU box reports 256 hardware threads
    but runs the job under a 16-CPU quota; 256 oracle threads there are 3x slower than 16)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                quota, period = txt[0], float(txt[1])
            else:
                quota, period = txt[0], float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota not in ("max", "-1") and float(quota) > 0:
                n = min(n, max(1, int(round(float(quota) / period))))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, n)


def cpu_baseline_reference(task: str, ctrl, cores: int, seconds_target: float = 15.0) -> dict | None:
    """The reference's own CPU path -- `mujoco.rollout.Rollout(nthread=cores)` driven as judo/utils/mj_rollout_backend.py:36-88 drives it -- on a bounded
    sample of the same workload, when the `mujoco` wheel and the task's MJCF are reachable (oracle/mujoco_probe.py).  None otherwise: MuJoCo is absent
    from this image and from the GPU box, and the leap / fr3 MJCF need mesh assets that are not in the repository."""
    from oracle import mujoco_probe as MP

    if task not in MP.MESH_FREE_TASKS or not MP.available(task):
        return None
    K, nu, H = ctrl.optimizer.num_nodes, ctrl.nu, ctrl.num_timesteps
    rng = np.random.default_rng(0)
    n = 64 * cores
    while True:
        U = np.repeat(rng.standard_normal((n, (H + 3) // 4, nu)), 4, axis=1)[:, :H]
        r = MP.time_reference_rollouts(task, np.asarray(ctrl.task.default_state()), U, cores)
        if r["seconds"] > seconds_target / 4 or n >= 65536:
            break
        n = int(min(65536, max(2 * n, n * (seconds_target / 2) / max(r["seconds"], 1e-3))))
    return {"value": n / r["seconds"], "unit": "rollouts/s", "cores": cores, "kind": "reference",
            "sample": f"{n} rollouts x H={H} through mujoco.rollout.Rollout(nthread={cores}) (MuJoCo {r['mujoco']}), {r['seconds']:.1f} s"}


def cpu_baseline(task: str, ctrl, seconds_target: float = 15.0) -> dict:
    """The reference's CPU rollout when MuJoCo is reachable (kind "reference"); otherwise oracle rollouts (threaded C, fp64; kind "port") on the host
    cores for a bounded sample of the same workload."""
    from oracle import oracle as O
    from tests.harness import oracle_plan_step

    cores = usable_cpus()
    ref = cpu_baseline_reference(task, ctrl, cores, seconds_target)
    if ref is not None:
        return ref
    om = O.Model(task)
    K, nu, H = ctrl.optimizer.num_nodes, ctrl.nu, ctrl.num_timesteps
    rng = np.random.default_rng(0)
    nom = np.tile(ctrl.task.optimizer_warm_start(), (K, 1))
    n = 4 * cores
    saved = ctrl.optimizer.config.num_rollouts
    total_rollouts, total_t = 0, 0.0
    while True:  # grow the sample until it takes a meaningful time, bounded by seconds_target
        ctrl.optimizer.config.num_rollouts = n
        noise = rng.standard_normal((n - 1, K, nu))
        t0 = time.perf_counter()
        cem_sigma = ctrl.optimizer.sigma.copy() if hasattr(ctrl.optimizer, "sigma_min") else None
        oracle_plan_step(om, ctrl, nom, noise, cem_sigma=cem_sigma, nthread=cores)
        dt = time.perf_counter() - t0
        total_rollouts, total_t = n, dt
        if dt > seconds_target / 4 or n >= 65536:
            break
        n = int(min(65536, max(2 * n, n * (seconds_target / 2) / max(dt, 1e-3))))
    ctrl.optimizer.config.num_rollouts = saved
    return {"value": total_rollouts / total_t, "unit": "rollouts/s", "cores": cores, "kind": "port", "mujoco": "not installed (oracle/mujoco_probe.py)",
            "sample": f"{total_rollouts} rollouts x H={H} of the same plan step (fp64 oracle engine, {cores} pthreads = usable CPUs of {os.cpu_count()} hardware threads), {total_t:.1f} s"}


def cpu_baseline_policy(ctrl, seconds_target: float = 15.0) -> dict:
    """The oracle's policy rollout (numpy actor + fp64 engine, one thread: `oracle.policy.policy_rollout` is a per-rollout Python loop) on a bounded sample."""
    from oracle import policy as P

    om = P.spot_model(self_collision=True)  # the same model the kernel steps by default: the robot's own 287 contact pairs besides the ground
    Ws, bs = P.load_actor()
    H = ctrl.num_timesteps
    rng = np.random.default_rng(0)
    x0 = np.asarray(ctrl.current_state, dtype=np.float64)
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds_target / 2:
        cmds = np.tile(P.DEFAULT_POLICY_COMMAND, (H, 1))
        cmds[:, :3] = rng.uniform(-0.5, 0.5, 3)
        P.policy_rollout(om, Ws, bs, x0, cmds, physics_substeps=ctrl.task.physics_substeps)
        n += 1
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "rollouts/s", "cores": 1, "kind": "port",
            "sample": f"{n} rollouts x {H} control steps x {ctrl.task.physics_substeps} substeps (numpy actor + fp64 oracle engine, 1 thread), {dt:.1f} s"}


def materialize_line(args, torch, world: int, rank: int) -> None:
    """One step = one `rollout` of N rollouts x H steps with controls resident in HBM; states and sensors are written once."""
    from judo_amd.rollout_backend import GpuRolloutBackend
    from judo_amd.distributed import shard_rollouts

    N = args.rollouts or (65536 if args.task in ("leap_cube", "fr3_pick") else 1 << 20)
    H = args.horizon_steps or WORKLOADS[args.task][2]
    n_local = shard_rollouts(N, world, rank).count
    be = GpuRolloutBackend(args.task, n_local)
    gm = be.model
    from judo_amd.tasks import get_registered_tasks

    task = get_registered_tasks()[args.task][0]()
    x0 = torch.as_tensor(np.asarray(task.default_state(), dtype=np.float32), device=gm.device)
    # controls = the task's warm-start command + N(0, 0.5^2) per step
    U = 0.5 * torch.randn((n_local, H, gm.nu), device=gm.device, generator=torch.Generator(device=gm.device).manual_seed(1234 + rank))
    U += torch.as_tensor(np.asarray(task.optimizer_warm_start(), dtype=np.float32), device=gm.device)
    for _ in range(args.warmup):
        be.rollout_device(x0, U)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        be.rollout_device(x0, U)
    ev1.record()
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if SHARED_GPU_TEST else "cuda")
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
    kern_ms = ev0.elapsed_time(ev1) / args.steps
    alg = 4 * n_local * H * (gm.nx + gm.ns + gm.nu)
    achieved = alg / (kern_ms * 1e-3) / 1e9
    if rank == 0:
        print(json.dumps({
            "metric": f"rollouts/sec, {args.task} materialise mode (RolloutBackend.rollout)", "value": N * args.steps / elapsed, "unit": "rollouts/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.task} materialise {N} rollouts x H={H} (nx={gm.nx}, ns={gm.ns}, nu={gm.nu})", "rollouts": N, "horizon_steps": H,
                       "parallelism": f"rollout-shard x{world}"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "kernel": "rollout materialise", "kernel_ms": kern_ms, "algorithmic_bytes_per_launch": alg}}))
    if world > 1:
        torch.distributed.destroy_process_group()


# Test hook (tests/test_gpu_nccl.py on a one-GPU box): all ranks share cuda:0 and rendezvous over gloo; everything else -- the launcher, the rank environment, the
# shard-local kernels, the per-rank record exchange, barrier + max-over-ranks timing, rank 0's line -- is the code the multi-GPU run executes.
SHARED_GPU_TEST = os.environ.get("JUDO_BENCH_SHARED_GPU") == "1"


def launch_ranks(n: int, n_devices: int) -> int:
    """`python bench.py --gpus N` without a launcher: re-run this script as N ranks, one per GPU, over RCCL
    (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...`).  Returns the exit status."""
    import socket
    import subprocess

    if n_devices < n and not SHARED_GPU_TEST:
        print(f"bench.py: --gpus {n} but only {n_devices} GPU(s) are visible; refusing to run a smaller job under that label", file=sys.stderr)
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--task", default="leap_cube", choices=sorted(WORKLOADS))
    ap.add_argument("--optimizer", default=None)
    ap.add_argument("--rollouts", type=int, default=None)
    ap.add_argument("--horizon-steps", type=int, default=None)
    # The bench closes the loop through the plan, so the noise stream decides which contact situations the plan visits and with them the cost of a plan step -- and the
    # loop is chaotic in the last bits: the same eight seeds gave 72.4-90.5 ms (mean 81.0) on one build of round 3 and 74.0-87.8 (mean 81.9) on the next, with every
    # seed's own number reshuffled (profiles/r03_seed_sweep.txt).  No seed stays representative across builds; the default is the one all rounds used.
    ap.add_argument("--seed", type=int, default=1234, help="seed of the optimizer's device noise stream (the same on every rank)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-self-collision", action="store_true", help="leap_cube: the cube's contacts only (round-1 model), not the hand's own")
    ap.add_argument("--settle", type=float, default=0.4, help="seconds of untimed plan steps on a throw-away plan before the W warm-up steps (runtime one-offs)")
    ap.add_argument("--no-with-traces", action="store_true", help="skip the extra plan steps that measure what reading Controller.traces costs (run after the timed region)")
    ap.add_argument("--traces-outside-step", action="store_true", help="do not read Controller.traces inside the timed plan steps (rounds 1-2 timed it that way)")
    ap.add_argument("--no-cube-only", action="store_true", help="leap_cube: skip the extra cube-contacts-only steps run after the timed region")
    ap.add_argument("--no-steady-state", action="store_true", help="skip the 10 + 100 plan steps of the reference's benchmark statistic run after the timed region (benchmark_100; steady_state: its last 20 plan steps)")
    ap.add_argument("--no-replay", action="store_true", help="skip the replay of the recorded plan inputs (the deterministic, round-to-round comparable figure) after the timed region")
    ap.add_argument("--mode", default="fused", choices=["fused", "materialize"],
                    help="fused = the plan step (headline); materialize = drop-in RolloutBackend.rollout writing every state/sensor (the HBM-bound exhibit, SURVEY 8d)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "WORLD_SIZE" in os.environ and world != args.gpus:  # never run a smaller (or larger) job under the --gpus label
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU (plain `python bench.py --gpus N` spawns them itself)")

    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: torch.cuda.is_available() is False (no CPU fallback exists)")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(launch_ranks(args.gpus, torch.cuda.device_count()))
    if SHARED_GPU_TEST:
        local_rank = 0
    if local_rank >= torch.cuda.device_count():
        raise SystemExit(f"bench.py: LOCAL_RANK {local_rank} but only {torch.cuda.device_count()} GPU(s) visible")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if SHARED_GPU_TEST:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    if args.mode == "materialize":
        return materialize_line(args, torch, world, rank)

    from judo_amd.controller import make_controller

    opt_name, N, H = WORKLOADS[args.task]
    opt_name = args.optimizer or opt_name
    N = args.rollouts or N
    H = args.horizon_steps or H
    ctrl = make_controller(args.task, opt_name)
    # JUDO_BENCH_RCCL_ONE_RANK=1 (one GPU): a process group of ONE rank on the nccl backend and the controller on the sharded path -- launch -> all_gather_into_tensor through
    # RCCL -> merge -- so that per_rank.exchange_ms is the exchange's cost with the real transport (its floor: no peer to wait for).  An exhibit, not the driver's line.
    rccl_one = os.environ.get("JUDO_BENCH_RCCL_ONE_RANK") == "1" and world == 1
    if rccl_one:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29791")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", local_rank))
        ctrl.group = dist.group.WORLD
        ctrl.force_shard_path = True
    ctrl.optimizer.config.num_rollouts = N
    ctrl.controller_cfg.horizon = H * ctrl.task.dt
    ctrl.reset()
    ctrl.current_state = ctrl.task.default_state()
    ctrl.system_metadata = {"goal_quat": np.array([0.0, 1.0, 0.0, 0.0])} if args.task == "leap_cube" else {}
    if args.no_self_collision and ctrl.model is not None:
        ctrl.model.set_self_collision(False)
    ctrl.optimizer.seed(args.seed)  # the same seed on every rank: each rank slices its shard out of the same noise, the plan does not depend on --gpus
    is_policy = ctrl.task.uses_locomotion_policy
    if is_policy:
        ctrl.rollout_cutoff_time = None  # throughput run: no 125 ms deadline
        ctrl.task.config.goal_position = np.array([2.0, 1.0, 0.52])
    K, nu = ctrl.optimizer.num_nodes, ctrl.nu
    assert ctrl.num_timesteps == H

    def barrier():
        if world > 1:
            dist.barrier()

    ctrl.record_kernel_events = True  # (already during the warm-up: the first timed HIP event costs ~40 ms of one-off initialisation)
    # settle: some 0.1 s into a process's first GPU work the ROCm runtime spends one ~35 ms stall (seen in about half of the runs, at a random early plan step);
    # plan steps on a throw-away plan until --settle seconds have passed keep it out of the W + K steps, which then start from the initial state again
    if args.settle > 0:
        ctrl.update_action()  # (one-off initialisation)
        torch.cuda.synchronize()
        t_settle = time.perf_counter()
        ctrl.update_action()
        torch.cuda.synchronize()
        n_settle = int(min(2000, max(0, np.ceil(args.settle / max(time.perf_counter() - t_settle, 1e-5)) - 1)))
        if world > 1:  # every plan step is a collective: all ranks must run the same number of them
            t = torch.tensor([n_settle], dtype=torch.int64, device="cpu" if SHARED_GPU_TEST else "cuda")
            dist.broadcast(t, 0)
            n_settle = int(t.item())
        for _ in range(n_settle):
            ctrl.update_action()
        torch.cuda.synchronize()
    ctrl.reset()
    ctrl.current_state = ctrl.task.default_state()
    ctrl.optimizer.seed(args.seed)
    t_plan = 0.0
    for _ in range(args.warmup):
        ctrl.time = t_plan
        ctrl.update_action()
        t_plan += 1.0 / ctrl.controller_cfg.control_freq
    ctrl.kernel_events.clear()
    ctrl.exchange_events.clear()
    ctrl.noise_events.clear()
    ctrl.reserve_timing_events(7 * args.steps * max(1, ctrl.max_opt_iters) + 8)  # (the timed region records HIP events, it does not create them)
    ctrl.solver_warnings = False
    if not is_policy:
        ctrl.solver_stats()  # zero the kernels' counters: the line reports the timed steps alone
    traces_in_step = bool(not is_policy and ctrl.trace_sensors and not args.traces_outside_step)
    torch.cuda.synchronize()
    barrier()
    t0 = time.perf_counter()
    per_step = []
    # closed-form models: a plan step is ~0.12 ms and the three HIP events of the kernel split cost ~4 us of it: they are recorded on every fourth timed step (kernel_ms =
    # their mean); every step of the articulated models carries them
    ev_every = 4 if (not is_policy and ctrl.model is not None and getattr(ctrl.model, "closed_form", False) and world == 1) else 1
    for i_step in range(args.steps):
        ts = time.perf_counter()
        ctrl.time = t_plan
        ctrl.record_kernel_events = (i_step % ev_every == 0)
        ctrl.update_action()
        if traces_in_step:
            _ = ctrl.traces  # the reference's update_action ends with update_traces (judo/controller/controller.py:299): the timed step does too
        t_plan += 1.0 / ctrl.controller_cfg.control_freq
        per_step.append(time.perf_counter() - ts)
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    ctrl.record_kernel_events = True
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if SHARED_GPU_TEST else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in ctrl.kernel_events])) if ctrl.kernel_events else float("nan")
    # The timed steps start from rest, the cheap end of the closed loop.  The reference's own statistic (judo/app/benchmark.py:19,96-107) is 100 timed plan steps after 10
    # warm-ups, summarised as mean / std / median / IQR / min / max: when --steps < 100 that loop is run here as well -- restarted from the same state and seed, outside
    # `value` -- so that the tail of the distribution is in the driver's line and not only under profiles/.  `steady_state` = its last 20 plan steps.
    steady, bench100 = None, None
    if world == 1 and not is_policy and not args.no_steady_state and args.steps < 100:
        n_warm, n_timed, n_last = 10, 100, 20
        n_ev = len(ctrl.kernel_events)
        ctrl.reserve_timing_events(4 * (n_warm + n_timed) * max(1, ctrl.max_opt_iters) + 8)
        ctrl.reset()
        ctrl.current_state = ctrl.task.default_state()
        ctrl.optimizer.seed(args.seed)
        tq, ms100 = 0.0, []
        for i in range(n_warm + n_timed):
            ts = time.perf_counter()
            ctrl.time = tq
            ctrl.update_action()
            if traces_in_step:
                _ = ctrl.traces
            tq += 1.0 / ctrl.controller_cfg.control_freq
            if i >= n_warm:
                ms100.append((time.perf_counter() - ts) * 1e3)
        torch.cuda.synchronize()
        m100 = np.array(ms100)
        k100 = np.array([a.elapsed_time(b) for a, b in ctrl.kernel_events[n_ev:]])[-n_timed * max(1, ctrl.max_opt_iters):]
        bench100 = {"plan_steps": n_timed, "warmup": n_warm, "mean": float(m100.mean()), "std": float(m100.std()), "median": float(np.median(m100)),
                    "iqr": float(np.percentile(m100, 75) - np.percentile(m100, 25)), "min": float(m100.min()), "max": float(m100.max()), "kernel_ms_mean": float(k100.mean()),
                    "budget_ms": 1e3 / ctrl.controller_cfg.control_freq, "steps_over_budget": int((m100 > 1e3 / ctrl.controller_cfg.control_freq).sum()),
                    "note": "the reference's benchmark statistic (judo/app/benchmark.py:96-107: 100 plan steps after 10 warm-ups), same state and seed as the timed steps, "
                            "traces read inside; budget_ms = 1 / control_freq (judo/controller/controller.py:39)"}
        steady = {"plan_steps": [n_warm + n_timed - n_last, n_warm + n_timed], "ms_per_step": float(m100[-n_last:].mean()), "kernel_ms": float(k100[-n_last * max(1, ctrl.max_opt_iters):].mean()),
                  "note": "the last 20 plan steps of the benchmark_100 loop (the plan has left the rest pose: more contacts per step)"}
        del ctrl.kernel_events[n_ev:]
        del ctrl.exchange_events[n_ev:]
    exch_ms = float(np.mean([a.elapsed_time(b) for a, b in ctrl.exchange_events])) if ctrl.exchange_events else 0.0
    # where a plan step goes on this rank: the rollout kernel, the exchange (update records: block partials, all-gather over the ranks, merge kernel) and the rest
    # (host: time shift, packing, launches, the one wait for the new nominal).  With several GPUs every rank reports its own split.
    split = {"rank": rank, "rollouts": int(ctrl.last_shard.count), "kernel_ms": kern_ms, "exchange_ms": exch_ms, "plan_step_ms": float(np.mean(per_step) * 1e3),
             "host_and_launch_ms": float(np.mean(per_step) * 1e3 - kern_ms - exch_ms), "kernel_events_every_n_steps": ev_every}
    if (world > 1 or rccl_one) and getattr(ctrl, "noise_events", None) and len(ctrl.noise_events) >= len(ctrl.exchange_events) > 0:
        # the next iteration's noise draw (side stream) on the exchange's clock: both measured from the event behind the rollout + record launch
        nev = ctrl.noise_events[-len(ctrl.exchange_events):]
        n0 = float(np.mean([x[0].elapsed_time(n[0]) for x, n in zip(ctrl.exchange_events, nev)]))
        n1 = float(np.mean([x[0].elapsed_time(n[1]) for x, n in zip(ctrl.exchange_events, nev)]))
        inside = float(np.mean([x[0].elapsed_time(n[1]) <= x[0].elapsed_time(x[1]) for x, n in zip(ctrl.exchange_events, nev)]))
        split["noise_draw_on_side_stream"] = {"start_ms": n0, "end_ms": n1, "exchange_end_ms": exch_ms, "finished_inside_exchange": inside,
                                              "note": "ms relative to the event behind this rank's rollout + record launch (negative: while that launch still runs): the next iteration's noise is drawn on a second stream, enqueued in front of the all-gather -- it runs beside the rollout kernel and the exchange, never between the record and the collective"}
    per_rank = [split]
    if world > 1:
        per_rank = [None] * world
        dist.all_gather_object(per_rank, split)
    solver = None
    if not is_policy and ctrl.model is not None and args.task not in ("cartpole", "cylinder_push"):
        st = ctrl.solver_stats()
        if st["steps"] > 0:  # (32-bit counters: fine for the default run lengths, they wrap after ~4e9 rollout-iterations)
            solver = {"newton_iters_per_step": st["newton_iters"] / st["steps"], "newton_cap_hits_per_step": st["newton_cap_hits"] / st["steps"],
                      "contacts_dropped_per_step": st["contact_overflow"] / st["steps"]}
            if st.get("wave_steps"):
                solver["wave_newton_iters_per_step"] = st["wave_newton_iters"] / st["wave_steps"]
                solver["lock_step_inflation"] = solver["wave_newton_iters_per_step"] / max(solver["newton_iters_per_step"], 1e-9)
    # the reference's plan time includes update_traces (judo/controller/controller.py:299).  Since round 3 the fused kernels write every rollout's trace sensors and the
    # elites' rows are gathered on the device, so reading `Controller.traces` is part of the timed steps above; what it costs on its own is measured here, on further steps
    with_traces = None
    if not is_policy and ctrl.trace_sensors and not args.no_with_traces:
        # (the plan keeps moving, and with it the cost of a plan step: the extra steps are not compared with the timed ones as wholes -- what is measured is the
        # time reading `Controller.traces` adds to each of them, which is then put on top of the timed steps' mean)
        n_extra, tq, t_tr = min(args.steps, 10), t_plan, 0.0
        for _ in range(n_extra):
            ctrl.time = tq
            ctrl.update_action()
            torch.cuda.synchronize()
            barrier()
            tw = time.perf_counter()
            _ = ctrl.traces
            torch.cuda.synchronize()
            t_tr += time.perf_counter() - tw
            tq += 1.0 / ctrl.controller_cfg.control_freq
        base = elapsed / args.steps * 1e3
        with_traces = {"ms_per_step": base if traces_in_step else base + t_tr / n_extra * 1e3, "traces_ms": t_tr / n_extra * 1e3, "steps": n_extra, "max_num_traces": int(ctrl.max_num_traces),
                       "in_value": traces_in_step,
                       "note": "traces_ms = mean time of reading Controller.traces after a plan step (fetch of the elites' trace rows + polyline packing), measured on further plan "
                               "steps; in_value: the timed steps read the traces themselves, as the reference's update_action does"}
    # leap_cube: the same measurement restarted with the hand's own contacts switched off (the model round 1 measured), outside the timed region
    cube_only = None
    if args.task == "leap_cube" and world == 1 and ctrl.model is not None and ctrl.model.self_collision and not args.no_cube_only:
        ctrl.model.set_self_collision(False)
        ctrl.reset()
        ctrl.current_state = ctrl.task.default_state()
        ctrl.optimizer.seed(args.seed)
        n_extra, tq = min(args.steps, 10), 0.0
        for i in range(args.warmup + n_extra):
            if i == args.warmup:
                ctrl.kernel_events.clear()
                torch.cuda.synchronize()
                tc = time.perf_counter()
            ctrl.time = tq
            ctrl.update_action()
            tq += 1.0 / ctrl.controller_cfg.control_freq
        torch.cuda.synchronize()
        tc = (time.perf_counter() - tc) / n_extra
        cube_only = {"ms_per_step": tc * 1e3, "rollouts_per_s": N / tc, "steps": n_extra, "warmup": args.warmup,
                     "kernel_ms": float(np.mean([a.elapsed_time(b) for a, b in ctrl.kernel_events])),
                     "note": "hand self-collision off (the cube's contacts only): the model of round 1's 81.8 ms line, restarted from the same state and seed"}
        ctrl.model.set_self_collision(True)
    # The timed steps above are a closed loop, chaotic in the last bits (+-9 % over seeds, reshuffled by every build).  The figure that IS comparable from build to build and
    # from round to round: the plan inputs of 40 consecutive plan steps recorded once in round 2 (tools/diag/ab_fixed_inputs.py record), replayed with fixed per-step seeds.
    replay = None
    rfile = os.path.join(ROOT, "tools", "diag", {"leap_cube": "ab_inputs_leap.npz", "fr3_pick": "ab_inputs_fr3.npz"}.get(args.task, "-"))
    if world == 1 and not is_policy and not args.no_replay and os.path.exists(rfile) and (N, H) == WORKLOADS[args.task][1:] and args.optimizer is None:
        rec = np.load(rfile)
        ctrl.reset()
        ctrl.current_state = ctrl.task.default_state()
        ctrl.kernel_events.clear()
        tr0 = None
        for i in range(rec["knots"].shape[0]):
            if i == 2:
                torch.cuda.synchronize()
                tr0 = time.perf_counter()
            ctrl.optimizer.seed(1000 + i)
            ctrl.nominal_knots = rec["knots"][i].copy()
            ctrl.times = rec["times"][i].copy()
            ctrl.update_spline(ctrl.times, ctrl.nominal_knots)
            ctrl.time = float(rec["t"][i])
            if opt_name == "cem":
                ctrl.optimizer.sigma = rec["sigma"][i].copy()
            ctrl.update_action()
        torch.cuda.synchronize()
        nrep = rec["knots"].shape[0]
        kk = np.array([a.elapsed_time(b) for a, b in ctrl.kernel_events])
        replay = {"kernel_ms": float(kk.mean()), "kernel_ms_first10": float(kk[:10].mean()), "kernel_ms_last10": float(kk[-10:].mean()), "plan_step_ms": (time.perf_counter() - tr0) / (nrep - 2) * 1e3,
                  "plan_steps": int(nrep), "inputs": os.path.relpath(rfile, ROOT),
                  "note": "recorded plan inputs of 40 consecutive plan steps replayed with fixed seeds: deterministic workload, the figure to compare builds and rounds by "
                          "(round 3's kernels on round 4's boxes: leap_cube 80.65 ms, fr3_pick 11.13 ms; round 2's: 80.9 / 19.2)"}
    n_local = ctrl.last_shard.count
    substeps = ctrl.task.physics_substeps
    if is_policy:  # per control step the tree kernel reads state + control + warm start and writes state + warm start; H launches inside the timed region
        alg_bytes = (2 * 51 + 19 + 2 * 25) * 4 * n_local
        achieved = alg_bytes * H / (kern_ms * 1e-3) / 1e9
    else:
        alg_bytes = (4 * K * nu + 4) * n_local
        achieved = alg_bytes / (kern_ms * 1e-3) / 1e9

    # HBM traffic of the dominant kernel: PMC counters cannot be read from inside this process; the committed rocprofv3
    # measurement of this exact launch (profiles/, separate --pmc passes) is reported when the workload matches.
    traffic, traffic_src, issue = None, None, None
    tfile = os.path.join(ROOT, "profiles", TRAFFIC_FILE)
    self_on = bool(ctrl.model is not None and ctrl.model.self_collision)
    if world == 1 and (N, H) == WORKLOADS[args.task][1:] and os.path.exists(tfile):
        t = json.load(open(tfile)).get(args.task if self_on or args.task != "leap_cube" else "leap_cube_cube_only")
        if t:
            traffic, traffic_src = t["hbm_bytes_per_launch"], f"profiles/{TRAFFIC_FILE} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)"
            issue = t.get("issue")  # the bound that does apply: VALU issue slots of the same profiled launches (SQ_INSTS_VALU against SQ_BUSY_CYCLES)

    if rank == 0:
        ms = np.array(per_step) * 1e3
        if os.environ.get("JUDO_BENCH_STEPS"):
            print("plan steps [ms]: " + " ".join(f"{x:.2f}" for x in ms), file=sys.stderr)
        line = {
            "metric": "rollouts/sec + plan-step ms, leap_cube MPPI 65536xH64" if args.task == "leap_cube" else f"rollouts/sec + plan-step ms, {args.task}",
            "value": N * args.steps / elapsed,
            "unit": "rollouts/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"{args.task} {opt_name.upper()} {N} rollouts x H={H} (K={K}, nu={nu}, spline {ctrl.spline_order}, dt={ctrl.task.dt})",
                       "rollouts": N, "horizon_steps": H, "num_nodes": K, "parallelism": f"rollout-shard x{world}", "max_opt_iters": ctrl.max_opt_iters,
                       "noise_seed": args.seed,
                       "closed_loop": "every plan step starts from the previous plan: ms_per_step depends on where the noise stream leads it, chaotically (leap_cube, "
                                      "round-6 build: eight seeds 46.3-59.5 ms, mean 51.4 over 20 steps -- the default seed 1234: 53.8 --; four seeds 51.8-55.7, mean 53.2 over 100 steps; "
                                      "profiles/r06_seed_sweep.txt; round 4: 57.6-72.4, mean 63.2; 63.4-68.8, mean 66.4; round 3: 74.0-87.8, mean 81.9; 85.8-96.8, mean 89.4)",
                       "hand_self_collision": self_on if args.task.startswith("leap") else None,
                       "traces": ("read inside every timed plan step (update_traces is part of the reference's update_action): the fused kernel writes every rollout's trace sensors, "
                                  "the elites' rows are gathered on the device" if traces_in_step else "not read inside the timed steps; see plan_step_ms_with_traces")},
            "plan_step_ms": {"mean": float(ms.mean()), "std": float(ms.std()), "median": float(np.median(ms)), "iqr": float(np.percentile(ms, 75) - np.percentile(ms, 25)),
                             "min": float(ms.min()), "max": float(ms.max())},
            "physics_steps_per_s": N * H * substeps * args.steps / elapsed,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": ("policy rollout: per control step k_tree_v4 (2 substeps) + the policy step kernels; kernel_ms covers all H control steps and the reward"
                                    if is_policy else "fused rollout+cost"), "kernel_ms": kern_ms, "algorithmic_bytes_per_launch": alg_bytes,
                         "traffic_source": traffic_src,
                         "note": "latency/VALU-issue-bound by construction (H serial physics steps, ~1e5 flop per step against a few hundred algorithmic bytes per rollout); see DESIGN.md section 6"},
        }
        if issue:
            line["roofline"]["issue"] = dict(issue, source=f"profiles/{TRAFFIC_FILE}: the committed profile of this workload, not this run")
        line["per_rank"] = per_rank
        if with_traces:
            line["plan_step_ms_with_traces"] = with_traces
        if solver:
            line["solver"] = solver
        if cube_only:
            line["cube_only"] = cube_only
        if bench100:
            line["benchmark_100"] = bench100
        if steady:
            line["steady_state"] = steady
        if replay:
            line["recorded_inputs"] = replay
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline_policy(ctrl) if is_policy else cpu_baseline(args.task, ctrl)
        print(json.dumps(line))
    if world > 1 or rccl_one:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

"""Sharding of rollouts over the GPUs of one node and the single per-plan-step exchange.

The reference has no collective (one process, one thread per rollout).  Rollouts are independent given
(x0, nominal knots, model), so GPU g owns the contiguous slice [offset, offset+count) of the N global rollouts
(global sample 0, the unperturbed nominal, lives on rank 0) and the ranks exchange ONE small record per optimiser
iteration: MPPI `(beta_g, S_g, V_g[K*nu])`, CEM/PS `k x (cost, index, knots[K*nu])`.  One all-gather (RCCL over xGMI on
the GPU box, gloo in the CPU tests), then every rank runs the same deterministic merge kernel, so all ranks hold the
identical new nominal without a broadcast.  Messages are <= a few KB: latency-bound, never per-link-bandwidth-bound.
"""

from __future__ import annotations

from dataclasses import dataclass

import torch
import torch.distributed as dist


@dataclass(frozen=True)
class Shard:
    world: int
    rank: int
    total: int
    count: int
    offset: int


def shard_rollouts(total: int, world: int, rank: int) -> Shard:
    """Contiguous split; the first `total % world` ranks take one extra rollout."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world of size {world}")
    if total < world:
        raise ValueError(f"cannot shard {total} rollouts over {world} ranks")
    base, rem = divmod(total, world)
    count = base + (1 if rank < rem else 0)
    offset = rank * base + min(rank, rem)
    return Shard(world, rank, total, count, offset)


def world_info(group=None) -> tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(group), dist.get_rank(group)
    return 1, 0


def all_gather_records(rec: torch.Tensor, group=None, force: bool = False) -> torch.Tensor:
    """(L,) record per rank -> (world*L,) on every rank, rank-major.  `force`: run the collective on a process group of ONE rank as well (tests/test_gpu_nccl.py: the
    RCCL branch below on a one-GPU box)."""
    world, _ = world_info(group)
    if world == 1 and not (force and dist.is_available() and dist.is_initialized()):
        return rec
    if rec.is_cuda and dist.get_backend(group) == "gloo":
        # gloo has no device all-gather: stage the (<= few KB) record through the host.  This is the rendezvous used when several
        # ranks share one GPU (tests); on a multi-GPU node the backend is nccl (RCCL) and the branch below runs on the device.
        host = torch.empty(world * rec.numel(), dtype=rec.dtype)
        dist.all_gather_into_tensor(host, rec.detach().cpu().contiguous(), group=group)
        return host.to(rec.device)
    out = torch.empty(world * rec.numel(), dtype=rec.dtype, device=rec.device)
    dist.all_gather_into_tensor(out, rec.contiguous(), group=group)
    return out


def all_gather_costs(costs: torch.Tensor, shard: Shard, group=None) -> torch.Tensor:
    """Optional: the full (N,) cost vector on every rank (N/G floats per rank, e.g. 32 KB at 8 GPUs)."""
    if shard.world == 1:
        return costs
    base = shard.total // shard.world
    pad = base + (1 if shard.total % shard.world else 0)
    buf = torch.full((pad,), float("inf"), dtype=costs.dtype, device=costs.device)
    buf[: shard.count] = costs
    out = torch.empty(shard.world * pad, dtype=costs.dtype, device=costs.device)
    dist.all_gather_into_tensor(out, buf, group=group)
    parts = [out[r * pad : r * pad + shard_rollouts(shard.total, shard.world, r).count] for r in range(shard.world)]
    return torch.cat(parts)

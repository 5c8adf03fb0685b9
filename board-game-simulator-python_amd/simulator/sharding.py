"""Sharding a batch of independent games over ranks (one process per GPU) and gathering the rewards.

Boards never interact, so the path partitions without any data-path exchange: rank r of R owns the contiguous
global game ids [r * per_rank, (r + 1) * per_rank).  RNG streams are keyed by GLOBAL game id
(``Batch.set_first_game``), so the union of the shards is bit-identical to the unsharded run for any R.
The single collective is the reward gather: every rank contributes int8[per_rank, 2]; torch.distributed's
"nccl" backend is RCCL on ROCm (xGMI between the GPUs of a node), "gloo" runs the same code on CPUs in tests.
"""

from __future__ import annotations

from typing import Tuple


def shard_range(total_games: int, rank: int, world: int) -> Tuple[int, int]:
    """(first global game id, number of games) of `rank`; every rank gets the same count (weak scaling)."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError(f"bad rank {rank} of {world}")
    if total_games % world:
        raise ValueError(f"{total_games} games do not split evenly over {world} ranks")
    per_rank = total_games // world
    return rank * per_rank, per_rank


def gather_rewards(dist, local_reward, out=None):
    """All-gather int8[per_rank, 2] rewards into int8[world * per_rank, 2], ordered by global game id.

    `dist` is torch.distributed (initialised); works for CUDA tensors over nccl/RCCL and CPU tensors over gloo."""
    import torch

    world = dist.get_world_size()
    per_rank = local_reward.shape[0]
    if out is None:
        out = torch.empty((world * per_rank, 2), dtype=local_reward.dtype, device=local_reward.device)
    if out.shape != (world * per_rank, 2):
        raise ValueError("gather buffer has the wrong shape")
    dist.all_gather_into_tensor(out, local_reward.contiguous())
    return out


def gather_outcomes(dist, local_packed, out=None, async_op: bool = False):
    """All-gather the ranks' packed 2-bit outcome codes (uint8[per_rank / 4] each; per_rank must be a multiple of 4)
    into uint8[world * per_rank / 4], ordered by global game id.  0.25 B per game crosses xGMI instead of the 2 B of
    an int8[.., 2] reward pair; `simulator.batch.expand_outcomes` turns the result into rewards on the device.

    async_op=True returns (out, work): the collective then runs beside whatever the caller enqueues next, and
    `work.wait()` (which only makes the CURRENT stream wait) is due before `out` is read."""
    import torch

    world = dist.get_world_size()
    if out is None:
        out = torch.empty(world * local_packed.numel(), dtype=local_packed.dtype, device=local_packed.device)
    if out.numel() != world * local_packed.numel():
        raise ValueError("gather buffer has the wrong size")
    work = dist.all_gather_into_tensor(out, local_packed.contiguous(), async_op=async_op)
    return (out, work) if async_op else out


def sum_steps(dist, local_steps: int, device) -> int:
    """Total env-steps over all ranks (one int64 all-reduce, outside any timed region)."""
    import torch

    t = torch.tensor([int(local_steps)], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return int(t.item())

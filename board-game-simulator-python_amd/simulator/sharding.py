"""Sharding a batch of independent games over ranks (one process per GPU) and gathering the rewards.

Boards never interact, so the path partitions without any data-path exchange: rank r of R owns the contiguous
global game ids [r * per_rank, (r + 1) * per_rank).  RNG streams are keyed by GLOBAL game id
(``Batch.set_first_game``), so the union of the shards is bit-identical to the unsharded run for any R.
The single collective is the reward gather to the rank that owns the host array: ranks contribute their 2-bit outcome
codes (`gather_outcomes_to`; `gather_rewards` ships int8 pairs to every rank instead); torch.distributed's "nccl"
backend is RCCL on ROCm (xGMI between the GPUs of a node), "gloo" runs the same code on CPUs in tests.
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


def gather_outcomes_to(dist, local_packed, out=None, dst: int = 0, async_op: bool = False):
    """Gather the ranks' packed 2-bit outcome codes (uint8[per_rank / 4] each; per_rank a multiple of 4) on rank `dst`:
    `out` = uint8[world * per_rank / 4] there (ordered by global game id), None elsewhere.  0.25 B per game crosses
    xGMI instead of the 2 B of an int8 reward pair, and only the rank that owns the host array receives anything
    (a gather, not an all-gather); `RewardSink.submit_packed` then takes the codes to the host and expands them.

    async_op=True returns the work handle: the collective then runs beside whatever the caller enqueues next, and
    `work.wait()` (which only makes the CURRENT stream wait) is due before `out` is read or `local_packed` reused."""
    world, rank = dist.get_world_size(), dist.get_rank()
    local = local_packed.contiguous()
    if rank == dst:
        if out is None or out.numel() != world * local.numel() or out.dtype != local.dtype:
            raise ValueError("rank dst needs a gather buffer of world * len(local_packed) elements")
        chunks = list(out.view(world, local.numel()).unbind(0))  # views: rank r's codes land in place
        work = dist.gather(local, gather_list=chunks, dst=dst, async_op=async_op)
    else:
        work = dist.gather(local, dst=dst, async_op=async_op)
    return work if async_op else out


def sum_steps(dist, local_steps: int, device) -> int:
    """Total env-steps over all ranks (one int64 all-reduce, outside any timed region)."""
    import torch

    t = torch.tensor([int(local_steps)], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return int(t.item())


def multi_device_rollout(devices, height: int, width: int, count: int, n_per_device: int, seed: int):
    """Connect rollouts sharded over several GPUs of this node from ONE process, without torch.distributed:
    `bgs_multi_connect_rollout` (RCCL point-to-point gather of the outcome codes to devices[0], one copy to the host).
    Returns (reward int8[len(devices) * n_per_device, 2] in global game order, env-steps)."""
    import ctypes

    import numpy as np

    from .game import _abi

    devs = (ctypes.c_int * len(devices))(*[int(d) for d in devices])
    reward = np.empty((len(devices) * n_per_device, 2), dtype=np.int8)
    steps = ctypes.c_uint64(0)
    _abi.check(
        _abi.lib().bgs_multi_connect_rollout(
            devs, len(devices), height, width, count, n_per_device, ctypes.c_uint64(seed), ctypes.c_void_p(reward.ctypes.data),
            ctypes.byref(steps),
        )
    )
    return reward, steps.value

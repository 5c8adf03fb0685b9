"""Sharding a batch of independent games over ranks (one process per GPU) and gathering the rewards.

Boards never interact, so the path partitions without any data-path exchange: rank r of R owns the contiguous
global game ids [r * per_rank, (r + 1) * per_rank).  RNG streams are keyed by GLOBAL game id
(``Batch.set_first_game``), so the union of the shards is bit-identical to the unsharded run for any R.
The single exchange is the hand-over of the rewards to the ONE host array, and there are two ways to do it:

* `SharedRewardRing` (ranks on one node; bench.py's default for N > 1): the host array lives in shared memory mapped by
  every rank, and each rank's own reward sink delivers its games into its rows -- every GPU uses its OWN PCIe link
  (0.25 B per game of outcome codes) and its own host threads for the expansion, nothing funnels through rank 0's GPU;
* `RewardGather` (any topology; `bench.py --gather rccl`): the same gather inside libbgs.so (`bgs_gather_*`) -- persistent
  communicator, its own stream and thread, one library call per step on the launching thread;
* `gather_outcomes_to` (torch collectives; the gloo rehearsals use it): ranks contribute their 2-bit outcome codes to rank 0's GPU over RCCL (xGMI inside
  a node), whose sink copies them to the host and expands all of them -- at 8 GPUs that is 2 MiB per step over one PCIe
  link and 16 MiB of host writes per step by one process (`gather_rewards` ships int8 pairs to every rank instead).
torch.distributed's "nccl" backend is RCCL on ROCm, "gloo" runs the same code on CPUs in tests.
"""

from __future__ import annotations

import ctypes
import mmap
import os
import uuid
from typing import Tuple

from .game import _abi


def shard_range(total_games: int, rank: int, world: int) -> Tuple[int, int]:
    """(first global game id, number of games) of `rank`; every rank gets the same count (weak scaling)."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError(f"bad rank {rank} of {world}")
    if total_games % world:
        raise ValueError(f"{total_games} games do not split evenly over {world} ranks")
    per_rank = total_games // world
    return rank * per_rank, per_rank


class SharedRewardRing:
    """`slots` host arrays int8[world * per_rank, 2] in ONE shared-memory segment mapped by every rank of a node, plus a
    progress word per rank and one for the consumer.

    Rank r delivers step s into `mine(s % slots)` (rows [r * per_rank, (r + 1) * per_rank) of `array(s % slots)`, e.g.
    as the destination of `RewardSink.rollout`) and announces it -- `publish(s)`, or `attach(sink)` once and the sink's
    worker threads do it themselves the moment a step is delivered.  The consumer -- any ONE rank, normally rank 0 --
    calls `wait_all(s)`, reads `array(s % slots)` (every rank's rewards of step s, in global game order) and hands the
    slot back with `release(s)`; a producer calls `acquire(s)` before it lets step s overwrite the slot of step
    s - slots (immediate unless the consumer is more than `slots` steps behind).  All waits sleep on a futex
    (`bgs_progress_wait`), nothing polls; progress only ever moves forward.
    The segment is a file in /dev/shm that rank 0 creates and unlinks as soon as everybody has mapped it, so nothing
    is left behind however the processes end.  `dist` (torch.distributed, initialised) only carries the name and one
    barrier at construction; the data path has no collective."""

    def __init__(self, dist, per_rank: int, slots: int, directory=None):
        """`directory`: where rank 0 creates the segment's file -- a path, or a list of candidates tried in order
        (default: /dev/shm, then the temporary directory: a container's /dev/shm may be a few dozen MiB, and a file
        mapping in /tmp is shared between the ranks just the same, only with the page cache behind it)."""
        import tempfile

        import numpy as np

        if directory is None:
            directory = ["/dev/shm", tempfile.gettempdir()]
        directories = [directory] if isinstance(directory, str) else list(directory)

        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.per_rank, self.slots = int(per_rank), int(slots)
        if self.per_rank < 1 or self.slots < 1:
            raise ValueError("per_rank and slots must be positive")
        self._slot_bytes = self.world * self.per_rank * 2
        data = -(-self.slots * self._slot_bytes // 4096) * 4096
        total = data + 64 * (self.world + 1)  # one cache line of progress per rank, one for the consumer
        # Rank 0 creates the file -- if the directory has room for it: a tmpfs accepts the ftruncate and kills the process
        # that touches the first page it cannot back -- and everybody learns the name, or that there is none.  Whether
        # all ranks mapped it is agreed on before anybody goes on: the constructor raises on every rank or on none, so
        # callers can fall back to the collective gather consistently.
        name, fd, error = [None], -1, None
        if self.rank == 0:
            problems = []
            for where in directories:
                try:
                    st = os.statvfs(where)
                    if st.f_bavail * st.f_frsize < total + (64 << 20):
                        raise OSError(f"{where} has {st.f_bavail * st.f_frsize >> 20} MiB free, the reward ring needs {total >> 20}")
                    path = os.path.join(where, f"bgs_rewards_{os.getpid()}_{uuid.uuid4().hex}")
                    fd = os.open(path, os.O_CREAT | os.O_EXCL | os.O_RDWR, 0o600)
                    name[0] = path
                    os.ftruncate(fd, total)
                    break
                except OSError as exc:
                    problems.append(str(exc))
                    if fd >= 0:
                        os.close(fd)
                        fd = -1
                    if name[0] is not None:
                        try:
                            os.unlink(name[0])
                        except OSError:
                            pass
                        name[0] = None
            if name[0] is None:
                error = "rank 0: " + "; ".join(problems)
        dist.broadcast_object_list(name, src=0)
        self.path = name[0]  # (unlinked below: kept for diagnostics only)
        self._map = None
        if name[0] is not None:
            try:
                if self.rank != 0:
                    fd = os.open(name[0], os.O_RDWR)
                self._map = mmap.mmap(fd, total, mmap.MAP_SHARED, mmap.PROT_READ | mmap.PROT_WRITE)
            except (OSError, ValueError) as exc:
                error = f"rank {self.rank}: {exc}"
        if fd >= 0:
            os.close(fd)
        errors = [None] * self.world
        dist.all_gather_object(errors, error)  # (also the barrier after which the name can go)
        if self.rank == 0 and name[0] is not None:
            os.unlink(name[0])
        if any(errors):
            if self._map is not None:
                self._map.close()
            raise RuntimeError("shared reward ring unavailable: " + "; ".join(e for e in errors if e))
        buf = np.frombuffer(self._map, dtype=np.int8, count=self.slots * self._slot_bytes)
        self._arrays = buf.reshape(self.slots, self.world * self.per_rank, 2)
        # (a fresh tmpfs file reads as zeros: every counter starts at 0 without anybody writing it)
        self._progress = np.frombuffer(self._map, dtype=np.int64, count=8 * (self.world + 1), offset=data).reshape(self.world + 1, 8)
        self._words = self._progress.ctypes.data  # rank r's word at + 64 r, the consumer's at + 64 world
        self._sink = None
        self._barriers = 0
        # first touch by the rank that will write the rows: the pages land on that rank's NUMA node
        for k in range(self.slots):
            self.mine(k)[...] = 0x55
        dist.barrier()

    def array(self, slot: int):
        """int8[world * per_rank, 2]: the host array of `slot` (every rank's rows)."""
        return self._arrays[slot]

    def mine(self, slot: int):
        """int8[per_rank, 2]: this rank's rows of `array(slot)` (C-contiguous: a valid sink destination)."""
        return self._arrays[slot, self.rank * self.per_rank : (self.rank + 1) * self.per_rank]

    # ---- producers ---------------------------------------------------------------------------------
    def publish(self, step: int) -> None:
        """This rank's rewards of every step <= `step` are in their host arrays (never moves backwards)."""
        _abi.check(_abi.lib().bgs_progress_store(ctypes.c_void_p(self._words + 64 * self.rank), int(step) + 1))

    def attach(self, sink) -> None:
        """Let `sink` (a RewardSink whose submissions are exactly this rank's steps 0, 1, 2, ... in order) publish by
        itself: its worker threads raise this rank's progress word the moment a step is in its host array."""
        sink.set_progress(self._words + 64 * self.rank, keepalive=self)
        self._sink = sink

    def acquire(self, step: int, timeout: float = 60.0) -> None:
        """Before step `step` may overwrite its slot: the consumer has released step `step - slots`."""
        if step >= self.slots:
            self._wait(self._words + 64 * self.world, 1, step - self.slots + 1, timeout, f"the consumer has not released step {step - self.slots}")

    # ---- the consumer ------------------------------------------------------------------------------
    def done(self, step: int) -> bool:
        return bool((self._progress[: self.world, 0] > step).all())

    def wait_all(self, step: int, timeout: float = 60.0) -> None:
        """Sleep until every rank has delivered `step`."""
        self._wait(self._words, self.world, step + 1, timeout, f"step {step} not delivered")

    def release(self, step: int) -> None:
        """The consumer is done with every step <= `step`: their slots may be overwritten."""
        _abi.check(_abi.lib().bgs_progress_store(ctypes.c_void_p(self._words + 64 * self.world), int(step) + 1))

    def barrier(self, spin_us: int = 2000, timeout: float = 60.0) -> None:
        """Every rank of the ring has arrived (word 1 of each rank's progress line counts its barriers): ranks that run
        in step meet within microseconds -- no collective, no GPU work (bgs_progress_barrier)."""
        self._barriers += 1
        rc = _abi.lib().bgs_progress_barrier(ctypes.c_void_p(self._words + 8), self.world, 8, self.rank, self._barriers,
                                             int(spin_us), int(timeout * 1000))
        if rc != 0:
            behind = [int(r) for r in (self._progress[: self.world, 1] < self._barriers).nonzero()[0]]
            raise TimeoutError(f"barrier {self._barriers}: ranks {behind} did not arrive within {timeout} s")

    def _wait(self, address: int, count: int, target: int, timeout: float, what: str) -> None:
        laggard = ctypes.c_int64(-1)
        rc = _abi.lib().bgs_progress_wait(ctypes.c_void_p(address), count, 8, int(target), int(timeout * 1000), ctypes.byref(laggard))
        if rc != 0:
            behind = [int(r) for r in (self._progress[: self.world, 0] < target).nonzero()[0]] if count > 1 else []
            raise TimeoutError(f"{what} after {timeout} s" + (f" by ranks {behind}" if behind else ""))

    def close(self) -> None:
        if self._sink is not None:
            try:
                self._sink.set_progress(None)  # the word is about to be unmapped
            except Exception:
                pass
            self._sink = None
        self._arrays = self._progress = None
        try:
            self._map.close()
        except BufferError:  # a caller still holds a view: the mapping goes with the process
            pass


class RewardGather:
    """The north-star's collective as a library object: every rank's outcome codes to rank 0 over RCCL (xGMI inside a
    node), rank 0's sink expands them into ONE host array int8[world * per_rank, 2] in global game order
    (`bgs_gather_*`: persistent communicator, communication stream and thread inside libbgs.so; the launching thread
    makes one call per step and never enters RCCL).  `dist` (torch.distributed, initialised, any backend) only carries
    the communicator's 128-byte id from rank 0 to the others.

    Every rank makes the same sequence of `rollout` calls.  A step's codes leave in a group with its neighbours: when the
    group is full, when somebody `wait`s for one of its steps, or by themselves a millisecond after the group's first step
    (BGS_GATHER_FLUSH_US) -- so a rank may submit a few steps and then block on something else.  A step that cannot be
    enqueued on one rank raises THERE (and on every later call of that rank's gather); its message still goes out, as
    zeros, so that the peers are not left waiting: rank 0 delivers reward 0 / 0 for that rank's rows of that step."""

    def __init__(self, dist, per_rank: int, slots: int = 6, host_threads: int = 6, device: int = 0):
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.per_rank = int(per_rank)
        ident = [None]
        if self.rank == 0:
            raw = (ctypes.c_uint8 * 128)()
            _abi.check(_abi.lib().bgs_gather_unique_id(raw))
            ident[0] = bytes(raw)
        dist.broadcast_object_list(ident, src=0)
        raw = (ctypes.c_uint8 * 128).from_buffer_copy(ident[0])
        self._handle = _abi.c_handle()
        self._alive = {}
        _abi.check(_abi.lib().bgs_gather_create(int(device), self.rank, self.world, raw, self.per_rank, int(slots),
                                                int(host_threads), ctypes.byref(self._handle)))

    def rollout(self, batch, host_reward, seed: int, max_plies: int = 2**31 - 1, from_initial: bool = True) -> int:
        """Enqueue one step: `batch.rollout(...)`, this rank's codes to rank 0 and -- on rank 0 -- everybody's rewards
        into `host_reward` int8[world * per_rank, 2] (None on the other ranks).  Returns the step's ticket."""
        from .batch import _reward_destination

        ptr = None
        if self.rank == 0:
            ptr = ctypes.c_void_p(_reward_destination(host_reward, self.world * self.per_rank))
        ticket = ctypes.c_int64(-1)
        _abi.check(_abi.lib().bgs_gather_rollout(self._handle, batch._handle, ctypes.c_uint64(seed), ctypes.c_int32(max_plies),
                                                 ctypes.c_uint32(_abi.ROLLOUT_FROM_INITIAL if from_initial else 0), ptr,
                                                 ctypes.byref(ticket)))
        self._alive[ticket.value] = (host_reward, batch)
        return ticket.value

    def info(self) -> dict:
        """How the gather runs: {"direct": receives straight into the sink's device-mapped slots (else device memory + a
        copy kernel), "batch": steps per group of point-to-point calls, "transport_check": "none" (one rank) / "passed" /
        "passed after falling back from direct receives", "transport": the library in use, "ranks" / "rank": what the
        COMMUNICATOR says about itself (ncclCommCount / ncclCommUserRank; None when the transport has no such query)}."""
        direct, batch, check = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
        _abi.check(_abi.lib().bgs_gather_info(self._handle, ctypes.byref(direct), ctypes.byref(batch), ctypes.byref(check)))
        ranks, rank = ctypes.c_int(-1), ctypes.c_int(-1)
        _abi.check(_abi.lib().bgs_gather_comm(self._handle, ctypes.byref(ranks), ctypes.byref(rank)))
        name = _abi.lib().bgs_gather_transport()
        return {"direct": bool(direct.value), "batch": batch.value,
                "transport_check": ("none", "passed", "passed after falling back from direct receives")[check.value],
                "transport": name.decode() if name else "",
                "ranks": ranks.value if ranks.value >= 0 else None, "rank": rank.value if rank.value >= 0 else None}

    def wait(self, ticket: int) -> None:
        """Rank 0: the step's rewards of ALL ranks are in its host array; other ranks: this rank's codes have left."""
        _abi.check(_abi.lib().bgs_gather_wait(self._handle, int(ticket)))
        for t in [t for t in self._alive if t <= ticket]:
            del self._alive[t]

    def close(self) -> None:
        if self._handle:
            _abi.lib().bgs_gather_destroy(self._handle)
            self._handle = _abi.c_handle()
        self._alive.clear()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


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


class MultiDeviceRollout:
    """Connect rollouts sharded over several GPUs of this node from ONE process, without torch.distributed, with the
    batches, streams and RCCL communicators kept between steps (`bgs_multi_create / rollout / destroy`)."""

    def __init__(self, devices, height: int, width: int, count: int, n_per_device: int):
        import ctypes

        self.devices = [int(d) for d in devices]
        self.n_per_device = int(n_per_device)
        devs = (ctypes.c_int * len(self.devices))(*self.devices)
        self._handle = _abi.c_handle()
        _abi.check(_abi.lib().bgs_multi_create(devs, len(self.devices), height, width, count, self.n_per_device, ctypes.byref(self._handle)))

    def rollout(self, seed: int, out=None):
        """One step: returns (reward int8[len(devices) * n_per_device, 2] in global game order, env-steps)."""
        import ctypes

        import numpy as np

        from .batch import _reward_destination

        if not self._handle:
            raise RuntimeError("this MultiDeviceRollout has been closed")
        if out is None:
            out = np.empty((len(self.devices) * self.n_per_device, 2), dtype=np.int8)
        # (dtype, size, contiguity and writeability checked: the library writes devices * n * 2 bytes through this pointer)
        ptr = _reward_destination(out, len(self.devices) * self.n_per_device)
        steps = ctypes.c_uint64(0)
        _abi.check(_abi.lib().bgs_multi_rollout(self._handle, ctypes.c_uint64(seed), ctypes.c_void_p(ptr), ctypes.byref(steps)))
        return out, steps.value

    def close(self) -> None:
        if self._handle:
            _abi.lib().bgs_multi_destroy(self._handle)
            self._handle = _abi.c_handle()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


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

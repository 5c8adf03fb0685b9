"""N > 1 path on CPU: two `gloo` ranks shard a batch by global game id and deliver the rewards to rank 0's host array
with the same host code bench.py runs over RCCL -- `gather_outcomes_to` (2-bit codes, gather to rank 0, synchronous and
async_op + wait) followed by the library's host expansion -- and rank 0 compares with the unsharded run, game order
included.  The boards themselves come from the CPU oracle here (no GPU in this test); on the GPU box
tests/test_gpu_handover.py::test_bench_multi_rank_rehearsal runs bench.py's own N > 1 loops (shared host array and RCCL-style gather), and
tests/test_gpu_parity.py::test_connect_sharding_is_invisible checks the sharding property on the device."""

import os
import socket
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent(
    """
    import os, sys
    sys.path[:0] = [{root!r}, os.path.join({root!r}, "board-game-simulator-python_amd")]
    import numpy as np, torch, torch.distributed as dist
    from oracle import oracle
    from simulator.batch import expand_outcomes_host
    from simulator.sharding import gather_outcomes_to, gather_rewards, shard_range, sum_steps

    def pack_codes(winner):  # what bgs_pack_outcomes does on the device: 2 bits per game, game 4i in the low bits
        status = np.where(winner == -1, 0, np.where(winner == 2, 3, winner + 1)).astype(np.uint8)
        q = status.reshape(-1, 4)
        return (q[:, 0] | (q[:, 1] << 2) | (q[:, 2] << 4) | (q[:, 3] << 6)).astype(np.uint8)

    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    total, seed = 4096, 0x0123456789ABCDEF
    first, count = shard_range(total, rank, world)
    assert (first, count) == (rank * total // world, total // world)
    shard = oracle.ConnectOracle(6, 7, 4, count)
    steps = shard.rollout(seed, first_game=first)
    gathered = gather_rewards(dist, torch.from_numpy(shard.reward))
    # the path bench.py takes: codes -> gather to rank 0 -> host expansion into the one array
    local = torch.from_numpy(pack_codes(shard.winner))
    buf = torch.empty(world * local.numel(), dtype=torch.uint8) if rank == 0 else None
    out = gather_outcomes_to(dist, local, buf, dst=0)
    assert (out is None) == (rank != 0)
    buf2 = torch.zeros(world * local.numel(), dtype=torch.uint8) if rank == 0 else None
    work = gather_outcomes_to(dist, local, buf2, dst=0, async_op=True)
    work.wait()
    all_steps = sum_steps(dist, steps, "cpu")
    if rank == 0:
        whole = oracle.ConnectOracle(6, 7, 4, total)
        assert whole.rollout(seed) == all_steps
        assert np.array_equal(gathered.numpy(), whole.reward)
        assert torch.equal(buf, buf2)
        host = np.full((total, 2), 77, dtype=np.int8)
        expand_outcomes_host(buf2.numpy(), total, host)
        assert np.array_equal(host, whole.reward), "gathered codes are not in global game order"
        assert not np.array_equal(host[:count], host[count:])  # the shards differ, so the order check means something
        try:
            gather_outcomes_to(dist, local, torch.empty(3, dtype=torch.uint8), dst=0)
            raise SystemExit("a wrong gather buffer was accepted")
        except ValueError:
            pass
        print("GLOO_SHARDING_OK", all_steps)
    dist.barrier()
    dist.destroy_process_group()
    """
)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_two_rank_sharding_and_reward_gather(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT))
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS="2")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=240)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\\n".join(outs)
    assert "GLOO_SHARDING_OK" in outs[0]


RING_WORKER = textwrap.dedent(
    """
    import os, sys, time
    sys.path[:0] = [{root!r}, os.path.join({root!r}, "board-game-simulator-python_amd")]
    import numpy as np, torch.distributed as dist
    from oracle import oracle
    from simulator.batch import expand_outcomes_host
    from simulator.sharding import SharedRewardRing, shard_range

    def pack_codes(winner):
        status = np.where(winner == -1, 0, np.where(winner == 2, 3, winner + 1)).astype(np.uint8)
        q = status.reshape(-1, 4)
        return (q[:, 0] | (q[:, 1] << 2) | (q[:, 2] << 4) | (q[:, 3] << 6)).astype(np.uint8)

    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    per_rank, slots, steps, seed = {per_rank}, {slots}, {steps}, 0x0123456789ABCDEF
    ring = SharedRewardRing(dist, per_rank, slots)
    assert ring.array(0).shape == (world * per_rank, 2) and ring.mine(1).shape == (per_rank, 2)
    assert ring.mine(0).flags["C_CONTIGUOUS"] and not ring.done(0)
    first, _ = shard_range(world * per_rank, rank, world)
    # Producers run as far ahead as the consumer's releases allow; the consumer (rank 0, which also produces) trails and
    # is slow on purpose: a producer that overwrote a slot before its release would be caught by the comparison below.
    consumed = 0
    def consume_upto(limit):
        global consumed
        while consumed <= limit:
            ring.wait_all(consumed)
            time.sleep(0.01)
            whole = oracle.ConnectOracle(6, 7, 4, world * per_rank)
            whole.rollout(seed + consumed)
            assert np.array_equal(ring.array(consumed % slots), whole.reward), f"step {{consumed}}: the shared array is not the unsharded run"
            ring.release(consumed)
            consumed += 1
    for step in range(steps):
        if rank == 0:
            consume_upto(step - slots)   # (what acquire() below is about to need from the consumer)
        ring.acquire(step)               # the consumer has released the step that used this slot before
        shard = oracle.ConnectOracle(6, 7, 4, per_rank)
        shard.rollout(seed + step, first_game=first)
        # what a rank's reward sink does: expand its own outcome codes into ITS rows of the shared array
        expand_outcomes_host(pack_codes(shard.winner), per_rank, ring.mine(step % slots))
        ring.publish(step)
    if rank == 0:
        consume_upto(steps - 1)
    ring.publish(steps - 3)              # progress never moves backwards
    assert ring._progress[rank, 0] == steps
    dist.barrier()
    assert ring.done(steps - 1)
    # the ring's own barrier (words of the shared segment, no collective): nobody passes before the last rank arrives,
    # whether the others are still spinning (20 ms) or already asleep (spin_us = 0), barrier after barrier
    for round_, spin in enumerate((2000, 0, 50)):
        time.sleep(0.02 * ((rank + round_) % world))
        arrived = time.monotonic()
        ring.barrier(spin_us=spin)
        passed = time.monotonic()
        stamps = [None] * world
        dist.all_gather_object(stamps, (arrived, passed))
        assert min(p for _, p in stamps) >= max(a for a, _ in stamps) - 1e-4, stamps
        assert max(p for _, p in stamps) - max(a for a, _ in stamps) < 0.5, stamps
    if rank == world - 1:
        time.sleep(0.5)   # a rank that does not come: the others time out and name it
    else:
        try:
            ring.barrier(spin_us=100, timeout=0.2)
            raise SystemExit("the barrier let a rank through before everybody had arrived")
        except TimeoutError as exc:
            assert str(world - 1) in str(exc), exc
    if rank == world - 1:
        ring.barrier()    # (arrives late: everybody else's word is already there)
    dist.barrier()
    # a directory that cannot hold the ring: the constructor raises on EVERY rank (callers then fall back together)
    try:
        SharedRewardRing(dist, 1 << 40, slots)
        raise SystemExit("a ring larger than /dev/shm was accepted")
    except RuntimeError as exc:
        assert "rank 0" in str(exc)
    try:
        SharedRewardRing(dist, per_rank, slots, directory="/nonexistent-directory")
        raise SystemExit("a ring in a missing directory was accepted")
    except RuntimeError:
        pass
    # a list of candidates: the first that works is taken (a container's /dev/shm may be tiny: the temporary directory
    # serves as well), and the file is gone from there too once everybody has mapped it
    import tempfile
    second = SharedRewardRing(dist, per_rank, slots, directory=["/nonexistent-directory", tempfile.gettempdir()])
    assert second.path.startswith(tempfile.gettempdir()) and not os.path.exists(second.path)
    second.mine(0)[...] = rank + 1
    dist.barrier()
    assert all((second.array(0)[r * per_rank] == r + 1).all() for r in range(world))
    dist.barrier()
    second.close()
    if rank == 0:
        t0 = time.monotonic()
        try:
            ring.wait_all(steps, timeout=0.3)
            raise SystemExit("wait_all returned for a step nobody published")
        except TimeoutError as exc:
            assert "ranks" in str(exc) and 0.25 < time.monotonic() - t0 < 5.0
        try:
            ring.acquire(steps + slots, timeout=0.2)
            raise SystemExit("acquire returned for a slot the consumer has not released")
        except TimeoutError:
            pass
        left = [f for f in os.listdir("/dev/shm") if f.startswith("bgs_rewards_")]
        assert not left, left  # unlinked at construction
        print("SHARED_RING_OK")
    dist.barrier()
    ring.close()
    dist.destroy_process_group()
    """
)


def _run_ring(tmp_path, ranks, per_rank, slots, steps):
    script = tmp_path / "ring_worker.py"
    script.write_text(RING_WORKER.format(root=ROOT, per_rank=per_rank, slots=slots, steps=steps))
    port = _free_port()
    procs = []
    for rank in range(ranks):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(ranks), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=420)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(o[-1500:] for o in outs)
    assert "SHARED_RING_OK" in outs[0]


def test_three_ranks_deliver_into_one_shared_host_array(tmp_path):
    """The N > 1 hand-over bench.py uses on one node: no collective in the data path, every rank expands its own outcome
    codes into its rows of one host array in shared memory; rank 0 consumes (futex waits, release / acquire hand-shake)
    and sees the unsharded run."""
    _run_ring(tmp_path, ranks=3, per_rank=1024, slots=2, steps=6)


def test_eight_ranks_rehearsal_of_the_shared_host_array(tmp_path):
    """The 8-GPU layout as 8 CPU processes: eight producers, a slow consumer three slots behind, the too-small-/dev/shm
    fall-back raised on all eight ranks."""
    _run_ring(tmp_path, ranks=8, per_rank=512, slots=3, steps=9)


def test_shard_range_validation():
    import pytest

    sys.path[:0] = [os.path.join(ROOT, "board-game-simulator-python_amd")]
    from simulator.sharding import shard_range

    assert shard_range(1 << 23, 3, 8) == (3 << 20, 1 << 20)
    assert [shard_range(12, r, 4) for r in range(4)] == [(0, 3), (3, 3), (6, 3), (9, 3)]
    with pytest.raises(ValueError):
        shard_range(10, 0, 4)
    with pytest.raises(ValueError):
        shard_range(8, 4, 4)

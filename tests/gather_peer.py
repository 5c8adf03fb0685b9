"""One rank of the in-library reward gather's rehearsal with PEERS (started by tests/test_gpu_gather_peers.py, never
imported by the product): 2-3 processes share the one GPU, libbgs.so loads the test-only transport
(BGS_RCCL_LIB=tests/c/libfake_rccl.so) and runs its real world > 1 code -- the communication thread, its groups of
ncclSend / ncclRecv, partial groups, the copy to / the direct receives into the sink's page-locked slots, the create-time
transport check.  Rank 0 compares the rows of EVERY rank in EVERY delivered step with the CPU oracle.

    python tests/gather_peer.py <dir> <rank>[,<rank>...] <world> <mode>   mode: steps | inject | inject_one | lone_step | full | multi

Several ranks in one process (each on a thread of its own, with its own batches, gather and communicator rank): the GPU
box allows at most 6 processes on the card, so a world of 8 runs as 4 processes x 2 ranks -- the library's arithmetic
(rank r's codes at dst + r * code_bytes, rank 0's sink expanding world x n games) is that of 8 ranks all the same.
"""

import os
import pickle
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "board-game-simulator-python_amd")]

import numpy as np  # noqa: E402

SEED = 0x0123456789ABCDEF


class FileDist:
    """What RewardGather needs of torch.distributed -- rank, world size, one broadcast of an object -- over a directory."""

    def __init__(self, directory, rank, world):
        self.directory, self.rank, self.world, self.count = directory, rank, world, 0

    def get_rank(self):
        return self.rank

    def get_world_size(self):
        return self.world

    def barrier(self, name):
        """Every rank has arrived at `name` (a file per rank and barrier)."""
        open(os.path.join(self.directory, f"{name}.{self.rank}"), "w").close()
        deadline = time.monotonic() + 120
        for r in range(self.world):
            while not os.path.exists(os.path.join(self.directory, f"{name}.{r}")):
                if time.monotonic() > deadline:
                    raise TimeoutError(f"barrier {name}: rank {r}")
                time.sleep(0.002)

    def broadcast_object_list(self, objects, src=0):
        path = os.path.join(self.directory, f"bcast{self.count}.pkl")
        self.count += 1
        if self.rank == src:
            with open(path + ".tmp", "wb") as fh:
                pickle.dump(list(objects), fh)
            os.rename(path + ".tmp", path)
            return
        deadline = time.monotonic() + 120
        while not os.path.exists(path):
            if time.monotonic() > deadline:
                raise TimeoutError(path)
            time.sleep(0.005)
        with open(path, "rb") as fh:
            objects[:] = pickle.load(fh)


def expected(world, n, seed):
    """The host array rank 0 must end up with: the oracle's rewards of global games [0, world * n)."""
    from oracle import oracle

    orc = oracle.ConnectOracle(6, 7, 4, world * n)
    orc.rollout(seed)
    return orc.reward


def run_steps(directory, rank, world):
    from simulator.batch import ConnectBatch
    from simulator.pipeline import RolloutExecutor
    from simulator.sharding import RewardGather

    n = int(os.environ.get("PEER_GAMES", "4096"))
    slots, depth = 8, 3
    slow = int(os.environ.get("PEER_SLOW_RANK", "-1"))
    dist = FileDist(directory, rank, world)
    batches = []
    for _ in range(depth):
        b = ConnectBatch(6, 7, 4, n, use_torch=False)
        b.set_first_game(rank * n)
        batches.append(b)
    gather = RewardGather(dist, n, slots=slots, host_threads=2)
    info = gather.info()
    hosts = [np.full((world * n, 2), 9, dtype=np.int8) if rank == 0 else None for _ in range(slots)]
    verified = 0

    def check(step):
        nonlocal verified
        if rank == 0:
            want = expected(world, n, SEED + step)
            got = hosts[step % slots]
            for r in range(world):
                assert np.array_equal(got[r * n:(r + 1) * n], want[r * n:(r + 1) * n]), f"step {step}: rank {r}'s rows differ"
        verified += 1

    # (a) single calls.  Every step is checked before its host array is reused; the newest ticket is waited for at a few
    # steps (a PARTIAL group must leave), and one rank dawdles so that the ranks' groups differ in size.
    tickets = {}
    steps = 29
    for s in range(steps):
        if s >= slots:
            gather.wait(tickets[s - slots])
            check(s - slots)
        if rank == slow and s % 5 == 3:
            time.sleep(0.03)
        tickets[s] = gather.rollout(batches[s % depth], hosts[s % slots], SEED + s)
        assert tickets[s] == s
        if s in (2, 9, 10, 22) and rank != slow:
            gather.wait(tickets[s])   # newest ticket: its group is flushed as it is (the old ones are checked in order below)
    for s in range(max(0, steps - slots), steps):
        gather.wait(tickets[s])
        check(s)
    # (b) the native loop on the same gather: tickets go on from `steps`
    exe = RolloutExecutor(batches, gather=gather, host_arrays=hosts, seed0=SEED + 1000)
    done = 0
    for count in (5, 1, 11):
        exe.enqueue(count)
        exe.drain()
        done += count
        # the last min(count, slots) hand-overs are still in their arrays (hand-over j of the executor -> array j % slots)
        for j in range(max(0, done - min(count, slots)), done):
            if rank == 0:
                want = expected(world, n, SEED + 1000 + j)
                assert np.array_equal(hosts[j % slots], want), f"native loop hand-over {j}"
            verified += 1
    exe.close()
    gather.close()
    print(f"PEER_OK rank {rank} verified {verified} info {info}", flush=True)


def run_inject(directory, rank, world):
    """BGS_EXPERIMENT=gather_inject_failure=5: step 5 "cannot be enqueued" after rank 0 claimed its sink ticket.  The call reports
    it, the steps before it are delivered, nothing stalls: waits return, close() returns."""
    from simulator.batch import ConnectBatch
    from simulator.game._abi import BgsError
    from simulator.sharding import RewardGather

    n, slots = 1024, 8
    dist = FileDist(directory, rank, world)
    batch = ConnectBatch(6, 7, 4, n, use_torch=False)
    batch.set_first_game(rank * n)
    gather = RewardGather(dist, n, slots=slots, host_threads=2)
    hosts = [np.full((world * n, 2), 9, dtype=np.int8) if rank == 0 else None for _ in range(slots)]
    tickets = []
    failed_at = None
    for s in range(8):
        try:
            tickets.append(gather.rollout(batch, hosts[s % slots], SEED + s))
        except BgsError as exc:
            failed_at = s
            assert "injected" in str(exc) or "failed earlier" in str(exc), str(exc)
            break
    assert failed_at == 5, failed_at
    # the steps before the failed one: their group was cut short by the failure and is reported as failed or delivered
    # intact -- never left hanging
    outcomes = []
    for t in tickets:
        try:
            gather.wait(t)
            outcomes.append("ok")
        except BgsError:
            outcomes.append("failed")
    if rank == 0:
        for s, o in enumerate(outcomes):
            if o == "ok":
                assert np.array_equal(hosts[s % slots], expected(world, n, SEED + s)), f"step {s} reported ok with wrong rows"
    try:
        gather.rollout(batch, hosts[0], SEED + 99)
        raise SystemExit("a failed gather accepted another step")
    except BgsError:
        pass
    gather.close()   # must return
    print(f"INJECT_OK rank {rank} outcomes {outcomes}", flush=True)


def run_inject_one(directory, rank, world):
    """Round-4 advisor: ONE rank fails locally (BGS_EXPERIMENT: gather_inject_rank names it, gather_inject_failure the step) while
    its peers have posted -- or will post -- the matching halves of the group.  The failing rank still posts its message
    (zeros), so nobody stalls: it reports the failure itself, rank 0 gets every step, with that rank's rows of the failed
    step reading 0 / 0 ("still running") and everything else equal to the oracle -- and, round 6, rank 0's wait for THAT step
    fails: its sink knows that every game of the step had to end and finds games that did not."""
    from simulator.batch import ConnectBatch
    from simulator.game._abi import BgsError
    from simulator.sharding import RewardGather

    n, slots, steps = 1024, 8, 6
    bad_rank, bad_step = int(os.environ["PEER_BAD_RANK"]), int(os.environ["PEER_BAD_STEP"])   # (= BGS_EXPERIMENT's gather_inject_rank / _failure)
    assert bad_step == steps - 1, "the failing rank cannot submit anything after its failure: make it the last step"
    dist = FileDist(directory, rank, world)
    batch = ConnectBatch(6, 7, 4, n, use_torch=False)
    batch.set_first_game(rank * n)
    gather = RewardGather(dist, n, slots=slots, host_threads=2)
    hosts = [np.full((world * n, 2), 9, dtype=np.int8) if rank == 0 else None for _ in range(slots)]
    tickets, failed_at = [], None
    for s in range(steps):
        try:
            tickets.append(gather.rollout(batch, hosts[s % slots], SEED + s))
        except BgsError as exc:
            failed_at = s
            assert "injected" in str(exc), str(exc)
            break
    assert failed_at == (bad_step if rank == bad_rank else None), (rank, failed_at)
    t0 = time.monotonic()
    outcomes = []
    for t in tickets:
        try:
            gather.wait(t)
            outcomes.append("ok")
        except BgsError:
            outcomes.append("failed")
    assert time.monotonic() - t0 < 30, "a peer was left waiting"
    if rank != bad_rank and rank != 0:
        assert outcomes == ["ok"] * steps, outcomes   # the peers' groups completed: the failing rank posted its half
    if rank == 0 and bad_rank != 0:
        # ... and rank 0 is TOLD (round 6): every game of an uncapped rollout from the start must have ended, so the zeros the
        # failing rank's message carried ("still running") fail that step's hand-over on rank 0 -- the steps around it are fine
        assert outcomes == ["ok"] * bad_step + ["failed"] + ["ok"] * (steps - bad_step - 1), outcomes
    gather.close()   # must return on every rank (and drains rank 0's sink: what was delivered is in the arrays now)
    if rank == 0:
        for s in range(steps):
            if s == bad_step and bad_rank == 0:
                continue   # rank 0's own failed step is published as failed: its array is not written
            want = expected(world, n, SEED + s).copy()
            if s == bad_step:
                want[bad_rank * n:(bad_rank + 1) * n] = 0   # the failed rank's message: zeros = "every game still running"
            assert np.array_equal(hosts[s % slots], want), f"step {s}"
    print(f"INJECT_ONE_OK rank {rank} outcomes {outcomes}", flush=True)


def run_lone_step(directory, rank, world):
    """Round-4 advisor: a rank that submits FEWER steps than a group holds and then blocks on something outside the
    library -- here: a barrier every rank only leaves once rank 0 has its rewards -- without waiting for its newest ticket.
    The communication thread sends the partial group by itself (BGS_GATHER_FLUSH_US after its first step)."""
    from simulator.batch import ConnectBatch
    from simulator.sharding import RewardGather

    n, slots = 2048, 8
    dist = FileDist(directory, rank, world)
    batch = ConnectBatch(6, 7, 4, n, use_torch=False)
    batch.set_first_game(rank * n)
    gather = RewardGather(dist, n, slots=slots, host_threads=2)
    assert gather.info()["batch"] == 4
    host = np.full((world * n, 2), 9, dtype=np.int8) if rank == 0 else None
    for round_ in range(3):
        t = gather.rollout(batch, host, SEED + round_)   # ONE step of a group of four
        if rank == 0:
            gather.wait(t)                                # needs every other rank's send: nobody there asks for a flush
            assert np.array_equal(host, expected(world, n, SEED + round_)), f"round {round_}"
        dist.barrier(f"lone{round_}")                     # the other ranks sit here meanwhile
    gather.close()
    print(f"LONE_OK rank {rank}", flush=True)


def run_full(directory, rank, world):
    """BASELINE config 5's shapes through the in-library gather: 2^20 games per rank, 12 slots / host arrays, groups of
    6, 3 batches in flight -- PEER_GAMES / PEER_STEPS shrink it.  Rank 0 compares every rank's rows of every step."""
    from simulator.batch import ConnectBatch
    from simulator.pipeline import RolloutExecutor
    from simulator.sharding import RewardGather

    n = int(os.environ.get("PEER_GAMES", str(1 << 20)))
    steps = int(os.environ.get("PEER_STEPS", "14"))
    slots, depth = 12, 3
    dist = FileDist(directory, rank, world)
    batches = []
    for _ in range(depth):
        b = ConnectBatch(6, 7, 4, n, use_torch=False)
        b.set_first_game(rank * n)
        batches.append(b)
    gather = RewardGather(dist, n, slots=slots, host_threads=min(24, 4 + 2 * world) if rank == 0 else 2)
    info = gather.info()
    assert info["ranks"] == world and info["rank"] == rank and info["batch"] == 6, info
    hosts = [np.full((world * n, 2), 9, dtype=np.int8) if rank == 0 else None for _ in range(slots)]
    exe = RolloutExecutor(batches, gather=gather, host_arrays=hosts, seed0=SEED + 500)
    verified = 0
    done = 0
    for count in (slots, steps - slots) if steps > slots else (steps,):
        exe.enqueue(count)
        exe.drain()
        done += count
        for j in range(max(0, done - min(count, slots)), done):
            if rank == 0:
                want = expected(world, n, SEED + 500 + j)
                got = hosts[j % slots]
                for r in range(world):
                    assert np.array_equal(got[r * n:(r + 1) * n], want[r * n:(r + 1) * n]), f"hand-over {j}: rank {r}'s rows differ"
            verified += 1
    exe.close()
    gather.close()
    print(f"FULL_OK rank {rank} of {world} verified {verified} steps of {world} x {n} games info {info}", flush=True)


def run_multi():
    """bgs_multi_* (one process, all devices) with TWO logical devices on the one GPU: the group of sends and receives
    of bgs_multi_rollout, three steps on the same handle."""
    from simulator.sharding import MultiDeviceRollout, multi_device_rollout

    n = 2048
    multi = MultiDeviceRollout([0, 0], 6, 7, 4, n)
    out = np.zeros((2 * n, 2), dtype=np.int8)
    for k in range(3):
        reward, steps = multi.rollout(SEED + k, out=out if k else None)
        want = expected(2, n, SEED + k)
        assert np.array_equal(reward, want), f"step {k}"
        assert steps > 2 * n * 7
    try:
        multi.rollout(SEED, out=np.zeros((n, 2), dtype=np.int8))
        raise SystemExit("a short destination was accepted")
    except ValueError:
        pass
    multi.close()
    try:
        multi.rollout(SEED)
        raise SystemExit("a closed handle was accepted")
    except RuntimeError:
        pass
    reward, _ = multi_device_rollout([0, 0, 0], 6, 7, 4, 1024, SEED + 7)
    assert np.array_equal(reward, expected(3, 1024, SEED + 7))
    print("MULTI_OK", flush=True)


if __name__ == "__main__":
    directory, ranks, world, mode = sys.argv[1], [int(r) for r in sys.argv[2].split(",")], int(sys.argv[3]), sys.argv[4]
    modes = {"steps": run_steps, "inject": run_inject, "inject_one": run_inject_one, "lone_step": run_lone_step, "full": run_full}
    if mode == "multi":
        run_multi()
    elif len(ranks) == 1:
        modes[mode](directory, ranks[0], world)
    else:
        import threading

        from simulator.game import _abi

        _abi.lib()   # (loaded once, before the threads start)
        errors = []

        def one(rank):
            try:
                modes[mode](directory, rank, world)
            except BaseException as exc:  # noqa: BLE001 -- reported below, with the rank
                import traceback

                errors.append(f"rank {rank}: {type(exc).__name__}: {exc}\n{traceback.format_exc()}")

        threads = [threading.Thread(target=one, args=(r,)) for r in ranks]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        if errors:
            raise SystemExit("\n".join(errors))

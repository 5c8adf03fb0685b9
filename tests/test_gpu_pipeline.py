"""GPU tests of the native rollout loop (bgs_pipeline_*, simulator.pipeline.RolloutExecutor), the in-library RCCL reward
gather (bgs_gather_*, simulator.sharding.RewardGather) with a world of one rank -- what a one-GPU box can run of it --
and the reward sink's behaviour under several submitting threads and bad destinations.  Everything is compared with the
CPU oracle, bit-exact, through the C ABI."""

import os
import subprocess
import sys
import threading

import numpy as np
import pytest

from oracle import oracle

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "board-game-simulator-python_amd")
SEED = 0x0123456789ABCDEF

DEFAULT_BOUNCE = np.zeros((9, 6), dtype=np.int8)
DEFAULT_BOUNCE[1] = DEFAULT_BOUNCE[7] = [1, 2, 3, 3, 2, 1]


def _batches(make, depth):
    import torch

    streams = [torch.cuda.Stream() for _ in range(depth)]
    out = []
    for s in streams:
        with torch.cuda.stream(s):
            out.append(make())
    return streams, out


@pytest.mark.parametrize("depth,slots", [(1, 2), (3, 9), (4, 5)])
def test_executor_delivers_every_step_in_order(depth, slots):
    """Step s = batch s % depth, seed seed0 + s, host array (hand-over index) % slots: the arrays of the last `slots`
    hand-overs equal the oracle's rewards for their seeds, whatever depth and slot count; steps without hand-over in
    between advance the seed but not the host array."""
    from simulator.batch import ConnectBatch, RewardSink
    from simulator.pipeline import RolloutExecutor

    n, first = 6000, 123
    streams, batches = _batches(lambda: ConnectBatch(6, 7, 4, n, use_torch=True), depth)
    for b in batches:
        b.set_first_game(first)
    hosts = [np.full((n, 2), 9, dtype=np.int8) for _ in range(slots)]
    sink = RewardSink(n, slots=slots, threads=3)
    with RolloutExecutor(batches, sink=sink, host_arrays=hosts, seed0=SEED + 40) as exe:
        exe.enqueue(7, True, time_stride=2)
        exe.enqueue(3, False)             # steps 7, 8, 9 stay on the device
        exe.enqueue(slots + 2, True)
        exe.drain()
        assert exe.steps == 12 + slots and exe.handovers == 9 + slots
        line = exe.timeline()                 # (start, end) of the bracketed launches, ms after the first one's start
        assert len(line) == 4 and line[0][0] == 0.0 and all(0.0 <= a < z for a, z in line)
        ms, pairs = exe.kernel_ms()
        assert pairs == 4 and ms > 0 and abs(ms - sum(z - a for a, z in line) / 4) < 1e-3
        assert exe.kernel_ms() == (None, 0) and exe.timeline() == []   # the brackets are consumed
        exe.drain(); exe.drain()              # (a drain with nothing to wait for, twice: the helpers are idle)
        # hand-over j came from step j (j < 7) or step j + 3 (j >= 7)
        for j in range(exe.handovers - slots, exe.handovers):
            step = j if j < 7 else j + 3
            orc = oracle.ConnectOracle(6, 7, 4, n)
            orc.rollout(SEED + 40 + step, first_game=first)
            np.testing.assert_array_equal(hosts[j % slots], orc.reward, err_msg=f"hand-over {j} (step {step})")
        assert exe.last_host_array() is hosts[(exe.handovers - 1) % slots]
        assert sum(b.steps for b in batches) > (12 + slots) * n * 7
    sink.close()
    for b in batches:
        b.close()


def test_executor_bounce_with_a_ply_cap():
    from simulator.batch import BounceBatch, RewardSink
    from simulator.pipeline import RolloutExecutor

    n = 900
    streams, batches = _batches(lambda: BounceBatch(DEFAULT_BOUNCE, n, use_torch=True), 2)
    hosts = [np.zeros((n, 2), dtype=np.int8) for _ in range(4)]
    sink = RewardSink(n, slots=4, threads=2)
    with RolloutExecutor(batches, sink=sink, host_arrays=hosts, seed0=5, max_plies=300) as exe:
        exe.enqueue(6)
        exe.drain()
        for j in range(2, 6):
            orc = oracle.BounceOracle(DEFAULT_BOUNCE, n)
            orc.rollout(5 + j, max_plies=300)
            np.testing.assert_array_equal(hosts[j % 4], orc.reward, err_msg=f"step {j}")
    sink.close()


def test_executor_argument_checks():
    from simulator.batch import ConnectBatch, RewardSink
    from simulator.pipeline import RolloutExecutor

    b = ConnectBatch(6, 7, 4, 64)
    sink = RewardSink(64, slots=2, threads=1)
    with pytest.raises(ValueError):
        RolloutExecutor([b, b], sink=sink, host_arrays=[np.zeros((64, 2), np.int8)])  # the same batch twice
    with pytest.raises(ValueError):
        RolloutExecutor([b], sink=sink, host_arrays=[np.zeros((63, 2), np.int8)])     # destination too small
    with pytest.raises(ValueError):
        RolloutExecutor([b], sink=sink, host_arrays=[])                               # a hand-over without arrays
    exe = RolloutExecutor([b])                                                        # no hand-over at all
    with pytest.raises(ValueError):
        exe.enqueue(1, True)
    exe.enqueue(2, False)
    exe.drain()
    assert b.steps > 0
    exe.close()
    sink.close()


def test_sink_serves_two_submitting_threads():
    """Two threads, each with its own batch and stream, submit to ONE sink at the same time: tickets are reserved under
    the sink's lock, so no two submissions share a slot and every host array ends up with its own step's rewards."""
    from simulator.batch import ConnectBatch, RewardSink

    n, rounds = 3000, 25
    streams, batches = _batches(lambda: ConnectBatch(6, 7, 4, n, use_torch=True), 2)
    sink = RewardSink(n, slots=3, threads=2)
    hosts = [[np.zeros((n, 2), dtype=np.int8) for _ in range(rounds)] for _ in range(2)]
    tickets = [[], []]
    errors = []

    def worker(k):
        try:
            for r in range(rounds):
                tickets[k].append(sink.rollout(batches[k], hosts[k][r], SEED + 1000 * k + r, from_initial=True))
        except Exception as exc:  # noqa: BLE001
            errors.append(exc)

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    assert sorted(tickets[0] + tickets[1]) == list(range(2 * rounds))  # every ticket handed out exactly once
    sink.wait(2 * rounds - 1)
    assert sink.completed == 2 * rounds
    for k in range(2):
        for r in range(rounds):
            orc = oracle.ConnectOracle(6, 7, 4, n)
            orc.rollout(SEED + 1000 * k + r)
            np.testing.assert_array_equal(hosts[k][r], orc.reward, err_msg=f"thread {k} round {r}")
    sink.close()


def test_sink_checks_and_keeps_its_destinations():
    from simulator.batch import ConnectBatch, RewardSink

    n = 4096
    b = ConnectBatch(6, 7, 4, n)
    sink = RewardSink(n, slots=2, threads=2)
    with pytest.raises(ValueError):
        sink.rollout(b, np.zeros((n - 1, 2), dtype=np.int8), SEED, from_initial=True)     # too small
    with pytest.raises(TypeError):
        sink.rollout(b, np.zeros((n, 2), dtype=np.int16), SEED, from_initial=True)        # wrong item size
    ro = np.zeros((n, 2), dtype=np.int8)
    ro.setflags(write=False)
    with pytest.raises(ValueError):
        sink.rollout(b, ro, SEED, from_initial=True)                                        # read-only
    with pytest.raises(TypeError):
        sink.rollout(b, np.zeros((n, 4), dtype=np.int8)[:, :2], SEED, from_initial=True)  # not contiguous
    # the sink holds on to a destination the caller drops before the delivery
    t = sink.rollout(b, np.zeros((n, 2), dtype=np.int8), SEED, from_initial=True)
    kept = sink._alive[t][0]
    sink.wait(t)
    orc = oracle.ConnectOracle(6, 7, 4, n)
    orc.rollout(SEED)
    np.testing.assert_array_equal(kept, orc.reward)
    assert t not in sink._alive
    # close() waits for what is still in flight
    last = np.zeros((n, 2), dtype=np.int8)
    sink.rollout(b, last, SEED + 1, from_initial=True)
    sink.close()
    orc.reset()
    orc.rollout(SEED + 1)
    np.testing.assert_array_equal(last, orc.reward)


GATHER_CHILD = """
import os, sys
sys.path[:0] = [{root!r}, {pkg!r}]
import numpy as np, torch, torch.distributed as dist
from oracle import oracle
from simulator.batch import ConnectBatch
from simulator.pipeline import RolloutExecutor
from simulator.sharding import RewardGather
dist.init_process_group(backend="gloo", init_method="tcp://127.0.0.1:{port}", rank=0, world_size=1)
n, depth, slots, seed0 = 40000, 3, 6, 0x0123456789ABCDEF + 9
streams = [torch.cuda.Stream() for _ in range(depth)]
batches = []
for s in streams:
    with torch.cuda.stream(s):
        batches.append(ConnectBatch(6, 7, 4, n, use_torch=True))
gather = RewardGather(dist, n, slots=slots, host_threads=3)
hosts = [np.full((n, 2), 7, dtype=np.int8) for _ in range(slots)]
exe = RolloutExecutor(batches, gather=gather, host_arrays=hosts, seed0=seed0)
exe.enqueue(20, True, 4)
exe.drain()
for j in range(20 - slots, 20):
    orc = oracle.ConnectOracle(6, 7, 4, n)
    orc.rollout(seed0 + j)
    assert np.array_equal(hosts[j % slots], orc.reward), f"step {{j}}"
# single calls through the object, too
t = gather.rollout(batches[0], hosts[0], 77)
gather.wait(t)
orc = oracle.ConnectOracle(6, 7, 4, n)
orc.rollout(77)
assert np.array_equal(hosts[0], orc.reward)
try:
    gather.rollout(batches[0], np.zeros((n - 4, 2), np.int8), 78)
    raise SystemExit("a short destination was accepted")
except ValueError:
    pass
exe.close(); gather.close()
dist.destroy_process_group()
print("GATHER_OK", os.environ.get("BGS_GATHER_DIRECT", "0"))
"""


@pytest.mark.parametrize("direct", ["0", "1"])
def test_in_library_rccl_gather_with_one_rank(direct):
    """bgs_gather_* over the real RCCL library with a world of ONE rank -- all that RCCL itself allows on a one-GPU box.
    What this runs: librccl loads, ncclGetUniqueId / ncclCommInitRank succeed, and the gather's one-rank form -- no
    communication thread, no send and no receive (there is no peer): a step is the sink's bgs_sink_rollout -- delivers
    every step through the native loop and by single calls; a short destination is refused.  The sends, receives, groups
    and the direct / copy receive modes run with peers in tests/test_gpu_gather_peers.py (over the tests' stand-in
    transport); BGS_GATHER_DIRECT is passed through here only to show that it is harmless with one rank.  (Child
    process under `timeout`: a collective that does not complete must not take the test session along.)"""
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    code = GATHER_CHILD.format(root=ROOT, pkg=PKG, port=port)
    env = dict(os.environ, BGS_GATHER_DIRECT=direct)
    proc = subprocess.run(["timeout", "-k", "10", "300", sys.executable, "-c", code], capture_output=True, text=True, env=env)
    assert proc.returncode == 0 and "GATHER_OK" in proc.stdout, proc.stdout[-2000:] + proc.stderr[-3000:]


@pytest.mark.parametrize("case", ["connect6x7", "connect12x13", "connect3x4", "bounce", "bounce_generic", "connect_generic"])
def test_grid_sink_delivers_the_boards_of_every_step(case):
    """The overlapped grid hand-over (State.grid of every game of a step, int8[n, H, W], reference connect.cpp:42 /
    bounce.cpp:39): bit-packed boards cross PCIe into page-locked slots, worker threads expand them.  Host grids of the
    last `slots` steps == oracle grids, for one- and three-word Connect boards, a tiny one, Bounce (value planes), and the
    generic batches whose device layout is the grid itself; through the native loop and by single calls."""
    from simulator.batch import BounceBatch, ConnectBatch, GridSink, RewardSink
    from simulator.pipeline import RolloutExecutor

    big = np.zeros((10, 8), dtype=np.int8)
    big[1] = big[8] = [1, 2, 3, 4, 4, 3, 2, 1]
    make, orc_of, cap = {
        "connect6x7": (lambda n: ConnectBatch(6, 7, 4, n, use_torch=True), lambda n: oracle.ConnectOracle(6, 7, 4, n), 2**31 - 1),
        "connect12x13": (lambda n: ConnectBatch(12, 13, 5, n, use_torch=True), lambda n: oracle.ConnectOracle(12, 13, 5, n), 2**31 - 1),
        "connect3x4": (lambda n: ConnectBatch(3, 4, 3, n, use_torch=True), lambda n: oracle.ConnectOracle(3, 4, 3, n), 5),
        "bounce": (lambda n: BounceBatch(DEFAULT_BOUNCE, n, use_torch=True), lambda n: oracle.BounceOracle(DEFAULT_BOUNCE, n), 300),
        "bounce_generic": (lambda n: BounceBatch(big, n, use_torch=True), lambda n: oracle.BounceOracle(big, n), 200),
        "connect_generic": (lambda n: ConnectBatch(20, 20, 5, n, use_torch=True), lambda n: oracle.ConnectOracle(20, 20, 5, n), 2**31 - 1),
    }[case]
    n, depth, slots = 3001, 2, 4
    streams, batches = _batches(lambda: make(n), depth)
    h, w = batches[0].height, batches[0].width
    sink = GridSink(batches[0], slots=slots, threads=3)
    hosts = [np.full((n, h, w), 77, dtype=np.int8) for _ in range(slots)]
    with RolloutExecutor(batches, sink=sink, host_arrays=hosts, seed0=SEED + 7, max_plies=cap) as exe:
        exe.enqueue(7)
        exe.drain()
        for j in range(7 - slots, 7):
            orc = orc_of(n)
            orc.rollout(SEED + 7 + j, max_plies=cap)
            np.testing.assert_array_equal(hosts[j % slots], orc.grid, err_msg=f"{case} step {j}")
    # single calls: the current boards (submit) and rollout + boards
    g = np.zeros((n, h, w), dtype=np.int8)
    t = sink.submit(batches[0], g)
    sink.wait(t)
    np.testing.assert_array_equal(g, batches[0].grid)
    with pytest.raises(ValueError):
        sink.rollout(batches[0], np.zeros((n, h, w - 1), dtype=np.int8), SEED, from_initial=True)
    with pytest.raises(TypeError):
        sink.submit_packed(None, n, g)
    other = ConnectBatch(4, 4, 3, n)
    if (h, w) != (4, 4):
        with pytest.raises(ValueError):
            sink.rollout(other, g, SEED, from_initial=True)   # a batch the sink was not made for
    sink.close()
    # a reward sink and a grid sink side by side on the same batch
    rs, gs = RewardSink(n, slots=2, threads=2), GridSink(batches[1], slots=2, threads=2)
    rew = np.zeros((n, 2), dtype=np.int8)
    t1 = rs.rollout(batches[1], rew, SEED + 99, max_plies=cap, from_initial=True)
    t2 = gs.submit(batches[1], g)
    rs.wait(t1)
    gs.wait(t2)
    orc = orc_of(n)
    orc.rollout(SEED + 99, max_plies=cap)
    np.testing.assert_array_equal(rew, orc.reward)
    np.testing.assert_array_equal(g, orc.grid)
    rs.close()
    gs.close()


def test_rollout_pipeline_refuses_a_depth_it_cannot_deliver():
    """More than 4 batches in flight need more than the HIP runtime's default 4 hardware queues, which can only be asked
    for before the runtime comes up.  Importing `simulator` asks for nothing (it used to set GPU_MAX_HW_QUEUES for the
    whole process); a pipeline asks when it is built.  A process that initialised HIP first: an EXPLICIT depth of 16 is
    refused loudly, the DEFAULT Bounce depth falls back to 4 with a warning; a process that builds the pipeline first
    gets the full default depth."""
    code_late = (
        "import sys, os, warnings; sys.path[:0] = [%r, %r]\n"
        "os.environ.pop('GPU_MAX_HW_QUEUES', None)\n"
        "import simulator\n"
        "assert 'GPU_MAX_HW_QUEUES' not in os.environ\n"             # the import leaves the process alone
        "import torch; assert torch.cuda.is_available()\n"          # the HIP runtime is up, with 4 queues
        "import numpy as np\n"
        "from simulator.batch import BounceBatch\n"
        "from simulator.pipeline import RolloutPipeline, request_hardware_queues\n"
        "from simulator.game import _abi\n"
        "assert request_hardware_queues() == 4 and _abi.hardware_queues() == 4 and 'GPU_MAX_HW_QUEUES' not in os.environ\n"
        "g = np.zeros((9, 6), dtype=np.int8); g[1] = g[7] = [1, 2, 3, 3, 2, 1]\n"
        "try:\n"
        "    RolloutPipeline(BounceBatch, (g,), 512, depth=16)\n"
        "    raise SystemExit('depth 16 on 4 hardware queues was accepted')\n"
        "except RuntimeError as exc:\n"
        "    assert 'GPU_MAX_HW_QUEUES' in str(exc)\n"
        "with warnings.catch_warnings(record=True) as seen:\n"
        "    warnings.simplefilter('always')\n"
        "    with RolloutPipeline(BounceBatch, (g,), 512, max_plies=200) as pipe:\n"
        "        assert pipe.depth == 4 and len(list(pipe.run(range(6)))) == 6\n"
        "assert any('hardware queues' in str(w.message) for w in seen), [str(w.message) for w in seen]\n"
        "with RolloutPipeline(BounceBatch, (g,), 512, depth=3, max_plies=200) as pipe:\n"
        "    assert len(list(pipe.run(range(4)))) == 4\n"
        "print('LATE_OK')\n" % (ROOT, PKG))
    code_early = (
        "import sys, os; sys.path[:0] = [%r, %r]\n"
        "os.environ.pop('GPU_MAX_HW_QUEUES', None)\n"
        "import numpy as np\n"
        "import simulator\n"
        "from simulator.batch import BounceBatch\n"
        "from simulator.pipeline import RolloutPipeline\n"
        "from simulator.game import _abi\n"
        "assert 'GPU_MAX_HW_QUEUES' not in os.environ\n"
        "g = np.zeros((9, 6), dtype=np.int8); g[1] = g[7] = [1, 2, 3, 3, 2, 1]\n"
        "with RolloutPipeline(BounceBatch, (g,), 512, max_plies=200) as pipe:\n"   # asks for the queues itself, in time
        "    assert os.environ['GPU_MAX_HW_QUEUES'] == '24' and _abi.hardware_queues() == 24\n"
        "    assert pipe.depth == 20 and len(list(pipe.run(range(24)))) == 24\n"
        "print('EARLY_OK')\n" % (ROOT, PKG))
    for code, word in ((code_late, "LATE_OK"), (code_early, "EARLY_OK")):
        proc = subprocess.run(["timeout", "-k", "10", "300", sys.executable, "-c", code], capture_output=True, text=True)
        assert proc.returncode == 0 and word in proc.stdout, proc.stdout[-1500:] + proc.stderr[-3000:]

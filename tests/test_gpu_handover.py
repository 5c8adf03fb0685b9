"""GPU tests of what sits either side of the rollout kernels: the asynchronous hand-over of rewards to host memory
(the batch form of State.reward, reference connect.cpp:41 / tensor.hpp:69-87), the device-tensor interface for
policy-driven stepping (N2), batch <-> reference-shaped JSON states (N4), stream ordering, loader validation and
bench.py's own N = 2 loop.  Everything goes through the C ABI and is compared with the CPU oracle, bit-exact."""

import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from tests.knobs import product_env

from oracle import oracle

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SEED = 0x0123456789ABCDEF

DEFAULT_BOUNCE = np.zeros((9, 6), dtype=np.int8)
DEFAULT_BOUNCE[1] = DEFAULT_BOUNCE[7] = [1, 2, 3, 3, 2, 1]


@pytest.fixture(scope="module")
def bm():
    from simulator import batch

    return batch


@pytest.fixture(scope="module")
def torch_mod():
    import torch

    return torch


# ------------------------------------------------------------------------------------------------ hand-over

@pytest.mark.parametrize("n", [1, 5, 4096, 100003])
def test_async_reward_and_outcome_reads(bm, n):
    dev = bm.ConnectBatch(6, 7, 4, n)
    orc = oracle.ConnectOracle(6, 7, 4, n)
    pairs = bm.PinnedArray((n, 2), np.int8)
    codes = bm.PinnedArray(((n + 3) // 4,), np.uint8)
    ev1, ev2 = bm.HostEvent(0), bm.HostEvent(0)
    pairs.array[:] = 55
    dev.rollout(SEED + 1, from_initial=True)
    dev.read_reward_async(pairs, ev1)
    dev.read_outcomes_async(codes, ev2)
    orc.rollout(SEED + 1)
    ev2.synchronize()  # same stream: the earlier copy has landed too
    assert ev1.done() and ev2.done()
    np.testing.assert_array_equal(pairs.array, orc.reward)
    np.testing.assert_array_equal(bm.expand_outcomes_host(codes, n), orc.reward)
    # one library call per step, both forms; an unfinished batch (capped) reports 0 / 0 for running boards
    dev.rollout_to_host(pairs, SEED + 2, max_plies=9, from_initial=True, codes=False, event=ev1)
    dev.rollout_to_host(codes, SEED + 3, from_initial=True, codes=True, event=ev2)
    ev1.synchronize()
    orc.reset()
    orc.rollout(SEED + 2, max_plies=9)
    np.testing.assert_array_equal(pairs.array, orc.reward)
    ev2.synchronize()
    orc.reset()
    orc.rollout(SEED + 3)
    np.testing.assert_array_equal(bm.expand_outcomes_host(codes, n), orc.reward)
    for obj in (pairs, codes, ev1, ev2, dev):
        obj.close()


@pytest.mark.parametrize("game", ["connect", "bounce"])
def test_reward_sink_delivers_every_step_in_order(bm, game):
    """K steps through a sink with fewer slots than steps and several batches in flight: every step's host array must
    equal the oracle's rewards of that step's seed (the last K steps of a bench-like loop)."""
    n, steps, depth = 30000, 7, 3
    if game == "connect":
        make_dev = lambda: bm.ConnectBatch(6, 7, 4, n)
        make_orc = lambda: oracle.ConnectOracle(6, 7, 4, n)
        kw = {}
    else:
        n = 3000
        make_dev = lambda: bm.BounceBatch(DEFAULT_BOUNCE, n)
        make_orc = lambda: oracle.BounceOracle(DEFAULT_BOUNCE, n)
        kw = {"max_plies": 300}
    batches = [make_dev() for _ in range(depth)]
    sink = bm.RewardSink(n, slots=2, threads=3)
    hosts = [np.full((n, 2), 42, dtype=np.int8) for _ in range(steps)]
    tickets = []
    for i in range(steps):
        b = batches[i % depth]
        if i % 2:
            b.rollout(SEED + i, from_initial=True, **kw)
            tickets.append(sink.submit(b, hosts[i]))
        else:  # the fused call
            tickets.append(sink.rollout(b, hosts[i], SEED + i, from_initial=True, **kw))
    assert tickets == list(range(steps))
    sink.wait(tickets[-1])  # submissions complete in order
    orc = make_orc()
    for i in range(steps):
        orc.reset()
        orc.rollout(SEED + i, **kw)
        np.testing.assert_array_equal(hosts[i], orc.reward, err_msg=f"step {i}")
    with pytest.raises(ValueError):
        sink.wait(steps)  # unknown ticket
    sink.close()
    for b in batches:
        b.close()


@pytest.mark.parametrize("h,w,k,n", [(6, 7, 4, 70001), (4, 5, 3, 999), (8, 8, 4, 4096), (12, 13, 5, 3000), (6, 7, 4, 16)])
def test_sink_rollout_fused_codes_all_entry_states(bm, h, w, k, n):
    """bgs_sink_rollout lets the rollout kernel itself deliver the outcome codes where it can (one-word boards; the
    pack kernel follows otherwise): from the initial state, from mid-game positions that include finished boards,
    capped, and with ragged batch sizes, the host rewards must equal the oracle's."""
    dev = bm.ConnectBatch(h, w, k, n)
    orc = oracle.ConnectOracle(h, w, k, n)
    sink = bm.RewardSink(n, slots=2, threads=2)
    host = np.full((n, 2), 9, dtype=np.int8)
    sink.wait(sink.rollout(dev, host, SEED + 4, from_initial=True))
    orc.rollout(SEED + 4)
    np.testing.assert_array_equal(host, orc.reward)
    np.testing.assert_array_equal(dev.reward, orc.reward)
    # mid-game: some boards already over, the rest resume; then a cap that leaves boards unfinished (reward 0 / 0)
    dev.reset()
    orc.reset()
    for _ in range(min(h * w, 9)):
        dev.step_random(SEED ^ 3)
        orc.step_random(SEED ^ 3)
    cap = min(h * w, 9) + 3
    host[:] = 9
    sink.wait(sink.rollout(dev, host, SEED ^ 3, max_plies=cap))
    orc.rollout(SEED ^ 3, max_plies=cap)
    np.testing.assert_array_equal(host, orc.reward)
    host[:] = 9
    sink.wait(sink.rollout(dev, host, SEED ^ 3))
    orc.rollout(SEED ^ 3)
    np.testing.assert_array_equal(host, orc.reward)
    np.testing.assert_array_equal(dev.grid, orc.grid)
    assert dev.steps == int(orc.plies.sum())
    sink.close()
    dev.close()


@pytest.mark.parametrize("case", ["connect6x7", "connect4x5", "connect12x13", "connect8x9", "bounce"])
def test_rollout_pack_equals_rollout_then_pack(bm, torch_mod, case):
    """bgs_rollout_pack (what a rank hands to the RCCL gather): the codes the rollout kernels write themselves, or the
    pack kernel behind the others, equal bgs_rollout + bgs_pack_outcomes and the oracle's winners."""
    torch = torch_mod
    n = 50001
    if case == "bounce":
        dev, orc, kw = bm.BounceBatch(DEFAULT_BOUNCE, n, use_torch=True), oracle.BounceOracle(DEFAULT_BOUNCE, n), {"max_plies": 200}
    else:
        h, w, k = {"connect6x7": (6, 7, 4), "connect4x5": (4, 5, 3), "connect12x13": (12, 13, 5), "connect8x9": (8, 9, 4)}[case]
        dev, orc, kw = bm.ConnectBatch(h, w, k, n, use_torch=True), oracle.ConnectOracle(h, w, k, n), {}
    buf = torch.full(((n + 63) // 64 * 16,), 0xAA, dtype=torch.uint8, device="cuda")
    dev.rollout_outcomes_tensor(buf, SEED + 9, from_initial=True, **kw)
    want = dev.outcomes_tensor()
    torch.cuda.synchronize()
    nbytes = (n + 3) // 4
    assert torch.equal(buf[:nbytes], want[:nbytes])
    orc.rollout(SEED + 9, **kw)
    np.testing.assert_array_equal(bm.expand_outcomes_host(buf[:nbytes].cpu().numpy(), n), orc.reward)
    dev.close()


def test_reward_sink_takes_gathered_codes(bm, torch_mod):
    """Rank 0's side of the multi-GPU gather: codes of several shards, already on the device, to one host array."""
    torch = torch_mod
    n, shards = 8192, 3
    parts = []
    want = []
    for r in range(shards):
        b = bm.ConnectBatch(6, 7, 4, n, use_torch=True)
        b.set_first_game(r * n)
        b.rollout(SEED, from_initial=True)
        parts.append(b.outcomes_tensor())
        o = oracle.ConnectOracle(6, 7, 4, n)
        o.rollout(SEED, first_game=r * n)
        want.append(o.reward)
        b.close()
    gathered = torch.cat(parts)
    torch.cuda.synchronize()
    sink = bm.RewardSink(shards * n, slots=2, threads=2)
    host = np.zeros((shards * n, 2), dtype=np.int8)
    t = sink.submit_packed(gathered, shards * n, host, stream=torch.cuda.current_stream().cuda_stream)
    sink.wait(t)
    np.testing.assert_array_equal(host, np.concatenate(want))
    whole = oracle.ConnectOracle(6, 7, 4, shards * n)
    whole.rollout(SEED)
    np.testing.assert_array_equal(host, whole.reward)
    sink.close()


MULTI_CHILD = """
import sys
sys.path[:0] = [{root!r}, {pkg!r}]
import numpy as np
from oracle import oracle
from simulator.game import _abi
from simulator.sharding import MultiDeviceRollout, multi_device_rollout
devices = list(range(min(_abi.device_count(), 2)))
n, seed = 40000, 0x0123456789ABCDEF + 5
reward, steps = multi_device_rollout(devices, 6, 7, 4, n, seed)
whole = oracle.ConnectOracle(6, 7, 4, n * len(devices))
assert whole.rollout(seed) == steps, (steps,)
assert np.array_equal(reward, whole.reward)
try:
    multi_device_rollout(devices, 6, 7, 4, 1001, seed)
    raise SystemExit("n_per_device = 1001 was accepted")
except ValueError:
    pass
# the same with batches, streams and communicators kept between steps
multi = MultiDeviceRollout(devices, 6, 7, 4, n)
for k in range(4):
    reward, steps = multi.rollout(seed + 10 + k)
    whole.reset()
    assert whole.rollout(seed + 10 + k) == steps
    assert np.array_equal(reward, whole.reward), k
multi.close()
print("MULTI_OK", len(devices), steps)
"""


def test_multi_device_entry_point_over_rccl():
    """bgs_multi_connect_rollout (one process, RCCL send/recv gather) on the devices this box has: the host array must
    equal the unsharded oracle run.  With one GPU the gather is a self send/recv -- the RCCL call sequence still runs.
    (In a child process under `timeout`: a collective that does not complete must not take the test session along.)"""
    code = MULTI_CHILD.format(root=ROOT, pkg=os.path.join(ROOT, "board-game-simulator-python_amd"))
    proc = subprocess.run(["timeout", "-k", "10", "240", sys.executable, "-c", code], capture_output=True, text=True)
    assert proc.returncode == 0 and "MULTI_OK" in proc.stdout, proc.stdout[-2000:] + proc.stderr[-3000:]


def test_batch_created_under_a_side_stream_is_ordered(bm, torch_mod):
    """A batch created inside `with torch.cuda.stream(s)` resets itself on the stream it is created on and is then
    re-bound to s: the first work on s must see the reset boards (stream-ordering race)."""
    torch = torch_mod
    n = 1 << 18
    orc = oracle.ConnectOracle(6, 7, 4, n)
    orc.rollout(SEED ^ 9)
    for _ in range(4):
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            dev = bm.ConnectBatch(6, 7, 4, n, use_torch=True)
            dev.rollout(SEED ^ 9, from_initial=False)  # reads the planes and status the reset has to have written
        s.synchronize()
        np.testing.assert_array_equal(dev.reward, orc.reward)
        np.testing.assert_array_equal(dev.plies, orc.plies)
        dev.close()


# ------------------------------------------------------------------------------------------------ N2: device tensors

def test_connect_policy_loop_on_device_tensors(bm, torch_mod):
    """Config 2's geometry: legal mask / grid / action count as device tensors, actions as a device int32 tensor --
    no host round trip -- against the host readers and the oracle, ply by ply."""
    torch = torch_mod
    n = 1 << 16
    dev = bm.ConnectBatch(6, 7, 4, n, use_torch=True)
    orc = oracle.ConnectOracle(6, 7, 4, n)
    gen = torch.Generator(device="cuda").manual_seed(5)
    legal = torch.empty((n, 7), dtype=torch.uint8, device="cuda")
    grid = torch.empty((n, 6, 7), dtype=torch.int8, device="cuda")
    for ply in range(43):  # 42 cells + the ply lost to the refused moves below
        dev.legal_tensor(legal)
        np.testing.assert_array_equal(legal.cpu().numpy(), orc.legal(), err_msg=f"legal, ply {ply}")
        if ply % 6 == 0:
            np.testing.assert_array_equal(dev.grid_tensor(grid).cpu().numpy(), orc.grid)
            np.testing.assert_array_equal(dev.action_count_tensor().cpu().numpy(), orc.legal().sum(axis=1))
            np.testing.assert_array_equal(legal.cpu().numpy(), dev.legal)
        # a torch "policy": random scores, illegal columns masked out; boards that have ended skip (-1)
        scores = torch.rand((n, 7), device="cuda", generator=gen) + legal.float()
        cols = torch.where(legal.bool().any(dim=1), scores.argmax(dim=1), torch.full((n,), -1, device="cuda")).to(torch.int32)
        if ply == 3:
            cols[::97] = 7  # out of range: refused, board untouched
        st_dev = dev.step_actions(cols)
        st_orc = orc.step_actions(cols.cpu().numpy())
        np.testing.assert_array_equal(st_dev, st_orc)
        np.testing.assert_array_equal(dev.reward_copy_tensor().cpu().numpy(), orc.reward, err_msg=f"reward, ply {ply}")
    assert orc.ended.all() and (st_orc == 0).all()
    np.testing.assert_array_equal(dev.grid, orc.grid)
    assert dev.steps == int(orc.plies.sum())
    # without a status read-back the call does not synchronise at all
    dev.reset()
    orc.reset()
    cols = torch.full((n,), 3, dtype=torch.int32, device="cuda")
    assert dev.step_actions(cols, want_status=False) is None
    orc.step_actions(cols.cpu().numpy())
    np.testing.assert_array_equal(dev.grid, orc.grid)
    with pytest.raises(TypeError):
        dev.step_actions(cols.to(torch.int64))
    dev.close()


def test_a_refused_hand_over_does_not_poison_the_sink(bm):
    """Round-3 advisor: bgs_sink_rollout claimed its ticket before the rollout's arguments were checked, so a plain argument
    error was published as a failed delivery and the sink's sticky flag failed EVERY later wait.  Arguments are now refused
    before a ticket exists, and failures belong to tickets: after the refusal the sink delivers as before."""
    n = 4096
    dev = bm.ConnectBatch(6, 7, 4, n)
    sink = bm.RewardSink(n, slots=2, threads=2)
    host = np.zeros((n, 2), dtype=np.int8)
    t0 = sink.rollout(dev, host, SEED, from_initial=True)
    with pytest.raises(ValueError):
        sink.rollout(dev, host, SEED + 1, max_plies=-1, from_initial=True)      # refused: no ticket was taken
    with pytest.raises(ValueError):
        sink.rollout(dev, np.zeros((n - 1, 2), dtype=np.int8), SEED + 1, from_initial=True)
    sink.wait(t0)
    orc = oracle.ConnectOracle(6, 7, 4, n)
    orc.rollout(SEED)
    np.testing.assert_array_equal(host, orc.reward)
    for k in range(5):   # tickets go on where they were; every delivery is good
        t = sink.rollout(dev, host, SEED + 10 + k, from_initial=True)
        assert t == t0 + 1 + k
        sink.wait(t)
        orc.reset()
        orc.rollout(SEED + 10 + k)
        np.testing.assert_array_equal(host, orc.reward)
    sink.close()
    dev.close()


@pytest.mark.parametrize("geometry,n", [((6, 7, 4), 1 << 20), ((6, 7, 4), 1 << 16), ((6, 7, 4), 1022), ((4, 5, 3), 3000), ((8, 8, 5), 2048),
                                         ((6, 7, 4), 4097), ((12, 13, 5), 1024), ((20, 20, 5), 256)])
def test_one_call_per_policy_ply_connect(bm, torch_mod, geometry, n):
    """bgs_step_actions_observe: the chosen moves, the NEXT legal mask, the ended flags and the per-board results in one
    call -- one kernel for one-word boards and an even batch (config 2's geometry, a smaller and a larger one-word board,
    the BASELINE batch of 2^20 boards and sizes around the workgroup boundary), the separate kernels back to back otherwise (odd batch, multi-word and
    generic boards) -- against the oracle's step_actions + legal() + ended, ply by ply, with refused moves (column out of
    range, full column, boards that have ended) in between."""
    torch = torch_mod
    h, w, k = geometry
    dev = bm.ConnectBatch(h, w, k, n, use_torch=True)
    orc = oracle.ConnectOracle(h, w, k, n)
    gen = torch.Generator(device="cuda").manual_seed(11)
    legal = dev.legal_tensor()
    ended = torch.full((n,), 9, dtype=torch.uint8, device="cuda")
    status = torch.full((n,), 9, dtype=torch.int32, device="cuda")
    for ply in range(h * w + 2):
        np.testing.assert_array_equal(legal.cpu().numpy(), orc.legal(), err_msg=f"legal before ply {ply}")
        scores = torch.rand((n, w), device="cuda", generator=gen) + legal.float()
        cols = scores.argmax(dim=1).to(torch.int32)          # a legal column while there is one -- ended boards get one too
        if ply == 2:
            cols[::5] = w          # out of range
            cols[1::5] = -1        # skipped
        if ply == h + 1:
            cols[:] = 0            # column 0, full on many boards by now
        want = orc.step_actions(cols.cpu().numpy())
        out = dev.step_actions_observe(cols, legal, ended=ended, status=status)
        assert out is legal
        np.testing.assert_array_equal(status.cpu().numpy(), want, err_msg=f"status, ply {ply}")
        np.testing.assert_array_equal(ended.cpu().numpy().astype(bool), orc.ended, err_msg=f"ended, ply {ply}")
        np.testing.assert_array_equal(dev.reward, orc.reward, err_msg=f"reward, ply {ply}")
    np.testing.assert_array_equal(legal.cpu().numpy(), orc.legal())
    np.testing.assert_array_equal(dev.grid, orc.grid)
    assert dev.steps == int(orc.plies.sum())
    with pytest.raises(TypeError):
        dev.step_actions_observe(cols.to(torch.int64), legal)
    with pytest.raises(TypeError):
        dev.step_actions_observe(cols, legal[:, :-1])
    # the optional outputs are optional, and the observation is allocated when none is passed
    dev.reset()
    orc.reset()
    cols = torch.full((n,), w // 2, dtype=torch.int32, device="cuda")
    fresh = dev.step_actions_observe(cols)
    orc.step_actions(cols.cpu().numpy())
    np.testing.assert_array_equal(fresh.cpu().numpy(), orc.legal())
    dev.close()


@pytest.mark.parametrize("geometry,n", [((6, 7, 4), 1 << 16), ((6, 7, 4), 4097), ((5, 4, 3), 2000), ((12, 13, 5), 1024)])
def test_vector_environment_step_with_auto_reset_connect(bm, torch_mod, geometry, n):
    """bgs_env_step: the moves, the reward pairs, the ended flags, the restart of finished boards and the next legal mask in
    one call (one kernel on one-word boards with an even batch, the separate kernels otherwise).  Against the oracle with the
    restart done by hand on its arrays: 3 games' worth of plies, so every board is restarted several times; without
    auto_reset the call is step_actions_observe + rewards."""
    torch = torch_mod
    h, w, k = geometry
    dev = bm.ConnectBatch(h, w, k, n, use_torch=True)
    orc = oracle.ConnectOracle(h, w, k, n)
    gen = torch.Generator(device="cuda").manual_seed(23)
    legal = dev.legal_tensor()
    ended = torch.zeros(n, dtype=torch.uint8, device="cuda")
    reward = torch.zeros((n, 2), dtype=torch.int8, device="cuda")
    status = torch.zeros(n, dtype=torch.int32, device="cuda")
    episodes = 0
    for ply in range(3 * h * w):
        np.testing.assert_array_equal(legal.cpu().numpy(), orc.legal(), err_msg=f"legal before ply {ply}")
        cols = (torch.rand((n, w), device="cuda", generator=gen) + legal.float()).argmax(dim=1).to(torch.int32)
        if ply % 7 == 3:
            cols[::11] = -1    # skipped boards
        want = orc.step_actions(cols.cpu().numpy())
        dev.env_step(cols, legal, ended=ended, reward=reward, status=status)
        np.testing.assert_array_equal(status.cpu().numpy(), want, err_msg=f"status, ply {ply}")
        done = orc.ended.copy()
        np.testing.assert_array_equal(ended.cpu().numpy().astype(bool), done, err_msg=f"ended, ply {ply}")
        np.testing.assert_array_equal(reward.cpu().numpy(), orc.reward, err_msg=f"reward, ply {ply}")
        episodes += int(done.sum())
        # the restart, by hand: Config::sample_initial_state for the boards that have ended
        orc.grid[done] = -1
        orc.player[done] = 0
        orc.winner[done] = -1
        orc.plies[done] = 0
        np.testing.assert_array_equal(dev.grid, orc.grid, err_msg=f"grid, ply {ply}")
        assert not dev.has_ended.any()
    assert episodes > 2 * n
    # without the restart: boards stay ended, the rewards are the batch's
    dev.reset()
    orc.reset()
    for ply in range(h * w + 1):
        cols = (torch.rand((n, w), device="cuda", generator=gen) + legal.float()).argmax(dim=1).to(torch.int32)
        orc.step_actions(cols.cpu().numpy())
        dev.env_step(cols, legal, ended=ended, reward=reward, auto_reset=False)
    assert orc.ended.all()
    np.testing.assert_array_equal(ended.cpu().numpy().astype(bool), orc.ended)
    np.testing.assert_array_equal(reward.cpu().numpy(), orc.reward)
    np.testing.assert_array_equal(dev.grid, orc.grid)
    dev.close()


def test_vector_environment_step_with_auto_reset_bounce(bm, torch_mod):
    """The same on Bounce (target masks as the observation; the separate kernels back to back): against the oracle's
    step_actions with the restart done by hand on its arrays."""
    torch = torch_mod
    n = 1024
    dev = bm.BounceBatch(DEFAULT_BOUNCE, n, use_torch=True)
    orc = oracle.BounceOracle(DEFAULT_BOUNCE, n)
    start = orc.grid.copy()
    targets = dev.targets_tensor()
    ended = torch.zeros(n, dtype=torch.uint8, device="cuda")
    reward = torch.zeros((n, 2), dtype=torch.int8, device="cuda")
    rng = np.random.default_rng(5)
    episodes = 0
    for ply in range(90):
        t = targets.cpu().numpy().view(np.uint64)
        moves = np.full((n, 4), -1, dtype=np.int32)
        for i in range(n):
            row = int(t[i, 6]) if t[i, 6] < 64 else -1
            cols = [x for x in range(6) if t[i, x]] if row >= 0 else []
            if cols:
                x = cols[rng.integers(len(cols))]
                cells = [c for c in range(54) if (int(t[i, x]) >> c) & 1]
                cell = cells[rng.integers(len(cells))]
                moves[i] = (x, row, cell % 6, cell // 6)
        want = orc.step_actions(moves)
        assert (want == 0).all()
        dev.env_step(torch.from_numpy(moves).cuda(), targets, ended=ended, reward=reward)
        done = orc.ended.copy()
        np.testing.assert_array_equal(ended.cpu().numpy().astype(bool), done, err_msg=f"ended, ply {ply}")
        np.testing.assert_array_equal(reward.cpu().numpy(), orc.reward, err_msg=f"reward, ply {ply}")
        episodes += int(done.sum())
        orc.grid[done] = start[done]
        orc.player[done] = 0
        orc.winner[done] = -1
        orc.plies[done] = 0
        np.testing.assert_array_equal(dev.grid, orc.grid, err_msg=f"grid, ply {ply}")
    assert episodes > n
    dev.close()


def test_one_call_per_policy_ply_bounce(bm, torch_mod):
    """The same call on Bounce (moves int32[n, 4] -> target masks uint64[n, W + 1] + ended): the kernels of the separate
    calls back to back; compared with step_actions + the 't' export and with the oracle's rewards."""
    torch = torch_mod
    n = 2048
    dev = bm.BounceBatch(DEFAULT_BOUNCE, n, use_torch=True)
    ref = bm.BounceBatch(DEFAULT_BOUNCE, n, use_torch=True)
    targets = dev.targets_tensor()
    ended = torch.zeros((n,), dtype=torch.uint8, device="cuda")
    rng = np.random.default_rng(3)
    for ply in range(40):
        t = targets.cpu().numpy().view(np.uint64)
        moves = np.full((n, 4), -1, dtype=np.int32)
        for i in range(n):   # the first target of a random movable column (host-side policy: this is a test of the call)
            row = int(t[i, 6]) if t[i, 6] < 64 else -1
            cols = [x for x in range(6) if t[i, x]] if row >= 0 else []
            if cols:
                x = cols[rng.integers(len(cols))]
                cell = int(t[i, x]).bit_length() - 1
                moves[i] = (x, row, cell % 6, cell // 6)
        mv = torch.from_numpy(moves).cuda()
        dev.step_actions_observe(mv, targets, ended=ended)
        ref.step_actions(mv, want_status=False)
        np.testing.assert_array_equal(targets.cpu().numpy(), ref.targets_tensor().cpu().numpy(), err_msg=f"targets, ply {ply}")
        np.testing.assert_array_equal(ended.cpu().numpy().astype(bool), ref.has_ended, err_msg=f"ended, ply {ply}")
    np.testing.assert_array_equal(dev.grid, ref.grid)
    np.testing.assert_array_equal(dev.reward, ref.reward)
    assert ref.has_ended.any()
    dev.close()
    ref.close()


def test_bounce_device_targets_and_moves(bm, torch_mod):
    """Config 4's geometry: target masks ('t'), action counts ('c') and rewards ('r') exported to device memory,
    moves chosen on the device from the masks, against the oracle's action lists."""
    torch = torch_mod
    n = 1 << 12
    dev = bm.BounceBatch(DEFAULT_BOUNCE, n, use_torch=True)
    orc = oracle.BounceOracle(DEFAULT_BOUNCE, n)
    w = 6
    targets = torch.empty((n, w + 1), dtype=torch.int64, device="cuda")
    rng = np.random.default_rng(8)
    for ply in range(40):
        dev._export("t", targets)
        t_host = targets.cpu().numpy().view(np.uint64)
        np.testing.assert_array_equal(t_host, dev.targets)
        counts = dev.action_count_tensor().cpu().numpy()
        np.testing.assert_array_equal(counts, orc.count_actions(), err_msg=f"ply {ply}")
        # pick the k-th action of the canonical list per board from the device masks (sources ascending x, targets
        # ascending cell index) and check a sample of boards against the oracle's explicit lists
        moves = np.full((n, 4), -1, dtype=np.int32)
        pick = (rng.random(n) * np.maximum(counts, 1)).astype(np.int64)
        for i in np.flatnonzero(counts > 0):
            row, k = int(t_host[i, w]), int(pick[i])
            for x in range(w):
                m = int(t_host[i, x])
                c = bin(m).count("1")
                if k < c:
                    cells = [b for b in range(54) if (m >> b) & 1]
                    moves[i] = (x, row, cells[k] % w, cells[k] // w)
                    break
                k -= c
        for i in rng.integers(0, n, size=16):
            if counts[i]:
                src, dst = orc.actions(int(i))[int(pick[i])]
                assert tuple(moves[i]) == (*src, *dst)
        st_dev = dev.step_actions(torch.from_numpy(moves).cuda())
        st_orc = orc.step_actions(moves)
        np.testing.assert_array_equal(st_dev, st_orc)
        np.testing.assert_array_equal(dev.reward_copy_tensor().cpu().numpy(), orc.reward)
        np.testing.assert_array_equal(dev.grid_tensor().cpu().numpy(), orc.grid)
    assert orc.ended.any()
    dev.close()


def test_device_views_and_dlpack(bm, torch_mod):
    torch = torch_mod
    n = 5000
    dev = bm.ConnectBatch(6, 7, 4, n, use_torch=True)
    dev.rollout(SEED, from_initial=True)
    orc = oracle.ConnectOracle(6, 7, 4, n)
    orc.rollout(SEED)
    rt = dev.reward_tensor()  # zero-copy torch view of the batch's own reward buffer
    again = torch.from_dlpack(rt.__dlpack__())  # the standard hand-over from here: DLPack, no copy
    assert again.data_ptr() == rt.data_ptr() == dev.buffer(3)[0]
    np.testing.assert_array_equal(again.cpu().numpy(), orc.reward)
    # without torch on the producing side: __cuda_array_interface__ views of the library's buffers
    for what, want in (("reward", orc.reward), ("status", np.where(orc.winner == 2, 3, orc.winner + 1).astype(np.uint8))):
        view = dev.device_view(what)
        t = torch.as_tensor(view, device="cuda")
        assert t.data_ptr() == view.__cuda_array_interface__["data"][0]
        np.testing.assert_array_equal(t.cpu().numpy(), want)
    planes = torch.as_tensor(dev.device_view("planes"), device="cuda")
    assert tuple(planes.shape) == (2, n)
    stones = np.array([bin(int(a) & (2**64 - 1)).count("1") + bin(int(b) & (2**64 - 1)).count("1")
                       for a, b in planes.cpu().numpy().T[:64]])
    np.testing.assert_array_equal(stones, orc.plies[:64])
    dev.close()


# ------------------------------------------------------------------------------------------------ N4: JSON states

def test_batch_json_states_against_the_reference_fixtures(bm, golden_dir):
    """Batch.from_json_states / to_json_states on the State JSON dicts the reference's own tests hold
    (tests/test_connect.py:131-138, tests/test_bounce.py:392-403)."""
    with open(os.path.join(golden_dir, "reference_connect.json")) as fh:
        ref = json.load(fh)
    state = ref["json"]["state"]
    cfg = ref["json"]["config"]
    dev = bm.ConnectBatch(cfg["height"], cfg["width"], cfg["count"], 3)
    assert (dev.from_json_states([state, state, state]) == 0).all()
    assert dev.to_json_states() == [state] * 3
    np.testing.assert_array_equal(dev.player, [state["player"]] * 3)
    dev.close()
    with open(os.path.join(golden_dir, "reference_bounce.json")) as fh:
        ref = json.load(fh)
    state = ref["json"]["state"]
    dev = bm.BounceBatch(np.array(ref["json"]["config"]["grid"], dtype=np.int8), 2)
    assert (dev.from_json_states([state, state]) == 0).all()
    assert dev.to_json_states() == [state, state]
    dev.close()


def test_batch_json_round_trip_of_played_positions(bm):
    n = 600
    dev = bm.ConnectBatch(6, 7, 4, n)
    dev.rollout(SEED, max_plies=23, from_initial=True)  # a mix of running and finished boards
    states = dev.to_json_states()
    assert any(s["winner"] == -1 for s in states) and any(s["winner"] in (0, 1) for s in states)
    other = bm.ConnectBatch(6, 7, 4, n)
    assert (other.from_json_states(states) == 0).all()
    assert other.to_json_states() == states
    np.testing.assert_array_equal(other.reward, dev.reward)
    dev.rollout(SEED)
    other.rollout(SEED)
    np.testing.assert_array_equal(other.grid, dev.grid)
    bdev = bm.BounceBatch(DEFAULT_BOUNCE, 200)
    bdev.rollout(SEED, max_plies=12, from_initial=True)
    states = bdev.to_json_states()
    bother = bm.BounceBatch(DEFAULT_BOUNCE, 200)
    assert (bother.from_json_states(states) == 0).all()
    assert bother.to_json_states() == states
    with pytest.raises(TypeError):
        bother.from_json_states(states[:3])
    with pytest.raises(RuntimeError):
        bother.from_json_states([{"grid": 1}] * 200)
    for b in (dev, other, bdev, bother):
        b.close()


# ------------------------------------------------------------------------------------------------ loader validation

def test_connect_loader_cross_checks_the_winner_against_the_grid(bm):
    """A declared winner has to be the one the grid implies: winner = -1 on a board that holds a
    k-run, a run for the side that did not move last, or a wrong winner are refused and leave the board untouched."""
    dev = bm.ConnectBatch(6, 7, 4, 6)
    e = -1
    won = np.full((6, 7), e, dtype=np.int8)  # player 0 has four in the bottom row, player 1 three stones
    won[0, :4] = 0
    won[1, :3] = 1
    running = np.full((6, 7), e, dtype=np.int8)
    running[0, :3] = 0
    running[1, :3] = 1
    late = won.copy()  # ... and player 1 moved once more AFTER the run was complete
    late[1, 3] = 1
    grids = np.stack([won, won, won, running, running, late])
    winner = np.array([0, -1, 1, -1, 0, 0], dtype=np.int8)
    player = np.array([1, 1, 1, 0, 0, 0], dtype=np.int8)
    status = dev.write_state(grids, player, winner)
    np.testing.assert_array_equal(status, [0, -1, -1, 0, -1, -1])
    np.testing.assert_array_equal(dev.winner, [0, -1, -1, -1, -1, -1])
    np.testing.assert_array_equal(dev.reward[0], [1, -1])
    # derived winners (winner=None) agree with the oracle's verdict on played-out boards
    big = bm.ConnectBatch(6, 7, 4, 4000)
    orc = oracle.ConnectOracle(6, 7, 4, 4000)
    orc.rollout(SEED)
    assert (big.write_state(orc.grid) == 0).all()
    np.testing.assert_array_equal(big.winner, orc.winner)
    np.testing.assert_array_equal(big.reward, orc.reward)
    # the fused kernels and the per-ply kernel agree on what a loaded board is: nothing moves on finished boards
    big.rollout(SEED)
    np.testing.assert_array_equal(big.grid, orc.grid)
    assert big.steps == 0
    dev.close()
    big.close()


def test_bounce_plies_saturate_instead_of_wrapping(bm):
    """Plies are stored as uint16: a board at 65535 plies is not stepped any further."""
    n = 4
    dev = bm.BounceBatch(DEFAULT_BOUNCE, n)
    grid = np.repeat(DEFAULT_BOUNCE[None], n, axis=0)
    plies = np.array([65535, 65534, 65533, 0], dtype=np.int32)
    player = (plies & 1).astype(np.int8)
    assert (dev.write_state(grid, player, None, plies) == 0).all()
    for _ in range(3):
        dev.step_random(SEED)
    np.testing.assert_array_equal(dev.plies, [65535, 65535, 65535, 3])
    moves = np.array([[0, 7, 0, 6]] * n, dtype=np.int32)  # a legal move for player 1 on the untouched board 0
    assert dev.step_actions(moves)[0] == -2
    np.testing.assert_array_equal(dev.grid[0], DEFAULT_BOUNCE)
    dev.close()


def test_object_api_reward_comes_from_the_device(bm):
    from simulator.game.connect import Config

    s = Config(2, 3, 2).sample_initial_state()
    np.testing.assert_array_equal(s.reward, [0, 0])
    for col in (1, 1, 2):
        s = s.action_at(col).sample_next_state()
    assert s.has_ended
    np.testing.assert_array_equal(s.reward, [1, -1])  # tests/test_connect.py:115
    assert s.reward.dtype == np.int8 and s.reward is not s.reward


# ------------------------------------------------------------------------------------------------ bench.py, N = 2

def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("gather,ranks", [("shm", 3), ("rccl", 2), ("both", 2), ("both", 4)])
def test_bench_multi_rank_rehearsal(gather, ranks):
    """bench.py's own N > 1 loops as child processes sharing this box's GPU (gloo carries the control messages; RCCL
    cannot run several ranks on one GPU): `shm` -- one host array in shared memory, every rank's sink delivers its rows,
    rank 0 consumes (futex hand-shake inside the native loop); `rccl` -- the IN-LIBRARY gather (bgs_gather_*, the code a
    real 8-GPU run executes) over the tests' shared-memory stand-in for RCCL; `both` -- the default of an N > 1 run: the
    two one after the other, `value` from the RCCL gather (the north-star's collective; round 6), `rccl_ranks` = what the
    communicator itself reports.  Rank 0's host array must verify against a replay of EVERY
    rank's first games.  World sizes 2 and 4 of the metric's 1 / 2 / 4 / 8: the box allows at most 6 processes on the card and
    this test process is one of them (tools/r5_dist.sh runs 6 ranks outside pytest); the gather's N = 8 arithmetic runs as 4
    processes x 2 ranks in tests/test_gpu_gather_peers.py, the shared array's 8-rank hand-shake on the CPU in
    tests/test_sharding_gloo.py."""
    from tests.test_gpu_gather_peers import build_fake_rccl

    port = _free_port()
    procs = []
    for rank in range(ranks):
        env = product_env( RANK=str(rank), WORLD_SIZE=str(ranks), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), BGS_DIST_BACKEND="gloo", OMP_NUM_THREADS="4", BGS_RCCL_LIB=build_fake_rccl())
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(ranks), "--steps", "25", "--warmup", "2", "--batch",
               str(1 << 16), "--no-cpu-baseline", "--gather", gather, "--host-threads", "2"]
        procs.append(subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=600) for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(o[1][-2000:] for o in outs)
    line = [ln for ln in outs[0][0].splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == ranks and d["config"]["gathered_rewards_verified"] is True
    assert d["config"]["rewards_to_host"] is True and d["config"]["global_batch"] == ranks << 16
    assert ("shared memory" in d["config"]["sharding"]) == (gather == "shm")
    assert d["config"]["gather"] == ("shm" if gather == "shm" else "rccl")
    assert d["rccl_ranks"] == (None if gather == "shm" else ranks)
    assert 0 < d["ms_per_step_fastest_rank"] <= d["ms_per_step_slowest_rank"] and "failed_handovers" not in d
    assert d["ms_per_step_slowest_rank"] == pytest.approx(d["ms_per_step"])
    if gather == "both":
        assert d["value"] == pytest.approx(d["gather_rccl"]["value"]) and d["config"]["gathers_measured"] == ["shm", "rccl"]
    for kind in (("shm", "rccl") if gather == "both" else (gather,)):
        blk = d[f"gather_{kind}"]
        assert blk["gathered_rewards_verified"] is True and blk["value"] > 0 and len(blk["values_of_3"]) == 3
    if gather != "shm":
        info = d["gather_rccl"]["gather_info"]
        assert info["direct"] is False and info["batch"] == 6 and info["transport_check"] == "passed"
        assert info["ranks"] == ranks and info["rank"] == 0   # what the COMMUNICATOR says, not what the launcher asked for
        assert "libfake_rccl" in info["transport"] and "RCCL gather inside the library" in d["gather_rccl"]["sharding"]
    assert not [ln for ln in outs[1][0].splitlines() if ln.startswith("{")]  # only rank 0 prints the line


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 3` with no launcher around it: the script starts its three ranks as child processes (before
    it touches the GPU itself), relays rank 0's line and exits 0.  (gloo: the ranks share this box's one GPU.)"""
    from tests.test_gpu_gather_peers import build_fake_rccl

    env = product_env( BGS_DIST_BACKEND="gloo", OMP_NUM_THREADS="4", BGS_RCCL_LIB=build_fake_rccl())
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    proc = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--steps", "20", "--warmup", "2",
                           "--batch", str(1 << 16), "--host-threads", "2"], env=env, capture_output=True, text=True, timeout=900)
    assert proc.returncode == 0, proc.stderr[-3000:]
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 3 and d["config"]["gathered_rewards_verified"] is True and d["config"]["gather"] == "rccl"
    assert d["rccl_ranks"] == 3 and d["value"] == pytest.approx(d["gather_rccl"]["value"])
    # the default N > 1 run measures both hand-overs, each verified on its own
    assert d["config"]["gathers_measured"] == ["shm", "rccl"]
    assert d["gather_shm"]["gathered_rewards_verified"] is True and d["gather_rccl"]["gathered_rewards_verified"] is True
    assert d["config"]["global_batch"] == 3 << 16 and d["value"] > 0 and d["steps"] == 20
    assert "cpu_baseline" not in d  # rank 0 at N = 1 only


def test_bench_fails_loudly_when_the_gather_dies():
    """Round-5 review: a driver that reads the exit code must not see success with the north-star collective dead.  Two
    ranks over the stand-in, rank 1 goes silent after its first messages (BGS_FAKE_RCCL_MUTE_AFTER: its sends are dropped,
    no error anywhere): rank 0's gather never completes, the watchdog gives it up after BGS_BENCH_GATHER_TIMEOUT -- the line
    is still printed, with what the shared array measured and the gather's error, and the exit code is NOT 0."""
    from tests.test_gpu_gather_peers import build_fake_rccl

    env = product_env( BGS_DIST_BACKEND="gloo", OMP_NUM_THREADS="4", BGS_RCCL_LIB=build_fake_rccl(), BGS_FAKE_RCCL_MUTE_AFTER="1:3",
               BGS_BENCH_GATHER_TIMEOUT="25", BGS_FAKE_RCCL_TIMEOUT_MS="120000")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    proc = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "2",
                           "--batch", str(1 << 16), "--host-threads", "2"], env=env, capture_output=True, text=True, timeout=900)
    assert proc.returncode != 0, proc.stderr[-3000:]
    assert "hand-over did not finish within 25 s" in proc.stderr
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, proc.stderr[-3000:]
    d = json.loads(lines[0])
    assert d["failed_handovers"] == ["rccl"] and "error" in d["gather_rccl"] and d["rccl_ranks"] is None
    assert d["config"]["gather"] == "shm" and d["gather_shm"]["gathered_rewards_verified"] is True and d["value"] > 0


@pytest.mark.parametrize("gather", ["rccl", "shm"])
def test_bench_sharded_path_with_one_rank_over_rccl(gather):
    """The N > 1 loops over the REAL collective backend (nccl = RCCL) with a world of one rank -- what a one-GPU box can
    run of them: `rccl` = the in-library gather (bgs_gather_*: ncclCommInitRank, communication thread, send / receive
    group per step, rank 0's sink), `shm` = the shared host array with the consumer hand-shake (RCCL then only carries
    the barriers and the step count); verified host array."""
    env = product_env( RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
               BGS_FORCE_DIST="1")
    env.pop("BGS_DIST_BACKEND", None)
    proc = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "30", "--warmup", "3",
                           "--batch", str(1 << 18), "--no-cpu-baseline", "--gather", gather], env=env, capture_output=True,
                          text=True, timeout=900)
    assert proc.returncode == 0, proc.stderr[-3000:]
    d = json.loads([ln for ln in proc.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["config"]["gathered_rewards_verified"] is True and d["config"]["gather"] == gather
    assert ("RCCL gather inside the library" in d["config"]["sharding"]) == (gather == "rccl")
    assert d["config"]["rewards_to_host"] is True and d["value"] > 0
    assert d["config"]["loop"].startswith("native")


def test_bench_single_gpu_line():
    """The default hand-over on one GPU at a small batch: contract fields, host rewards verified against the oracle,
    the other BASELINE configs measured and parity-checked by their child processes."""
    proc = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "12", "--warmup", "3", "--batch",
                           str(1 << 16)], env=product_env(), capture_output=True, text=True, timeout=900)
    assert proc.returncode == 0, proc.stderr[-2000:]
    d = json.loads([ln for ln in proc.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["config"]["rewards_to_host"] is True and d["cpu_baseline"]["parity_with_host_rewards"] is True
    assert d["cpu_baseline"]["single_game_latency_us"] > 0 and d["cpu_baseline"]["gpu_single_game_latency_us"] > 0
    roof = d["roofline"]
    assert roof["bound"] == "valu_issue" and roof["unit"] == "Ginstr/s" and roof["peak"] == pytest.approx(1228.8)
    assert roof["frac"] is None and roof["achieved"] is None  # the committed counters are for 2^20 boards per launch
    assert roof["kernel_ms_per_launch"] > 0 and roof["hbm"]["frac"] > 0 and roof["pcie"]["frac"] <= 1
    assert d["device_resident"]["value"] > 0 and d["steps"] == 12 and d["warmup"] == 3
    assert len(d["values_of_3"]) == 3 and min(d["values_of_3"]) <= d["value_median_of_3"] <= max(d["values_of_3"])
    for name in ("connect_12x13x5", "bounce_default"):
        o = d["other_configs"][name]
        assert "error" not in o, o
        assert o["parity_with_oracle"] is True and o["value"] > 0 and o["solo"]["value"] > 0 and o["rewards_to_host"] is True


def test_rollout_pipeline_matches_the_oracle_step_by_step():
    """simulator.pipeline.RolloutPipeline (the bench's loop as a library object): every step's host rewards equal the
    oracle's for that step's seed, for Connect and for Bounce, whatever the depth; arrays are reused after 3 x depth."""
    from simulator.batch import BounceBatch, ConnectBatch
    from simulator.pipeline import RolloutPipeline

    n = 5000
    for depth in (1, 3):
        with RolloutPipeline(ConnectBatch, (6, 7, 4), n, depth=depth, host_threads=3, first_game=77) as pipe:
            seen = 0
            for step, rewards in pipe.run(seeds=[SEED + 11 * s for s in range(12)]):
                orc = oracle.ConnectOracle(6, 7, 4, n)
                orc.rollout(SEED + 11 * step, first_game=77)
                np.testing.assert_array_equal(rewards, orc.reward, err_msg=f"depth {depth} step {step}")
                seen += 1
            assert seen == 12 and pipe.env_steps > 12 * n * 7
            with pytest.raises(KeyError):
                pipe.result(0)  # 12 steps on 3 * depth <= 9 arrays: step 0's array has been reused
    grid = np.zeros((9, 6), dtype=np.int8)
    grid[1] = grid[7] = [1, 2, 3, 3, 2, 1]
    with RolloutPipeline(BounceBatch, (grid,), 700, depth=2, host_threads=2, max_plies=2000) as pipe:
        for step, rewards in pipe.run(seeds=range(5)):
            orc = oracle.BounceOracle(grid, 700)
            orc.rollout(step, max_plies=2000)
            np.testing.assert_array_equal(rewards, orc.reward, err_msg=f"bounce step {step}")


def test_rollout_pipeline_feeder_survives_a_consumer_that_stops_early():
    """`RolloutPipeline.run` feeds its seeds to the native loop's own thread (bgs_pipeline_feed / _release): a consumer
    that breaks out of the loop after a few steps, starts a new loop, mixes in `submit` / `result`, and a slow consumer
    that holds every array for a while -- every step it sees equals the oracle's for that seed."""
    import time

    from simulator.batch import ConnectBatch
    from simulator.pipeline import RolloutPipeline

    n = 3000

    def want(seed):
        orc = oracle.ConnectOracle(6, 7, 4, n)
        orc.rollout(seed)
        return orc.reward

    with RolloutPipeline(ConnectBatch, (6, 7, 4), n, depth=2, host_threads=2) as pipe:
        seen = []
        for step, rewards in pipe.run(seeds=range(100, 140)):
            np.testing.assert_array_equal(rewards, want(100 + step))
            seen.append(step)
            if len(seen) == 3:
                break                      # 37 fed steps are still played and delivered, nobody reads them
        assert seen == [0, 1, 2]
        first = pipe._next                 # (steps fed so far: the next loop's indices go on from here)
        for step, rewards in pipe.run(seeds=[7, 8, 9, 10, 11, 12, 13]):
            time.sleep(0.002)              # a slow consumer: the feeder must not overwrite the array it is looking at
            np.testing.assert_array_equal(rewards, want(7 + step - first), err_msg=f"second loop, step {step}")
        i = pipe.submit(555)               # one step at a time still works beside it
        np.testing.assert_array_equal(pipe.result(i), want(555))
        assert sum(1 for _ in pipe.run(seeds=[])) == 0


def test_policy_loop_replays_from_a_hip_graph(bm, torch_mod):
    """Policy-driven stepping (N2) captured once and replayed: legal mask on the device -> a torch policy -> bgs_step_actions
    with device actions are plain enqueues on the batch's stream (no allocation, no synchronisation), so torch.cuda.graphs
    can record a whole game's plies and replay them without per-launch host cost (tools/policy_loop.py: 1.8x at 2^16
    boards).  A deterministic policy (first legal column, shifted by the board index) so that eager, graph and oracle
    must agree board for board."""
    torch = torch_mod
    n, plies = 3000, 42
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        dev = bm.ConnectBatch(6, 7, 4, n, use_torch=True)
        legal = torch.empty((n, 7), dtype=torch.uint8, device="cuda")
        shift = (torch.arange(n, device="cuda") % 7).to(torch.int64)
        cols = torch.arange(7, device="cuda").unsqueeze(0)

        def one_ply():
            dev.legal_tensor(legal)
            # the first legal column at or after `shift` (cyclically); -1 when the board has ended
            order = (cols - shift.unsqueeze(1)) % 7
            score = torch.where(legal.bool(), order, torch.full_like(order, 99))
            col = score.argmin(dim=1)
            col = torch.where(legal.bool().any(dim=1), col, torch.full_like(col, -1)).to(torch.int32)
            dev.step_actions(col, want_status=False)

        for _ in range(plies):
            one_ply()
        torch.cuda.synchronize()
        eager_grid, eager_reward, eager_steps = dev.grid, dev.reward, dev.steps
        assert dev.has_ended.all()

        graph = torch.cuda.CUDAGraph()
        dev.reset()
        torch.cuda.synchronize()
        with torch.cuda.graph(graph, stream=stream):
            for _ in range(plies):
                one_ply()
        for _ in range(2):  # replay twice: the graph holds no state of its own
            dev.reset()
            graph.replay()
            torch.cuda.synchronize()
            np.testing.assert_array_equal(dev.grid, eager_grid)
            np.testing.assert_array_equal(dev.reward, eager_reward)
            assert dev.steps == eager_steps
    # the same policy on the oracle
    orc = oracle.ConnectOracle(6, 7, 4, n)
    sh = np.arange(n) % 7
    for _ in range(plies):
        lg = orc.legal().astype(bool)
        order = (np.arange(7)[None, :] - sh[:, None]) % 7
        col = np.where(lg, order, 99).argmin(axis=1)
        col = np.where(lg.any(axis=1), col, -1).astype(np.int32)
        orc.step_actions(col)
    np.testing.assert_array_equal(eager_grid, orc.grid)
    np.testing.assert_array_equal(eager_reward, orc.reward)


def test_bench_under_the_strict_rng_contract():
    """`bench.py --rng per-ply`: `value` under the strict contract (config.rng says so, the CPU baseline's parity check runs the
    oracle under the same contract), the default contract's rate beside it (`rng_other`), both hand-overs checked against the
    oracle."""
    proc = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "12", "--warmup", "3", "--batch", str(1 << 16),
                           "--rng", "per-ply", "--no-other-configs"], env=product_env(), capture_output=True, text=True, timeout=900)
    assert proc.returncode == 0, proc.stderr[-2000:]
    d = json.loads([ln for ln in proc.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["config"]["rng"].startswith("per-ply") and d["cpu_baseline"]["parity_with_host_rewards"] is True
    other = d["rng_other"]
    assert other["rng"].startswith("per-block") and other["parity_with_oracle_first_4096"] is True and other["value"] > 0
    assert d["value"] > 0 and d["rccl_ranks"] is None

"""HIP path (through the C ABI) against the CPU oracle on the same seeds -- bit-exact, integer work.

Everything here needs a real MI355X: `pytest -m gpu`.
"""

import numpy as np
import pytest

from tests.knobs import knobs

from oracle import oracle

pytestmark = pytest.mark.gpu

SEED = 0x0123456789ABCDEF

DEFAULT_BOUNCE = np.zeros((9, 6), dtype=np.int8)
DEFAULT_BOUNCE[1] = DEFAULT_BOUNCE[7] = [1, 2, 3, 3, 2, 1]


@pytest.fixture(scope="module")
def batch_mod():
    from simulator import batch

    return batch


def assert_same(dev, orc, what=""):
    np.testing.assert_array_equal(dev.grid, orc.grid, err_msg=f"grid {what}")
    np.testing.assert_array_equal(dev.winner, orc.winner, err_msg=f"winner {what}")
    np.testing.assert_array_equal(dev.has_ended, orc.ended, err_msg=f"ended {what}")
    np.testing.assert_array_equal(dev.plies, orc.plies, err_msg=f"plies {what}")
    np.testing.assert_array_equal(dev.player, orc.player, err_msg=f"player {what}")
    np.testing.assert_array_equal(dev.reward, orc.reward, err_msg=f"reward {what}")


# ------------------------------------------------------------------------------------------------ Connect

CONNECT_GEOMETRIES = [
    (6, 7, 4),     # static instantiation, 1 word
    (12, 13, 5),   # static instantiation, 3 words
    (2, 3, 2),     # the reference's own test board
    (4, 5, 3),
    (1, 6, 2),     # single row
    (5, 1, 3),     # single column
    (6, 7, 1),     # any stone wins
    (8, 9, 4),     # 2 words, run-time geometry
    (15, 12, 5),   # 192 bits, 3 words, run-time geometry
    (3, 16, 4),    # widest supported
    (6, 7, 9),     # count larger than the board: draws only
]


@pytest.mark.parametrize("h,w,k", CONNECT_GEOMETRIES)
def test_connect_step_random_lockstep(batch_mod, h, w, k):
    n = 1500
    dev = batch_mod.ConnectBatch(h, w, k, n)
    orc = oracle.ConnectOracle(h, w, k, n)
    dev.set_first_game(77)
    assert_same(dev, orc, "after reset")
    np.testing.assert_array_equal(dev.legal, orc.legal())
    total = 0
    for ply in range(h * w + 1):
        total += orc.step_random(SEED, first_game=77)
        dev.step_random(SEED)
        assert_same(dev, orc, f"ply {ply}")
        if ply % 5 == 0:
            np.testing.assert_array_equal(dev.legal, orc.legal())
            np.testing.assert_array_equal(dev.action_count, orc.legal().sum(axis=1))
        assert dev.steps == total
    assert orc.ended.all()


@pytest.mark.parametrize("h,w,k,n", [(6, 7, 4, 5000), (6, 7, 4, 4999), (2, 3, 2, 64), (4, 5, 3, 1002), (8, 8, 4, 770),
                                     (7, 8, 5, 300), (12, 13, 5, 500), (1, 6, 2, 130),
                                     # the streaming kernel finds the sampled column by arithmetic on the column fields when the
                                     # column count fits under a field's top bit (w <= 2^h, h <= 15), else by the general search:
                                     (3, 8, 3, 600), (2, 5, 2, 600), (2, 4, 2, 600), (15, 4, 4, 400), (31, 2, 4, 200), (63, 1, 4, 200),
                                     (7, 7, 4, 1000), (4, 12, 4, 1000)])
def test_connect_step_random_n(batch_mod, h, w, k, n):
    """bgs_step_random_n == that many bgs_step_random calls: the streaming kernel (one-word boards, even n: pairs of
    boards per lane, several plies per launch, philox block boundaries inside a launch) and the fallbacks (odd n,
    multi-word boards)."""
    dev = batch_mod.ConnectBatch(h, w, k, n)
    orc = oracle.ConnectOracle(h, w, k, n)
    dev.set_first_game(12345)
    total = 0
    for plies in (1, 3, 4, 2, 5, 7, 1, 64, 100):
        dev.step_random(SEED ^ 77, plies=plies)
        for _ in range(plies):
            total += orc.step_random(SEED ^ 77, first_game=12345)
        assert_same(dev, orc, f"after {plies} more plies")
        assert dev.steps == total
    assert orc.ended.all()
    dev.step_random(SEED, plies=0)  # nothing to do
    assert dev.steps == total


@pytest.mark.parametrize("h,w,k", CONNECT_GEOMETRIES)
@pytest.mark.parametrize("from_initial", [False, True])
def test_connect_rollout(batch_mod, h, w, k, from_initial):
    n = 20000 if h * w <= 64 else 6000
    dev = batch_mod.ConnectBatch(h, w, k, n)
    orc = oracle.ConnectOracle(h, w, k, n)
    dev.set_first_game(1 << 33)
    if from_initial:
        dev.step_random(SEED)  # dirty the boards: FROM_INITIAL must ignore them
        dev.reset_steps()
    dev.rollout(SEED, from_initial=from_initial)
    total = orc.rollout(SEED, first_game=1 << 33)
    assert_same(dev, orc)
    assert dev.steps == total == int(orc.plies.sum())
    assert dev.has_ended.all()
    np.testing.assert_array_equal(dev.legal, np.zeros((n, w), dtype=np.uint8))


def test_connect_rollout_from_mid_game_and_max_plies(batch_mod):
    n = 9000
    dev = batch_mod.ConnectBatch(6, 7, 4, n)
    orc = oracle.ConnectOracle(6, 7, 4, n)
    for _ in range(9):
        dev.step_random(SEED ^ 5)
        orc.step_random(SEED ^ 5)
    dev.rollout(SEED ^ 5, max_plies=17)
    orc.rollout(SEED ^ 5, max_plies=17)
    assert_same(dev, orc, "capped at 17 plies")
    assert orc.plies.max() == 17 and not orc.ended.all()
    dev.rollout(SEED ^ 5, max_plies=0)
    assert_same(dev, orc, "max_plies=0 is a no-op")
    dev.rollout(SEED ^ 5)
    orc.rollout(SEED ^ 5)
    assert_same(dev, orc, "finished")


def test_connect_step_actions_legal_illegal_skip(batch_mod):
    n = 4096
    rng = np.random.default_rng(3)
    dev = batch_mod.ConnectBatch(6, 7, 4, n)
    orc = oracle.ConnectOracle(6, 7, 4, n)
    for _ in range(50):
        cols = rng.integers(-2, 9, size=n).astype(np.int32)  # negatives skip, 7 and 8 are out of range
        st_dev = dev.step_actions(cols)
        st_orc = orc.step_actions(cols)
        np.testing.assert_array_equal(st_dev, st_orc)
        assert_same(dev, orc)
    assert (st_orc == -2).any() and orc.ended.any()
    assert dev.steps == int(orc.plies.sum())


@pytest.mark.parametrize("h,w,k", [(6, 7, 4), (12, 13, 5), (8, 9, 4)])
def test_connect_write_state_roundtrip(batch_mod, h, w, k):
    n = 3000
    orc = oracle.ConnectOracle(h, w, k, n)
    for i in range(h * w // 2):
        orc.step_random(SEED + 1)
    dev = batch_mod.ConnectBatch(h, w, k, n)
    # winner derived on the device from the grid alone
    status = dev.write_state(orc.grid)
    assert (status == 0).all()
    assert_same(dev, orc, "loaded, winner derived")
    # winner and player given explicitly
    dev.reset()
    assert (dev.write_state(orc.grid, orc.player, orc.winner) == 0).all()
    assert_same(dev, orc, "loaded with winner")
    # continue the games from the loaded boards
    dev.rollout(SEED + 2)
    orc.rollout(SEED + 2)
    assert_same(dev, orc, "continued")


def test_connect_write_state_rejects_malformed(batch_mod):
    dev = batch_mod.ConnectBatch(6, 7, 4, 4)
    before = dev.grid
    g = np.full((4, 6, 7), -1, dtype=np.int8)
    g[0, 1, 0] = 0              # floating stone
    g[1, 0, 0] = 1              # player 1 moved first
    g[2, 0, 0] = 3              # bad cell code
    g[3, 0, 0] = 0              # fine
    status = dev.write_state(g)
    assert status.tolist() == [-1, -1, -1, 0]
    got = dev.grid
    np.testing.assert_array_equal(got[:3], before[:3])
    np.testing.assert_array_equal(got[3], g[3])


def test_connect_sharding_is_invisible(batch_mod):
    n, shards = 8192, 4
    whole = batch_mod.ConnectBatch(6, 7, 4, n)
    whole.rollout(SEED, from_initial=True)
    grid, reward = whole.grid, whole.reward
    per = n // shards
    for r in range(shards):
        part = batch_mod.ConnectBatch(6, 7, 4, per)
        part.set_first_game(r * per)
        part.rollout(SEED, from_initial=True)
        np.testing.assert_array_equal(part.grid, grid[r * per : (r + 1) * per])
        np.testing.assert_array_equal(part.reward, reward[r * per : (r + 1) * per])


def test_connect_full_size_batch(batch_mod):
    """BASELINE config 2: Connect4(6,7,4), batch 2^20 -- full comparison with the oracle plus size-free properties."""
    n = 1 << 20
    dev = batch_mod.ConnectBatch(6, 7, 4, n)
    dev.rollout(SEED, from_initial=True)
    steps = dev.steps
    grid, reward, winner, plies = dev.grid, dev.reward, dev.winner, dev.plies
    # properties that hold at any size
    assert dev.has_ended.all()
    assert steps == int(plies.sum())
    assert (reward.sum(axis=1) == 0).all()
    assert ((winner == 2) == (reward == 0).all(axis=1)).all()
    stones = (grid >= 0).sum(axis=(1, 2))
    np.testing.assert_array_equal(stones, plies)
    assert (((grid == 0).sum(axis=(1, 2)) - (grid == 1).sum(axis=(1, 2))) == (plies & 1)).all()
    assert plies.min() >= 7 and plies.max() <= 42
    # determinism + idempotence: same seed twice, and a rollout of finished boards changes nothing
    dev.rollout(SEED)
    np.testing.assert_array_equal(dev.grid, grid)
    assert dev.steps == steps
    dev.reset()
    dev.rollout(SEED)
    np.testing.assert_array_equal(dev.grid, grid)
    np.testing.assert_array_equal(dev.reward, reward)
    # and the oracle, all 2^20 games
    orc = oracle.ConnectOracle(6, 7, 4, n)
    total = orc.rollout(SEED)
    assert total == steps
    np.testing.assert_array_equal(grid, orc.grid)
    np.testing.assert_array_equal(reward, orc.reward)
    np.testing.assert_array_equal(plies, orc.plies)


@pytest.mark.parametrize("opening", ["3", "0"])
def test_connect_large_board_full_size(batch_mod, opening):
    """BASELINE config 3: Connect4(12,13,5), batch 2^18 -- with the opening launch (k_connect_open_lds: the first 8 plies
    of every game, then the LDS-staged kernel picks the boards up from memory) and without it."""
    import os

    old = knobs.get("rollout_opening")
    knobs["rollout_opening"] = opening
    try:
        n = 1 << 18
        dev = batch_mod.ConnectBatch(12, 13, 5, n)
        dev.set_first_game(3 << 20)
        dev.rollout(SEED, from_initial=True)
        orc = oracle.ConnectOracle(12, 13, 5, n)
        total = orc.rollout(SEED, first_game=3 << 20)
        assert dev.steps == total
        np.testing.assert_array_equal(dev.grid, orc.grid)
        np.testing.assert_array_equal(dev.reward, orc.reward)
        np.testing.assert_array_equal(dev.plies, orc.plies)
        dev.close()
    finally:
        if old is None:
            del knobs["rollout_opening"]
        else:
            knobs["rollout_opening"] = old


@pytest.mark.parametrize("kernel", ["lds", "registers"])
def test_connect_large_board_entry_states(batch_mod, kernel):
    """Connect(12,13,5) rollouts from every entry state on the LDS-staged kernel (K2c: from the initial state, boards
    loaded from memory at any ply of a 4-ply block, ply caps, finished boards in the batch, fused outcome codes) and on
    the register kernel it replaces (BGS_EXPERIMENT=rollout_no_lds): boards, rewards, plies, step counts against the oracle."""
    import os

    old = knobs.get("rollout_no_lds")
    if kernel == "registers":
        knobs["rollout_no_lds"] = "1"
    try:
        n = 20011
        dev = batch_mod.ConnectBatch(12, 13, 5, n)
        orc = oracle.ConnectOracle(12, 13, 5, n)
        dev.set_first_game(7 << 40)
        total = 0
        for lead in (1, 2, 3, 6):    # boards enter the rollout 1, 3, 6, 12 plies into the game: every ply & 3
            for _ in range(lead):
                dev.step_random(SEED ^ 21)
                total += orc.step_random(SEED ^ 21, first_game=7 << 40)
            cap = int(orc.plies.max()) + 9
            dev.rollout(SEED ^ 21, max_plies=cap)
            total += orc.rollout(SEED ^ 21, first_game=7 << 40, max_plies=cap)
            assert_same(dev, orc, f"{kernel}: loaded after {lead} more plies, capped at {cap}")
            assert dev.steps == total
        sink = batch_mod.RewardSink(n, slots=2, threads=2)
        host = np.zeros((n, 2), dtype=np.int8)
        sink.wait(sink.rollout(dev, host, SEED ^ 21, max_plies=60))     # from memory, capped, codes fused
        orc.rollout(SEED ^ 21, first_game=7 << 40, max_plies=60)
        assert_same(dev, orc, f"{kernel}: capped at 60, codes")
        np.testing.assert_array_equal(host, orc.reward)
        assert not orc.ended.all() and orc.ended.any()
        sink.wait(sink.rollout(dev, host, SEED ^ 21))                   # from memory to the end, finished boards inside
        orc.rollout(SEED ^ 21, first_game=7 << 40)
        assert_same(dev, orc, f"{kernel}: finished")
        np.testing.assert_array_equal(host, orc.reward)
        assert orc.ended.all() and dev.steps == int(orc.plies.sum())
        sink.wait(sink.rollout(dev, host, SEED ^ 22, max_plies=33, from_initial=True))   # from the initial state, capped
        orc.reset()
        orc.rollout(SEED ^ 22, first_game=7 << 40, max_plies=33)
        assert_same(dev, orc, f"{kernel}: from the initial state, capped at 33")
        np.testing.assert_array_equal(host, orc.reward)
        sink.close()
        dev.close()
    finally:
        if old is None:
            knobs.pop("rollout_no_lds", None)
        else:
            knobs["rollout_no_lds"] = old


@pytest.mark.parametrize("mode", ["opening0", "opening1", "opening2", "opening3", "opening4", "generic"])
def test_connect_rollout_kernel_families_agree_with_the_oracle(batch_mod, mode):
    """Every from-the-initial-state rollout kernel for one-word boards on the same batches: K2a (no opening stage), K2o
    with one and with two opening blocks, and the any-geometry kernel -- boards, rewards, step count and the fused
    outcome codes.  Geometries on both sides of the opening stage's conditions (H >= 4, W >= 2, K >= 3; two blocks need
    K >= 4 and H >= 6), ragged sizes, a chunk that ends inside a round of 64."""
    import os

    env = {"opening0": {"rollout_opening": "0"}, "opening1": {"rollout_opening": "1"},
           "opening2": {"rollout_opening": "2"}, "opening3": {"rollout_opening": "3"},
           "opening4": {"rollout_opening": "4"}, "generic": {"rollout_generic": "1"}}[mode]
    old = {k: knobs.get(k) for k in env}
    knobs.update(env)
    try:
        for (h, w, k), n in [((6, 7, 4), 70001), ((6, 7, 4), 63), ((4, 2, 3), 5000), ((5, 5, 3), 4097), ((8, 8, 4), 9999),
                             ((6, 2, 4), 3000), ((7, 8, 5), 8191), ((3, 7, 3), 2000), ((6, 1, 4), 500), ((6, 7, 2), 1000),
                             ((8, 3, 4), 2222)]:
            dev = batch_mod.ConnectBatch(h, w, k, n)
            orc = oracle.ConnectOracle(h, w, k, n)
            dev.set_first_game(1 << 33)
            dev.rollout(SEED ^ 9, from_initial=True)
            total = orc.rollout(SEED ^ 9, first_game=1 << 33)
            assert_same(dev, orc, f"{mode} {h}x{w}x{k} n={n}")
            assert dev.steps == total
            # the same launch with the outcome codes delivered by the kernel
            import torch

            packed = torch.zeros((n + 63) // 64 * 16, dtype=torch.uint8, device="cuda")
            dev.reset()
            dev.rollout_outcomes_tensor(packed, SEED ^ 10, from_initial=True)
            orc.reset()
            orc.rollout(SEED ^ 10, first_game=1 << 33)
            host = batch_mod.expand_outcomes_host(packed.cpu().numpy(), n)
            np.testing.assert_array_equal(host, orc.reward)
            assert_same(dev, orc, f"{mode} {h}x{w}x{k} n={n} with codes")
            dev.close()
    finally:
        for k, v in old.items():
            if v is None:
                del knobs[k]
            else:
                knobs[k] = v


@pytest.mark.parametrize("mode", ["1", "1:flat", "8", "1:nested", "passes"])
def test_bounce_rollout_kernel_families_agree_with_the_oracle(batch_mod, mode):
    """Every fused Bounce rollout kernel on the same mid-size batch: one lane per board on the piece list (K3p, default for
    large batches from the start position) or with the flattened cell search (K3f: boards loaded from memory), both with
    the work queue; 8 lanes per board (default for small batches), the nested-loop kernel and the multi-pass plan (flat
    passes, then a lane-group pass over the compacted work list)."""
    import os

    env = {"1": {"bounce_group": "1"}, "1:flat": {"bounce_group": "1", "bounce_pieces": "0"}, "8": {"bounce_group": "8"},
           "1:nested": {"bounce_group": "1", "bounce_flat": "0"},
           "passes": {"bounce_plan": "16:1,64:1,0:8", "bounce_chunk": "8"}}[mode]
    old = {k: knobs.get(k) for k in env}
    knobs.update(env)
    try:
        n = 9000
        dev = batch_mod.BounceBatch(DEFAULT_BOUNCE, n)
        orc = oracle.BounceOracle(DEFAULT_BOUNCE, n)
        dev.set_first_game(31)
        dev.rollout(SEED ^ 4, max_plies=700, from_initial=True)
        total = orc.rollout(SEED ^ 4, first_game=31, max_plies=700)
        assert_same(dev, orc, f"mode {mode}, from the initial state")
        assert dev.steps == total
        dev.reset()
        orc.reset()
        dev.reset_steps()
        dev.step_random(SEED ^ 5, plies=3)
        for _ in range(3):
            orc.step_random(SEED ^ 5, first_game=31)
        dev.rollout(SEED ^ 5, max_plies=40)
        orc.rollout(SEED ^ 5, first_game=31, max_plies=40)
        assert_same(dev, orc, f"mode {mode}, resumed and capped")
        dev.rollout(SEED ^ 5, max_plies=2000)
        orc.rollout(SEED ^ 5, first_game=31, max_plies=2000)
        assert_same(dev, orc, f"mode {mode}, finished")
        assert dev.steps == int(orc.plies.sum())
        dev.close()
    finally:
        for k, v in old.items():
            if v is None:
                del knobs[k]
            else:
                knobs[k] = v


@pytest.mark.parametrize("pieces", ["1", "0"])
@pytest.mark.parametrize("park,waves,chunk", [("0", "0", "32"), ("32", "0", "32"), ("32", "4", "7"), ("1", "8", "64"),
                                              ("32", "64", "1"), ("16", "3", "32")])
def test_bounce_flat_rollout_shared_drain_protocol(batch_mod, park, waves, chunk, pieces):
    """The flat Bounce kernel's drain: waves that run out of boards park their last ones in LDS for the waves of their
    workgroup that still run, and the last wave standing sweeps up.  Batch sizes around the wave / workgroup / chunk
    boundaries, few waves with many boards each, many waves with nothing to do, parking thresholds 0 (off), 1, 16, 32:
    every board must be finished exactly once and match the oracle, the step count included."""
    import os

    env = {"bounce_group": "1", "bounce_park": park, "bounce_chunk": chunk, "bounce_pieces": pieces}
    if waves != "0":
        env["bounce_flat_waves"] = waves
    old = {k: knobs.get(k) for k in env}
    knobs.update(env)
    try:
        for n in (1, 63, 64, 65, 255, 256, 257, 1000, 4099, 20011):
            dev = batch_mod.BounceBatch(DEFAULT_BOUNCE, n)
            orc = oracle.BounceOracle(DEFAULT_BOUNCE, n)
            dev.set_first_game(n)
            dev.rollout(SEED ^ n, max_plies=3000, from_initial=True)
            total = orc.rollout(SEED ^ n, first_game=n, max_plies=3000)
            assert_same(dev, orc, f"park {park} waves {waves} chunk {chunk} n={n}")
            assert dev.steps == total, (n, dev.steps, total)
            # resumed from memory with a cap, then to the end: the not-from-initial variant of the same kernel
            dev.reset()
            orc.reset()
            dev.reset_steps()
            dev.rollout(SEED ^ n, max_plies=5)
            dev.rollout(SEED ^ n, max_plies=3000)
            orc.rollout(SEED ^ n, first_game=n, max_plies=3000)
            assert_same(dev, orc, f"park {park} waves {waves} chunk {chunk} n={n}, capped then finished")
            assert dev.steps == total
            dev.close()
    finally:
        for k, v in old.items():
            if v is None:
                del knobs[k]
            else:
                knobs[k] = v


@pytest.mark.parametrize("pool,park,waves,chunk", [("1", "40", "0", "0"), ("1", "63", "128", "32"), ("1", "3", "512", "64"),
                                                   ("1", "32", "8", "32"), ("0", "40", "0", "0"), ("1", "40", "1024", "32")])
def test_bounce_device_wide_pool_of_parked_boards(batch_mod, pool, park, waves, chunk):
    """Round 4: the last wave of a workgroup parks its last boards in GLOBAL memory for the last waves of other workgroups
    (K3p's device-wide pool), and boards leave the kernel as piece positions that a follow-up kernel turns into value
    planes.  Parking thresholds 3 / 32 / 40 / 63, a handful and a thousand waves, the pool off: every board finished exactly
    once and equal to the oracle (grids, rewards, plies, status, step count), uncapped and with caps that stop every
    board of a wave in the same iteration (the race that loses parked boards), repeated."""
    import os

    env = {"bounce_group": "1", "bounce_pool": pool, "bounce_pieces_park": park}
    if waves != "0":
        env["bounce_flat_waves"] = waves
    if chunk != "0":
        env["bounce_chunk"] = chunk
    old = {k: knobs.get(k) for k in env}
    knobs.update(env)
    try:
        for n in (4099, 20011, 70001):
            want = {}
            for cap in (3000, 7, 2):
                orc = oracle.BounceOracle(DEFAULT_BOUNCE, n)
                want[cap] = (orc.rollout(SEED ^ n, first_game=n, max_plies=cap), orc)
            for rep in range(3):
                for cap in (3000, 7, 2):
                    dev = batch_mod.BounceBatch(DEFAULT_BOUNCE, n)
                    dev.set_first_game(n)
                    dev.rollout(SEED ^ n, max_plies=cap, from_initial=True)
                    assert_same(dev, want[cap][1], f"pool {pool} park {park} waves {waves} n={n} cap={cap} rep {rep}")
                    assert dev.steps == want[cap][0], (n, cap, rep, dev.steps, want[cap][0])
                    dev.close()
    finally:
        for k, v in old.items():
            if v is None:
                knobs.pop(k, None)
            else:
                knobs[k] = v


@pytest.mark.parametrize("pieces", ["1", "0"])
def test_bounce_short_caps_leave_no_parked_board_behind(batch_mod, pieces):
    """With a ply cap every board of a wave stops in the same iteration.  The last wave of a workgroup used to leave then --
    all its lanes had been busy, so it had not looked at the parked boards in that iteration -- and the boards another wave
    had parked meanwhile stayed at the state memory held (round 3: seen as a step count that was off in one run of eight
    at 20011 boards; found through the boards with plies = 0).  Repeated, because it is a race: capped rollouts from the
    start position (K3p / K3f) and capped-then-finished rollouts from memory (K3f), every board and the step counter."""
    import os

    env = {"bounce_group": "1", "bounce_park": "32", "bounce_chunk": "32", "bounce_pieces": pieces}
    old = {k: knobs.get(k) for k in env}
    knobs.update(env)
    try:
        n = 20011
        want = {}
        for cap in (5, 2, 9):
            orc = oracle.BounceOracle(DEFAULT_BOUNCE, n)
            want[cap] = (orc.rollout(SEED ^ n, first_game=n, max_plies=cap), orc.plies.copy(), orc.grid.copy())
        full = oracle.BounceOracle(DEFAULT_BOUNCE, n)
        full_steps = full.rollout(SEED ^ n, first_game=n, max_plies=3000)
        for rep in range(10):
            for cap in (5, 2, 9):
                dev = batch_mod.BounceBatch(DEFAULT_BOUNCE, n)
                dev.set_first_game(n)
                dev.rollout(SEED ^ n, max_plies=cap, from_initial=True)
                np.testing.assert_array_equal(dev.plies, want[cap][1], err_msg=f"rep {rep} cap {cap} from the start position")
                np.testing.assert_array_equal(dev.grid, want[cap][2])
                assert dev.steps == want[cap][0]
                dev.reset()
                dev.reset_steps()
                dev.rollout(SEED ^ n, max_plies=cap)        # boards from memory: K3f
                np.testing.assert_array_equal(dev.plies, want[cap][1], err_msg=f"rep {rep} cap {cap} from memory")
                assert dev.steps == want[cap][0]
                if cap == 5:
                    dev.rollout(SEED ^ n, max_plies=3000)
                    assert dev.steps == full_steps
                    np.testing.assert_array_equal(dev.grid, full.grid)
                dev.close()
    finally:
        for k, v in old.items():
            if v is None:
                del knobs[k]
            else:
                knobs[k] = v


def test_unsupported_geometry_is_an_error(batch_mod):
    """Beyond the bit-packed limits the generic kernels take over (tests/test_gpu_generic.py); beyond THEIR limits the
    library refuses, loudly and before touching the GPU."""
    assert batch_mod.ConnectBatch(16, 7, 4, 8).generic
    assert batch_mod.ConnectBatch(15, 13, 4, 8).generic  # 208 bits per plane
    assert batch_mod.BounceBatch(np.zeros((9, 8), dtype=np.int8), 8).generic  # 72 cells
    assert not batch_mod.ConnectBatch(15, 12, 4, 8).generic
    with pytest.raises(ValueError):
        batch_mod.ConnectBatch(65, 7, 4, 8)
    with pytest.raises(ValueError):
        batch_mod.ConnectBatch(6, 7, 0, 8)
    with pytest.raises(ValueError):
        batch_mod.BounceBatch(np.zeros((33, 32), dtype=np.int8), 8)  # more than 1024 cells
    with pytest.raises(ValueError):
        batch_mod.BounceBatch(np.zeros((2, 6), dtype=np.int8), 8)   # no interior row
    bad = np.zeros((9, 6), dtype=np.int8)
    bad[0, 2] = 1
    with pytest.raises(ValueError):
        batch_mod.BounceBatch(bad, 8)  # piece in a goal row


# ------------------------------------------------------------------------------------------------ Bounce

BOUNCE_GRIDS = {
    "default": DEFAULT_BOUNCE,
    "small": np.array([[0, 0, 0], [1, 2, 3], [0, 0, 0], [0, 0, 0], [1, 2, 3], [0, 0, 0]], dtype=np.int8),
    "big_values": np.array(
        [[0] * 6, [0, 0, 0, 0, 7, 0], [0] * 6, [0, 5, 0, 0, 0, 0], [0] * 6, [0] * 6, [0] * 6, [1, 0, 0, 15, 0, 0], [0] * 6],
        dtype=np.int8,
    ),
    "crowded": np.array(
        [[0] * 8, [1, 2, 3, 1, 2, 3, 1, 2], [3, 2, 1, 3, 2, 1, 3, 2], [0] * 8, [2, 2, 2, 2, 2, 2, 2, 2], [1, 1, 1, 3, 3, 1, 1, 1],
         [1, 2, 3, 4, 4, 3, 2, 1], [0] * 8],
        dtype=np.int8,
    ),
    "narrow": np.array([[0], [1], [0], [2], [0]], dtype=np.int8),
    "wide": np.array([[0] * 12, [1, 2, 0, 3, 1, 0, 2, 2, 0, 1, 3, 1], [0] * 12, [2, 1, 3, 0, 0, 2, 1, 0, 3, 1, 0, 2], [0] * 12], dtype=np.int8),
    "blocked_start": np.array([[0, 0], [2, 2], [2, 2], [0, 0]], dtype=np.int8),
}


def assert_bounce_actions(dev, orc, idx):
    masks = dev.targets
    width = dev.width
    for i in idx:
        want = orc.actions(int(i))
        got = []
        row = int(masks[i, width])
        for x in range(width):
            m = int(masks[i, x])
            got += [((x, row), (c % width, c // width)) for c in range(64) if (m >> c) & 1]
        assert got == want, f"board {i}"


@pytest.mark.parametrize("name", list(BOUNCE_GRIDS))
def test_bounce_step_random_lockstep(batch_mod, name):
    grid = BOUNCE_GRIDS[name]
    n = 700
    dev = batch_mod.BounceBatch(grid, n)
    orc = oracle.BounceOracle(grid, n)
    dev.set_first_game(12345)
    assert_same(dev, orc, "after reset")
    total = 0
    for ply in range(60):
        np.testing.assert_array_equal(dev.action_count, orc.count_actions(), err_msg=f"action count, ply {ply}")
        if ply % 7 == 0:
            assert_bounce_actions(dev, orc, range(0, n, 97))
        total += orc.step_random(SEED, first_game=12345)
        dev.step_random(SEED)
        assert_same(dev, orc, f"ply {ply}")
        assert dev.steps == total
        if orc.ended.all():
            break


@pytest.mark.parametrize("name", list(BOUNCE_GRIDS))
@pytest.mark.parametrize("from_initial", [False, True])
def test_bounce_rollout(batch_mod, name, from_initial):
    grid = BOUNCE_GRIDS[name]
    n = 6000
    dev = batch_mod.BounceBatch(grid, n)
    orc = oracle.BounceOracle(grid, n)
    dev.set_first_game(999)
    if from_initial:
        dev.step_random(SEED)
        dev.reset_steps()
    dev.rollout(SEED, max_plies=4096, from_initial=from_initial)
    total = orc.rollout(SEED, first_game=999, max_plies=4096)
    assert_same(dev, orc)
    assert dev.steps == total == int(orc.plies.sum())


PIECE_LIST_GRIDS = dict(
    BOUNCE_GRIDS,
    fourteen=np.array([[0] * 8, [1, 2, 3, 4, 3, 2, 1, 0], [0] * 8, [0, 2, 0, 0, 5, 0, 0, 1], [0] * 8, [1, 2, 3, 1, 0, 0, 0, 1], [0] * 8],
                      dtype=np.int8),
    sixteen=np.array([[0] * 8, [1, 2, 3, 1, 2, 3, 1, 2], [0] * 8, [0] * 8, [0] * 8, [2, 1, 3, 2, 1, 3, 2, 1], [0] * 8], dtype=np.int8),
    nine=np.array([[0] * 5, [1, 1, 2, 0, 3], [0, 0, 6, 0, 0], [0] * 5, [3, 0, 2, 1, 1], [0] * 5], dtype=np.int8),
)


@pytest.mark.parametrize("name", list(PIECE_LIST_GRIDS))
def test_bounce_piece_list_rollout(batch_mod, name):
    """K3p (one lane per board, the board as the cells of its pieces, every piece's segment run with a wave-uniform trip
    count) against the oracle on grids of 2 to 16 pieces -- the 8-, 12- and 16-piece instantiations -- values up to 15,
    a blocked start position, and grids outside its reach (40 pieces, 12 columns), which fall through to the kernels
    they always had."""
    import os

    grid = PIECE_LIST_GRIDS[name]
    old = knobs.get("bounce_group")
    knobs["bounce_group"] = "1"
    try:
        for n, cap in ((6000, 4096), (333, 60)):
            dev = batch_mod.BounceBatch(grid, n)
            orc = oracle.BounceOracle(grid, n)
            dev.set_first_game(5 << 32)
            dev.rollout(SEED + 3, max_plies=cap, from_initial=True)
            total = orc.rollout(SEED + 3, first_game=5 << 32, max_plies=cap)
            assert_same(dev, orc, f"{name} n={n} cap={cap}")
            assert dev.steps == total == int(orc.plies.sum())
            # ... and on from where the cap stopped them (boards loaded from memory: K3f)
            dev.rollout(SEED + 3, max_plies=4096)
            orc.rollout(SEED + 3, first_game=5 << 32, max_plies=4096)
            assert_same(dev, orc, f"{name} n={n} resumed")
            dev.close()
    finally:
        if old is None:
            del knobs["bounce_group"]
        else:
            knobs["bounce_group"] = old


@pytest.mark.parametrize("depth", ["1", "2", "3", "4"])
@pytest.mark.parametrize("name", list(PIECE_LIST_GRIDS))
def test_bounce_opening_book(batch_mod, monkeypatch, name, depth):
    """Round 5: K3p's lanes do not search the first plies of a game from the start position -- they walk the start
    position's OPENING BOOK (every path of `depth` plies, enumerated once per start position with K3p's own search) with the
    game's own draws and start `depth` plies in.  Boards, plies, rewards and step counts must be the oracle's -- on grids of 2
    to 16 pieces (values up to 15, a blocked start: no book; paths that END inside the book: a piece in the goal row after two
    plies on the small grids), for every depth, under ply caps below, at and above the depth, and resumed from memory."""
    grid = PIECE_LIST_GRIDS[name]
    monkeypatch.setitem(knobs, "bounce_group", "1")
    monkeypatch.setenv("BGS_BOUNCE_BOOK", depth)   # (forces the book for a batch of any size)
    d = int(depth)
    for n, cap in ((6000, 4096), (777, d), (500, max(1, d - 1)), (333, d + 1)):
        dev = batch_mod.BounceBatch(grid, n)
        orc = oracle.BounceOracle(grid, n)
        dev.set_first_game(9 << 32)
        dev.rollout(SEED + 5, max_plies=cap, from_initial=True)
        total = orc.rollout(SEED + 5, first_game=9 << 32, max_plies=cap)
        assert_same(dev, orc, f"{name} depth={depth} n={n} cap={cap}")
        assert dev.steps == total == int(orc.plies.sum())
        dev.rollout(SEED + 5, max_plies=4096)
        orc.rollout(SEED + 5, first_game=9 << 32, max_plies=4096)
        assert_same(dev, orc, f"{name} depth={depth} n={n} resumed")
        dev.close()


def test_bounce_opening_book_is_shared_and_released(batch_mod, monkeypatch):
    """The book belongs to the start position, not to the batch: batches of one start position on one device share it (the
    second create does not build), a different start position gets its own, and everything still plays the oracle's games
    after the first batch is gone."""
    monkeypatch.setitem(knobs, "bounce_group", "1")
    monkeypatch.setenv("BGS_BOUNCE_BOOK", "3")
    a = batch_mod.BounceBatch(DEFAULT_BOUNCE, 4000)
    b2 = batch_mod.BounceBatch(DEFAULT_BOUNCE, 2500)
    c = batch_mod.BounceBatch(PIECE_LIST_GRIDS["nine"], 3000)
    a.close()
    for dev, grid, n in ((b2, DEFAULT_BOUNCE, 2500), (c, PIECE_LIST_GRIDS["nine"], 3000)):
        orc = oracle.BounceOracle(grid, n)
        dev.rollout(SEED + 8, max_plies=4096, from_initial=True)
        total = orc.rollout(SEED + 8, max_plies=4096)
        assert_same(dev, orc, "shared book")
        assert dev.steps == total
        dev.close()


@pytest.mark.parametrize("name", list(PIECE_LIST_GRIDS))
def test_bounce_one_board_per_wave_pass(batch_mod, name):
    """K3w, the last pass of the automatic plan (one board per wave, a piece per lane: the few games that run for thousands
    of plies), forced onto MOST of a batch by a plan whose earlier passes stop after 6 and 20 plies -- against the oracle on
    grids of 2 to 16 pieces (the 8-, 12- and 16-lane instantiations), values up to 15, a blocked start position, ply caps
    that stop games inside the pass; grids it cannot take (40 pieces, 12 columns) keep the kernels they had."""
    import os

    grid = PIECE_LIST_GRIDS[name]
    env = {"bounce_group": "1", "bounce_plan": "6:1,20:8,0:64"}
    old = {k: knobs.get(k) for k in env}
    knobs.update(env)
    try:
        for n, cap in ((2500, 4096), (700, 57), (64, 21), (300, 22)):
            dev = batch_mod.BounceBatch(grid, n)
            orc = oracle.BounceOracle(grid, n)
            dev.set_first_game(7 << 32)
            dev.rollout(SEED + 11, max_plies=cap, from_initial=True)
            total = orc.rollout(SEED + 11, first_game=7 << 32, max_plies=cap)
            assert_same(dev, orc, f"{name} n={n} cap={cap}")
            assert dev.steps == total == int(orc.plies.sum())
            dev.close()
    finally:
        for k, v in old.items():
            if v is None:
                del knobs[k]
            else:
                knobs[k] = v


@pytest.mark.parametrize("launches", [1, 4, 8, 16, 64])
def test_bounce_launch_shape_follows_the_hint_and_the_boards_do_not(batch_mod, launches):
    """`set_launches_in_flight` only shapes the launch (bulk-pass ply cap, boards per wave: bounce_shape() in
    bgs_internal.h): one launch at a time, 4, 8, 16 in flight -- the same boards as the oracle's, ply caps either side of
    every shape's bulk cap included."""
    for n, cap in ((40000, 4096), (3000, 100), (3000, 200), (3000, 400)):
        dev = batch_mod.BounceBatch(DEFAULT_BOUNCE, n)
        orc = oracle.BounceOracle(DEFAULT_BOUNCE, n)
        dev.set_launches_in_flight(launches)
        dev.set_first_game(9 << 32)
        dev.rollout(SEED + launches, max_plies=cap, from_initial=True)
        total = orc.rollout(SEED + launches, first_game=9 << 32, max_plies=cap)
        assert_same(dev, orc, f"in flight {launches}: n={n} cap={cap}")
        assert dev.steps == total
        dev.close()
    with pytest.raises(ValueError):
        batch_mod.BounceBatch(DEFAULT_BOUNCE, 8).set_launches_in_flight(0)


def test_bounce_rollout_max_plies_and_resume(batch_mod):
    n = 4000
    dev = batch_mod.BounceBatch(DEFAULT_BOUNCE, n)
    orc = oracle.BounceOracle(DEFAULT_BOUNCE, n)
    dev.rollout(SEED, max_plies=11)
    orc.rollout(SEED, max_plies=11)
    assert_same(dev, orc, "capped")
    assert not orc.ended.all()
    dev.rollout(SEED, max_plies=4096)
    orc.rollout(SEED, max_plies=4096)
    assert_same(dev, orc, "resumed")


def test_bounce_step_actions(batch_mod):
    n = 600
    rng = np.random.default_rng(11)
    dev = batch_mod.BounceBatch(DEFAULT_BOUNCE, n)
    orc = oracle.BounceOracle(DEFAULT_BOUNCE, n)
    for _ in range(25):
        moves = np.full((n, 4), -1, dtype=np.int32)
        for i in range(n):
            acts = orc.actions(i)
            r = rng.random()
            if acts and r < 0.7:
                (sx, sy), (tx, ty) = acts[rng.integers(len(acts))]
                moves[i] = [sx, sy, tx, ty]
            elif r < 0.85:
                moves[i] = rng.integers(0, 9, size=4)  # mostly illegal, sometimes out of range
        st_dev = dev.step_actions(moves)
        st_orc = orc.step_actions(moves)
        np.testing.assert_array_equal(st_dev, st_orc)
        assert_same(dev, orc)
    assert (st_orc == -2).any()


def test_bounce_write_state_roundtrip(batch_mod):
    n = 2000
    orc = oracle.BounceOracle(DEFAULT_BOUNCE, n)
    for _ in range(9):
        orc.step_random(SEED + 9)
    dev = batch_mod.BounceBatch(DEFAULT_BOUNCE, n)
    assert (dev.write_state(orc.grid, orc.player, orc.winner, orc.plies) == 0).all()
    assert_same(dev, orc, "loaded")
    dev.rollout(SEED + 9, max_plies=4096)
    orc.rollout(SEED + 9, max_plies=4096)
    assert_same(dev, orc, "continued")
    # malformed boards are refused and left untouched
    bad = orc.grid.copy()
    bad[0, 0, 0] = 3        # piece in a goal row of a running board
    bad[1, 4, 2] = -1       # negative value
    winner = orc.winner.copy()
    winner[:2] = -1
    status = dev.write_state(bad, orc.player, winner, orc.plies)
    assert status[0] == -1 and status[1] == -1 and (status[2:] == 0).all()


@pytest.mark.parametrize("seed_offset,first_game", [(0, 196997), (5, 2278), (5, 260696)])
def test_bounce_games_that_never_end(batch_mod, seed_offset, first_game):
    """The games of the 2^18-board batches that run into max_plies (tools/bounce_endless.py found them: a handful of positions
    visited over and over), each with 47 neighbours in a small batch: the one-board-per-wave pass plays them from its memo
    of action lists and hops along its links between remembered positions for thousands of plies -- the board after 4096,
    1000 and 999 plies (the cap inside a run of hops, even and odd) must be the oracle's, and so must a rollout that stops
    at 300 plies and goes on from memory."""
    n = 48
    for cap in (4096, 1000, 999):
        dev = batch_mod.BounceBatch(DEFAULT_BOUNCE, n)
        orc = oracle.BounceOracle(DEFAULT_BOUNCE, n)
        dev.set_first_game(first_game - 17)
        dev.rollout(SEED + seed_offset, max_plies=cap, from_initial=True)
        total = orc.rollout(SEED + seed_offset, first_game=first_game - 17, max_plies=cap)
        assert int(orc.plies[17]) == cap, "the fixture's endless game is not where the test expects it"
        assert_same(dev, orc, f"game {first_game}, cap {cap}")
        assert dev.steps == total
        dev.close()
    dev = batch_mod.BounceBatch(DEFAULT_BOUNCE, n)
    orc = oracle.BounceOracle(DEFAULT_BOUNCE, n)
    dev.set_first_game(first_game - 17)
    dev.rollout(SEED + seed_offset, max_plies=300, from_initial=True)
    dev.rollout(SEED + seed_offset, max_plies=4096)
    orc.rollout(SEED + seed_offset, first_game=first_game - 17, max_plies=4096)
    assert_same(dev, orc, f"game {first_game}, resumed at 300")
    dev.close()


@pytest.mark.parametrize("policy", ["0:0", "4:28", "1:3"])
@pytest.mark.parametrize("limit", ["2", "3", "7"])
def test_bounce_memo_starts_over_when_its_epochs_run_out(batch_mod, monkeypatch, limit, policy):
    """Round-4 advisor: a link of K3w's memo holds its epoch in 16 bits, and when the count of replacements reaches the
    limit the memo starts over empty -- stale rows of links are left behind and must not be followed.  2^16 replacements
    do not happen in a test, so BGS_EXPERIMENT=bounce_epoch_limit=<n> brings the restart within reach: a few waves, each playing a dozen
    boards -- among them a game that never ends -- replace remembered positions all the time.
    Round 5: a game whose look-ups keep missing plays some plies WITHOUT the memo (bounce_memo_policy=misses:plies; default
    4:28).  "0:0" keeps the memo on every ply (the most replacements, as in round 4), "1:3" switches between the two modes
    all the time -- links written before a stretch without the memo must still be right after it."""
    monkeypatch.setitem(knobs, "bounce_epoch_limit", limit)
    monkeypatch.setitem(knobs, "bounce_memo_policy", policy)
    monkeypatch.setitem(knobs, "bounce_wave_grid", "4")
    n = 48
    for seed_offset, first_game, cap in ((0, 196997, 4096), (5, 2278, 1500)):
        dev = batch_mod.BounceBatch(DEFAULT_BOUNCE, n)
        orc = oracle.BounceOracle(DEFAULT_BOUNCE, n)
        dev.set_first_game(first_game - 17)
        dev.rollout(SEED + seed_offset, max_plies=cap, from_initial=True)
        total = orc.rollout(SEED + seed_offset, first_game=first_game - 17, max_plies=cap)
        assert int(orc.plies[17]) == cap
        assert_same(dev, orc, f"epoch limit {limit}, game {first_game}, cap {cap}")
        assert dev.steps == total
        dev.close()


def test_bounce_loaded_boards_of_any_crowd_finish_on_one_board_per_wave(batch_mod):
    """A batch CONFIGURED with 12 pieces, LOADED with other people's boards -- 12, 16 and 22 pieces on the same 9x6 cells --
    and rolled out from memory: the tail pass (K3w, 16 lanes here) plays what fits its lanes a piece per lane and hands a
    board with more pieces to lane 0's thread-per-board code; all of it against the oracle, ply caps inside both passes."""
    n = 1536
    crowded16 = DEFAULT_BOUNCE.copy()
    crowded16[3, [0, 2, 3, 5]] = [2, 1, 3, 1]
    crowded22 = crowded16.copy()
    crowded22[5] = [1, 1, 2, 2, 3, 3]
    starts = [DEFAULT_BOUNCE, crowded16, crowded22]
    for cap in (4096, 40, 17):
        grids = np.stack([starts[i % 3] for i in range(n)]).astype(np.int8)
        orc = oracle.BounceOracle(DEFAULT_BOUNCE, n)
        orc.grid[...] = grids
        dev = batch_mod.BounceBatch(DEFAULT_BOUNCE, n)
        assert (dev.write_state(orc.grid, orc.player, orc.winner, orc.plies) == 0).all()
        dev.set_first_game(11 << 32)
        dev.rollout(SEED + 21, max_plies=cap)
        total = orc.rollout(SEED + 21, first_game=11 << 32, max_plies=cap)
        assert_same(dev, orc, f"loaded crowds, cap {cap}")
        assert dev.steps == total
        dev.close()


def test_bounce_full_size_batch(batch_mod):
    """BASELINE config 4: Bounce default grid, batch 2^18, max_plies 4096."""
    n = 1 << 18
    dev = batch_mod.BounceBatch(DEFAULT_BOUNCE, n)
    dev.rollout(SEED, max_plies=4096, from_initial=True)
    steps = dev.steps
    reward, winner, plies, grid = dev.reward, dev.winner, dev.plies, dev.grid
    assert steps == int(plies.sum())
    assert (reward.sum(axis=1) == 0).all()
    assert (grid.sum(axis=(1, 2)) == int(DEFAULT_BOUNCE.sum())).all()  # pieces are conserved
    assert ((grid > 0).sum(axis=(1, 2)) == 12).all()
    in_goal = (grid[:, 0] > 0).any(axis=1) | (grid[:, -1] > 0).any(axis=1)
    assert (in_goal <= (winner >= 0)).all()
    assert (dev.action_count[winner >= 0] == 0).all()
    orc = oracle.BounceOracle(DEFAULT_BOUNCE, n)
    total = orc.rollout(SEED, max_plies=4096)
    assert total == steps
    np.testing.assert_array_equal(grid, orc.grid)
    np.testing.assert_array_equal(reward, orc.reward)
    np.testing.assert_array_equal(plies, orc.plies)


# ------------------------------------------------------------------------------------------------ ragged sizes


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 255, 257, 4097])
def test_ragged_batch_sizes(batch_mod, n):
    """Batches that do not fill a wave / workgroup / rollout chunk."""
    dev = batch_mod.ConnectBatch(6, 7, 4, n)
    orc = oracle.ConnectOracle(6, 7, 4, n)
    dev.set_first_game(5)
    for _ in range(3):
        dev.step_random(SEED)
        orc.step_random(SEED, first_game=5)
    assert_same(dev, orc, "stepped")
    dev.rollout(SEED)
    orc.rollout(SEED, first_game=5)
    assert_same(dev, orc, "rolled out")
    assert dev.steps == int(orc.plies.sum())
    bd = batch_mod.BounceBatch(DEFAULT_BOUNCE, n)
    bo = oracle.BounceOracle(DEFAULT_BOUNCE, n)
    bd.rollout(SEED, max_plies=300, from_initial=True)
    bo.rollout(SEED, max_plies=300)
    assert_same(bd, bo, "bounce")


def test_large_batch_2_pow_23(batch_mod):
    """BASELINE config 5's total size on one GPU: 2^23 boards; rewards of the shards [r*2^20, (r+1)*2^20) must be the
    slices of the whole (this is what 8 ranks would gather)."""
    n = 1 << 23
    dev = batch_mod.ConnectBatch(6, 7, 4, n)
    dev.rollout(SEED, from_initial=True)
    reward = dev.reward
    assert dev.has_ended.all() and dev.steps == int(dev.plies.sum())
    for r in (0, 3, 7):
        part = batch_mod.ConnectBatch(6, 7, 4, 1 << 20)
        part.set_first_game(r << 20)
        part.rollout(SEED, from_initial=True)
        np.testing.assert_array_equal(part.reward, reward[r << 20 : (r + 1) << 20])
    orc = oracle.ConnectOracle(6, 7, 4, 1 << 16)
    orc.rollout(SEED, first_game=(7 << 20) + 12345)
    np.testing.assert_array_equal(orc.reward, reward[(7 << 20) + 12345 : (7 << 20) + 12345 + (1 << 16)])


def test_outcome_codes_roundtrip(batch_mod):
    """2-bit outcome codes (what ranks exchange) expand to exactly the reward array."""
    import torch

    for n in (1, 5, 4096, 100003):
        dev = batch_mod.ConnectBatch(6, 7, 4, n, use_torch=True)
        dev.rollout(SEED, max_plies=20, from_initial=True)  # a mix of finished and running boards
        codes = dev.outcomes_tensor()
        assert codes.shape == ((n + 3) // 4,)
        winner = dev.winner
        want = np.where(winner == -1, 0, np.where(winner == 2, 3, winner + 1)).astype(np.uint8)
        got = codes.cpu().numpy()
        unpacked = np.stack([(got >> (2 * j)) & 3 for j in range(4)], axis=1).reshape(-1)[:n]
        np.testing.assert_array_equal(unpacked, want)
        reward = batch_mod.expand_outcomes(codes, n)
        torch.cuda.synchronize()
        np.testing.assert_array_equal(reward.cpu().numpy(), dev.reward)


def test_transition_round_trip_matches_separate_calls(batch_mod):
    """bgs_transition (load + move + observe in one round trip) against the oracle, Connect and Bounce."""
    rng = np.random.default_rng(21)
    n = 700
    orc = oracle.ConnectOracle(6, 7, 4, n)
    for _ in range(11):
        orc.step_random(SEED)
    dev = batch_mod.ConnectBatch(6, 7, 4, n)
    cols = rng.integers(-1, 8, size=n).astype(np.int32)
    status, grid, player, winner, plies, legal, reward = dev.transition(orc.grid, orc.player, orc.winner, None, cols)
    np.testing.assert_array_equal(status, orc.step_actions(cols))
    np.testing.assert_array_equal(reward, orc.reward)
    np.testing.assert_array_equal(grid, orc.grid)
    np.testing.assert_array_equal(player, orc.player)
    np.testing.assert_array_equal(winner, orc.winner)
    np.testing.assert_array_equal(plies, orc.plies)
    np.testing.assert_array_equal(legal, orc.legal())
    # malformed boards are reported and leave the previous board in place
    bad = orc.grid.copy()
    bad[0, 5, 0] = 1 if bad[0, 4, 0] == -1 else bad[0, 5, 0]
    bad[1, 0, 0] = 7
    status2, grid2, *_ = dev.transition(bad, orc.player, orc.winner)
    assert status2[1] == -1 and (status2[2:] == 0).all()
    np.testing.assert_array_equal(grid2[1], orc.grid[1])

    bo = oracle.BounceOracle(DEFAULT_BOUNCE, n)
    for _ in range(5):
        bo.step_random(SEED)
    bd = batch_mod.BounceBatch(DEFAULT_BOUNCE, n)
    moves = np.full((n, 4), -1, dtype=np.int32)
    for i in range(n):
        acts = bo.actions(i)
        if acts and rng.random() < 0.8:
            (sx, sy), (tx, ty) = acts[rng.integers(len(acts))]
            moves[i] = [sx, sy, tx, ty]
        elif rng.random() < 0.5:
            moves[i] = rng.integers(0, 9, size=4)
    status, grid, player, winner, plies, masks, reward = bd.transition(bo.grid, bo.player, bo.winner, bo.plies, moves)
    np.testing.assert_array_equal(status, bo.step_actions(moves))
    np.testing.assert_array_equal(reward, bo.reward)
    np.testing.assert_array_equal(grid, bo.grid)
    np.testing.assert_array_equal(winner, bo.winner)
    np.testing.assert_array_equal(plies, bo.plies)
    np.testing.assert_array_equal(masks, bd.targets)


# ------------------------------------------------------------------------------------------------ the strict RNG contract
# Round 6: Connect under BGS_RNG_PER_PLY (a philox word per ply, exactly Bounce's rule; the form SURVEY 7.3 wrote) -- every
# kernel family that draws, against the oracle's ORC_RNG_PER_PLY mode.

@pytest.mark.parametrize("h,w,k", CONNECT_GEOMETRIES + [(17, 5, 4), (4, 20, 3)])   # (the last two: the generic kernels)
def test_connect_strict_contract_step_random_lockstep(batch_mod, h, w, k):
    """K1s (one-word boards, even n), k_connect_step_random (the rest) and g_connect_play, one ply and several per launch."""
    n = 1500
    dev = batch_mod.ConnectBatch(h, w, k, n)
    dev.set_rng_contract("per-ply")
    orc = oracle.ConnectOracle(h, w, k, n, per_ply=True)
    dev.set_first_game(4242)
    total = 0
    for plies in [1] * 6 + [3, 4, 5, 7, h * w]:
        dev.step_random(SEED, plies=plies)
        for _ in range(plies):
            total += orc.step_random(SEED, first_game=4242)
        assert_same(dev, orc, f"after {plies} more plies")
        assert dev.steps == total
    assert orc.ended.all()
    # ... and the two contracts are different games (same seed): the default draws differently from ply 1 on
    other = batch_mod.ConnectBatch(h, w, k, n)
    other.set_first_game(4242)
    other.step_random(SEED, plies=h * w)
    if w > 1 and h * w > 2 and k > 1:   # (ply 0 draws the same word under both: a game of one ply cannot differ)
        assert not np.array_equal(other.grid, dev.grid)


@pytest.mark.parametrize("h,w,k", CONNECT_GEOMETRIES + [(17, 5, 4)])
@pytest.mark.parametrize("how", ["initial", "memory", "capped", "flag"])
def test_connect_strict_contract_rollout(batch_mod, h, w, k, how):
    """The fused rollouts: K2o (one-word, from the start, no cap), K2a (from memory / capped), K2c + its opening launch
    (12x13x5), K2b and the any-geometry kernel, the generic kernel; the contract set on the batch or asked for by flag."""
    from simulator.game import _abi
    import ctypes

    n = 20000 if h * w <= 64 else 6000
    dev = batch_mod.ConnectBatch(h, w, k, n)
    orc = oracle.ConnectOracle(h, w, k, n, per_ply=True)
    dev.set_first_game(1 << 35)
    if how != "flag":
        dev.set_rng_contract("per-ply")
    cap = 2**31 - 1
    if how == "memory":   # a few plies first (lanes then join in the middle of a block), then from memory
        for _ in range(3):
            dev.step_random(SEED ^ 5)
            orc.step_random(SEED ^ 5, first_game=1 << 35)
        dev.reset_steps()
    if how == "capped":
        cap = max(1, (h * w) // 2)
    if how == "flag":
        _abi.check(_abi.lib().bgs_rollout(dev._handle, ctypes.c_uint64(SEED), ctypes.c_int32(cap),
                                          ctypes.c_uint32(_abi.ROLLOUT_FROM_INITIAL | _abi.ROLLOUT_DRAW_PER_PLY)))
    else:
        dev.rollout(SEED, max_plies=cap, from_initial=(how != "memory"))
    total = orc.rollout(SEED, first_game=1 << 35, max_plies=cap)
    assert_same(dev, orc, how)
    assert dev.steps == total


def test_connect_strict_contract_kernel_families_agree(batch_mod, monkeypatch):
    """Connect4 under the strict contract through every one-word rollout family (K2o with 1-4 opening blocks, K2a, the
    any-geometry kernel) and 12x13x5 through K2c / K2b / the any-geometry kernel: the same boards, and the oracle's."""
    n = 30000
    orc = oracle.ConnectOracle(6, 7, 4, n, per_ply=True)
    total = orc.rollout(SEED + 9, first_game=5)
    for setting in ({}, {"rollout_opening": "0"}, {"rollout_opening": "1"}, {"rollout_opening": "2"}, {"rollout_opening": "4"},
                    {"rollout_generic": "1"}):
        for name, value in setting.items():
            monkeypatch.setitem(knobs, name, value)
        dev = batch_mod.ConnectBatch(6, 7, 4, n)
        dev.set_rng_contract("per-ply")
        dev.set_first_game(5)
        dev.rollout(SEED + 9, from_initial=True)
        assert_same(dev, orc, str(setting))
        assert dev.steps == total
        dev.close()
        for name in setting:
            monkeypatch.delitem(knobs, name)
    n = 6000
    orc = oracle.ConnectOracle(12, 13, 5, n, per_ply=True)
    total = orc.rollout(SEED + 9, first_game=5)
    for setting in ({}, {"rollout_opening": "0"}, {"rollout_no_lds": "1"}, {"rollout_generic": "1"}):
        for name, value in setting.items():
            monkeypatch.setitem(knobs, name, value)
        dev = batch_mod.ConnectBatch(12, 13, 5, n)
        dev.set_rng_contract("per-ply")
        dev.set_first_game(5)
        dev.rollout(SEED + 9, from_initial=True)
        assert_same(dev, orc, str(setting))
        assert dev.steps == total
        dev.close()
        for name in setting:
            monkeypatch.delitem(knobs, name)


def test_connect_strict_contract_through_the_reward_sink(batch_mod):
    """The hand-over (fused outcome codes) under the strict contract: host rewards == oracle."""
    n = 1 << 16
    dev = batch_mod.ConnectBatch(6, 7, 4, n)
    dev.set_rng_contract("per-ply")
    sink = batch_mod.RewardSink(n, slots=2, threads=2)
    host = np.full((n, 2), 9, dtype=np.int8)
    sink.wait(sink.rollout(dev, host, SEED + 3, from_initial=True))
    orc = oracle.ConnectOracle(6, 7, 4, n, per_ply=True)
    orc.rollout(SEED + 3)
    np.testing.assert_array_equal(host, orc.reward)
    sink.close()
    with pytest.raises(ValueError):
        dev.set_rng_contract("per-game")


def test_bounce_tail_kernel_beside_the_bulk_kernel(batch_mod, monkeypatch):
    """Experiment bounce_tail=1 (round 6; measured slower, no automatic plan takes it): K3p hands the games that reach its ply cap
    -- and the last boards of a workgroup's last wave -- to a device-wide queue, a second kernel on a stream of the batch's own
    finishes them while the bulk kernel runs.  Same boards as the oracle's, whatever the hand-over threshold and the number
    of tail waves; twice on the same batch (the queue's "entry complete" words are launch serials)."""
    n = 1 << 17
    orc = oracle.BounceOracle(DEFAULT_BOUNCE, n)
    total = orc.rollout(SEED + 41, max_plies=4096)
    for handoff, limit in (("0", "64"), ("16", "512"), ("32", "1024")):
        monkeypatch.setitem(knobs, "bounce_tail", "1")
        monkeypatch.setitem(knobs, "bounce_tail_handoff", handoff)
        monkeypatch.setitem(knobs, "bounce_tail_limit", limit)
        dev = batch_mod.BounceBatch(DEFAULT_BOUNCE, n)
        for _ in range(2):
            dev.reset_steps()
            dev.rollout(SEED + 41, max_plies=4096, from_initial=True)
            assert_same(dev, orc, f"handoff {handoff}, {limit} tail waves")
            assert dev.steps == total
        dev.close()


@pytest.mark.parametrize("hint,n", [(1, 1 << 17), (8, 1 << 16), (20, 1 << 16)])
def test_bounce_compile_time_geometry_equals_the_run_time_record(batch_mod, monkeypatch, hint, n):
    """Round 6: the default board runs on K3p / K3w instantiated on its compile-time geometry (bounce_unit.h: DefaultBounceGeom).
    The same batch on the run-time record (experiment bounce_static_geom=0: what every other grid gets) and the oracle: the
    same boards, in every launch shape, at caps around the bulk caps."""
    for cap in (4096, 241, 129, 81, 80, 5):
        out = []
        for static in ("1", "0"):
            monkeypatch.setitem(knobs, "bounce_static_geom", static)
            dev = batch_mod.BounceBatch(DEFAULT_BOUNCE, n)
            dev.set_launches_in_flight(hint)
            dev.set_first_game(123456789)
            dev.rollout(SEED + cap, max_plies=cap, from_initial=True)
            out.append(dev)
        np.testing.assert_array_equal(out[0].grid, out[1].grid)
        np.testing.assert_array_equal(out[0].reward, out[1].reward)
        np.testing.assert_array_equal(out[0].plies, out[1].plies)
        assert out[0].steps == out[1].steps
        m = 1 << 13
        orc = oracle.BounceOracle(DEFAULT_BOUNCE, m)
        orc.rollout(SEED + cap, first_game=123456789, max_plies=cap)
        np.testing.assert_array_equal(out[0].grid[:m], orc.grid)
        np.testing.assert_array_equal(out[0].reward[:m], orc.reward)
        np.testing.assert_array_equal(out[0].plies[:m], orc.plies)
        for dev in out:
            dev.close()

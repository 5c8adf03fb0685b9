"""The CPU oracle against every fixture the reference's own tests hold for the hot path (CPU only).

tests/golden/reference_connect.json <- reference tests/test_connect.py:68-145
tests/golden/reference_bounce.json  <- reference tests/test_bounce.py:92-410
"""

import json
import os

import numpy as np
import pytest

from oracle import oracle


def _load(golden_dir, name):
    with open(os.path.join(golden_dir, name)) as fh:
        return json.load(fh)


# ------------------------------------------------------------------ RNG contract


def test_philox_known_answers():
    # Random123 known-answer vectors for philox4x32-10 (kat_vectors)
    kat = [
        ([0, 0, 0, 0], [0, 0], [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]),
        ([0xFFFFFFFF] * 4, [0xFFFFFFFF] * 2, [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]),
        (
            [0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344],
            [0xA4093822, 0x299F31D0],
            [0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1],
        ),
    ]
    for ctr, key, want in kat:
        assert oracle.philox4x32_10(ctr, key).tolist() == want


def test_draw_layout():
    seed, game = 0x0123456789ABCDEF, 0x1_0000_0007
    for ply in range(9):
        block = oracle.philox4x32_10([game & 0xFFFFFFFF, game >> 32, ply >> 2, 0], [seed & 0xFFFFFFFF, seed >> 32])
        assert oracle.draw(seed, game, ply) == int(block[ply & 3])
        for n in (1, 2, 7, 13, 236):
            assert oracle.sample_index(seed, game, ply, n) == (int(block[ply & 3]) * n) >> 32


def test_connect_draw_layout():
    """Connect (round 5): one philox WORD per block of four plies -- counter (game lo, game hi, ply >> 4, 0), output word
    (ply >> 2) & 3 -- and the draw of ply j of the block is word * A^j mod 2^32, A = 747796405."""
    seed, game, a = 0x0123456789ABCDEF, 0x1_0000_0007, 747796405
    for ply in range(70):
        block = oracle.philox4x32_10([game & 0xFFFFFFFF, game >> 32, ply >> 4, 0], [seed & 0xFFFFFFFF, seed >> 32])
        word = int(block[(ply >> 2) & 3])
        want = (word * pow(a, ply & 3, 1 << 32)) & 0xFFFFFFFF
        assert oracle.connect_draw(seed, game, ply) == want
        for n in (1, 2, 7, 13, 64):
            assert oracle.connect_sample_index(seed, game, ply, n) == (want * n) >> 32
    # the first ply of a block uses the word itself
    assert oracle.connect_draw(seed, game, 0) == int(oracle.philox4x32_10([game & 0xFFFFFFFF, game >> 32, 0, 0], [seed & 0xFFFFFFFF, seed >> 32])[0])
    assert oracle.SUBDRAW_A == a


def test_connect_sub_draws_are_jointly_uniform_on_a_sample():
    """The four draws of a block are four states of x -> A x mod 2^32: every one of them is a bijection of the word (so a ply's
    index is as uniform as a word of its own would make it) and the exhaustive count over all 2^32 words
    (tools/subdraw_lattice.c; quoted in bgs_oracle.h) puts every four-move sequence of a 7-column board within 4.2e-5 of
    1 / 7^4.  Here: the same count on 2^23 pseudo-random words, loose bounds, to catch a wrong constant."""
    a = 747796405
    words = np.random.default_rng(20261005).integers(0, 1 << 32, size=1 << 23, dtype=np.uint64)
    idx = []
    for j in range(4):
        x = (words * np.uint64(pow(a, j, 1 << 32))) & np.uint64(0xFFFFFFFF)
        idx.append((x * np.uint64(7)) >> np.uint64(32))
    cell = ((idx[0] * 7 + idx[1]) * 7 + idx[2]) * 7 + idx[3]
    counts = np.bincount(cell.astype(np.int64), minlength=7**4)
    expect = len(words) / 7**4
    assert counts.min() > 0.9 * expect and counts.max() < 1.1 * expect
    for j in range(4):
        single = np.bincount(idx[j].astype(np.int64), minlength=7)
        assert abs(single / (len(words) / 7) - 1).max() < 2e-3


@pytest.mark.parametrize("counts", [(7, 7, 7, 6), (7, 6, 6, 5)])
def test_connect_sub_draws_are_jointly_uniform_for_mixed_action_counts(counts):
    """Round-5 review: real blocks see MIXED action counts -- a column fills inside the block -- not four times 7.  The
    exhaustive counts over all 2^32 words for the tuples a 6x7 game meets are in profiles/r06_subdraw_lattice_mixed_counts.txt
    (worst: 5.7e-5 relative, counts 7,7,6,6); here two of them on 2^23 pseudo-random words with loose bounds."""
    a = 747796405
    words = np.random.default_rng(20261006).integers(0, 1 << 32, size=1 << 23, dtype=np.uint64)
    cell = np.zeros(len(words), dtype=np.int64)
    for j, m in enumerate(counts):
        x = (words * np.uint64(pow(a, j, 1 << 32))) & np.uint64(0xFFFFFFFF)
        idx = ((x * np.uint64(m)) >> np.uint64(32)).astype(np.int64)
        assert idx.max() == m - 1
        single = np.bincount(idx, minlength=m)
        assert abs(single / (len(words) / m) - 1).max() < 2e-3
        cell = cell * m + idx
    cells = int(np.prod(counts))
    hist = np.bincount(cell, minlength=cells)
    expect = len(words) / cells
    assert hist.min() > 0.9 * expect and hist.max() < 1.1 * expect


def test_lattice_counts_for_mixed_tuples_are_committed():
    """The exhaustive figures quoted above exist and stay below 1e-4 for every 7-column tuple."""
    import os
    import re

    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r06_subdraw_lattice_mixed_counts.txt")
    rows = [ln for ln in open(path) if ln.startswith("A=747796405 n=7 ")]
    assert len(rows) >= 20
    for ln in rows:
        assert float(re.search(r"4-tuple=([0-9.e+-]+)", ln).group(1)) < 1e-4, ln


def test_connect_strict_contract_in_the_oracle():
    """ORC_RNG_PER_PLY: Connect draws exactly as Bounce does -- ply p of game g takes orc_draw(seed, g, p)."""
    seed, first = 0x0123456789ABCDEF, 1000
    o = oracle.ConnectOracle(6, 7, 4, 64, per_ply=True)
    shadow = oracle.ConnectOracle(6, 7, 4, 64)
    for ply in range(42):
        cols = []
        for g in range(64):
            legal = np.flatnonzero(o.legal()[g])
            cols.append(int(legal[oracle.sample_index(seed, first + g, ply, len(legal))]) if len(legal) and not o.ended[g] else -1)
        o.step_random(seed, first_game=first)
        shadow.step_actions(np.array(cols, dtype=np.int32))
        np.testing.assert_array_equal(o.grid, shadow.grid)
        np.testing.assert_array_equal(o.winner, shadow.winner)


# ------------------------------------------------------------------ Connect


def test_connect_reference_game(golden_dir):
    fx = _load(golden_dir, "reference_connect.json")
    for game in fx["games"]:
        h, w, k = game["config"]
        o = oracle.ConnectOracle(h, w, k, 1)
        for pos in game["positions"]:
            np.testing.assert_array_equal(o.grid[0], np.array(pos["grid"], dtype=np.int8))
            if pos["column"] is None:
                assert o.ended[0]
            else:
                assert not o.ended[0]
                assert o.player[0] == pos["player"]
                assert o.legal()[0, pos["column"]] == 1
                assert o.step_actions([pos["column"]])[0] == 0
        assert o.reward[0].tolist() == game["reward"]
        assert o.legal()[0].sum() == 0


def test_connect_reference_json_state(golden_dir):
    fx = _load(golden_dir, "reference_connect.json")["json"]
    h, w, k = fx["config_args"]
    o = oracle.ConnectOracle(h, w, k, 1)
    for col in fx["state_after_columns"]:
        o.step_actions([col])
    assert o.grid[0].tolist() == fx["state"]["grid"]
    assert int(o.player[0]) == fx["state"]["player"]
    assert int(o.winner[0]) == fx["state"]["winner"]
    assert o.legal()[0, fx["action_column"]] == 1


def _win_case(h, w, k, cols):
    o = oracle.ConnectOracle(h, w, k, 1)
    for c in cols[:-1]:
        assert o.step_actions([c])[0] == 0
        assert o.winner[0] == -1
    assert o.step_actions([cols[-1]])[0] == 0
    return o


def test_connect_unpinned_rules_are_the_standard_ones():
    # vertical, both diagonals, draw, full column: not constrained by any reference test (UNPINNED)
    o = _win_case(6, 7, 4, [0, 1, 0, 1, 0, 1, 0])
    assert o.winner[0] == 0 and o.reward[0].tolist() == [1, -1] and o.plies[0] == 7
    o = _win_case(6, 7, 4, [0, 1, 1, 2, 2, 3, 2, 3, 3, 6, 3])  # rising diagonal for player 0
    assert o.winner[0] == 0
    o = _win_case(6, 7, 4, [3, 2, 2, 1, 1, 0, 1, 0, 0, 6, 0])  # falling diagonal for player 0
    assert o.winner[0] == 0
    o = _win_case(6, 7, 4, [6, 0, 6, 1, 5, 2, 5, 3])  # horizontal for player 1
    assert o.winner[0] == 1 and o.reward[0].tolist() == [-1, 1]
    # 2x2 with k=3 can only be drawn
    o = _win_case(2, 2, 3, [0, 0, 1, 1])
    assert o.winner[0] == 2 and o.reward[0].tolist() == [0, 0] and o.ended[0]
    # full column / ended board are illegal
    o = oracle.ConnectOracle(2, 3, 3, 1)
    assert o.step_actions([0])[0] == 0 and o.step_actions([0])[0] == 0
    assert o.step_actions([0])[0] == -2
    assert o.legal()[0].tolist() == [0, 1, 1]


def test_connect_rollout_is_step_random_repeated():
    n, seed = 257, 0x0123456789ABCDEF
    a = oracle.ConnectOracle(6, 7, 4, n)
    b = oracle.ConnectOracle(6, 7, 4, n)
    total = a.rollout(seed, first_game=1000)
    steps = 0
    while not b.ended.all():
        steps += b.step_random(seed, first_game=1000)
    assert steps == total == int(a.plies.sum())
    np.testing.assert_array_equal(a.grid, b.grid)
    np.testing.assert_array_equal(a.winner, b.winner)
    assert a.ended.all() and 7 <= a.plies.min() and a.plies.max() <= 42
    # sharding invariance: games [1000+100, 1000+200) replayed alone give the same boards
    c = oracle.ConnectOracle(6, 7, 4, 100)
    c.rollout(seed, first_game=1100)
    np.testing.assert_array_equal(c.grid, a.grid[100:200])


# ------------------------------------------------------------------ Bounce


def test_bounce_reference_games(golden_dir):
    fx = _load(golden_dir, "reference_bounce.json")
    n_positions = 0
    for test in fx["tests"]:
        o = oracle.BounceOracle(np.array(test["positions"][0]["grid"], dtype=np.int8), 1)
        for pos in test["positions"]:
            n_positions += 1
            np.testing.assert_array_equal(o.grid[0], np.array(pos["grid"], dtype=np.int8), err_msg=test["name"])
            assert o.player[0] == pos["player"], test["name"]
            assert not o.ended[0]
            sx, sy = pos["source"]
            assert o.targets(0, sx, sy) == {tuple(t) for t in pos["targets"]}, (test["name"], pos["source"])
            listed = [dst for src, dst in o.actions(0) if src == (sx, sy)]
            assert listed == [tuple(t) for t in pos["targets"]]  # canonical (y, x) order, no duplicates
            tx, ty = pos["chosen"]
            assert o.step_actions([[sx, sy, tx, ty]])[0] == 0
        assert bool(o.ended[0]) == test["final"]["has_ended"], test["name"]
        assert len(o.actions(0)) == test["final"]["n_actions"]
        assert o.reward[0].tolist() == test["final"]["reward"], test["name"]
    # SURVEY 8c counts 16 reference positions: these 15 scripted ones (tests/test_bounce.py:92-362) plus the JSON
    # round-trip position (tests/test_bounce.py:365-410), which test_bounce_reference_json_position checks below
    assert n_positions == 15


def test_bounce_reference_json_position(golden_dir):
    fx = _load(golden_dir, "reference_bounce.json")["json"]
    o = oracle.BounceOracle(np.array(fx["config"]["grid"], dtype=np.int8), 1)
    assert o.grid[0].tolist() == fx["state"]["grid"]
    assert int(o.player[0]) == fx["state"]["player"] and int(o.winner[0]) == fx["state"]["winner"]
    src, dst = tuple(fx["action"]["source"]), tuple(fx["action"]["target"])
    assert (src, dst) in o.actions(0)


def test_bounce_illegal_moves_and_non_movable_cells(golden_dir):
    fx = _load(golden_dir, "reference_bounce.json")
    grid = np.array(fx["tests"][1]["positions"][0]["grid"], dtype=np.int8)  # default 9x6 start
    o = oracle.BounceOracle(grid, 1)
    assert o.targets(0, 0, 7) == set()  # far row is not player 0's active row
    assert o.targets(0, 0, 3) == set()  # empty cell
    assert o.step_actions([[0, 1, 0, 3]])[0] == -2  # value-1 piece cannot travel two cells straight
    assert o.step_actions([[0, 7, 0, 6]])[0] == -2
    np.testing.assert_array_equal(o.grid[0], grid)
    with pytest.raises(RuntimeError):
        oracle.BounceOracle(np.array([[1, 0], [0, 0], [0, 0]], dtype=np.int8), 1)  # piece in a goal row


def test_bounce_rollout_is_step_random_repeated(golden_dir):
    fx = _load(golden_dir, "reference_bounce.json")
    grid = np.array(fx["tests"][1]["positions"][0]["grid"], dtype=np.int8)
    n, seed = 96, 0x0123456789ABCDEF
    a = oracle.BounceOracle(grid, n)
    b = oracle.BounceOracle(grid, n)
    total = a.rollout(seed, first_game=5, max_plies=4096)
    steps = 0
    while not b.ended.all():
        steps += b.step_random(seed, first_game=5)
    assert steps == total == int(a.plies.sum())
    np.testing.assert_array_equal(a.grid, b.grid)
    np.testing.assert_array_equal(a.winner, b.winner)
    assert a.ended.all()
    # every finished game is a goal-row win, a block win or a draw
    for i in range(n):
        in_goal = (a.grid[i, 0] > 0).any() or (a.grid[i, -1] > 0).any()
        assert in_goal or a.winner[i] in (0, 1, 2)
    assert (a.count_actions() == 0).all()

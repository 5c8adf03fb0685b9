"""Differential fuzzing of the HIP path against the CPU oracle: random geometries, random start grids, random mixes of
sampled plies, caller-chosen moves, loads and rollouts.  Seeds are fixed; everything is bit-exact or it fails."""

import numpy as np
import pytest

from tests.knobs import knobs

from oracle import oracle

pytestmark = pytest.mark.gpu

import os

SEED = 0x0123456789ABCDEF
# BGS_FUZZ_CASES=N widens the sweep (default: 24 Connect + 32 Bounce cases)
EXTRA = int(os.environ.get("BGS_FUZZ_CASES", "0"))


def same(dev, orc, what):
    np.testing.assert_array_equal(dev.grid, orc.grid, err_msg=f"grid {what}")
    np.testing.assert_array_equal(dev.winner, orc.winner, err_msg=f"winner {what}")
    np.testing.assert_array_equal(dev.plies, orc.plies, err_msg=f"plies {what}")
    np.testing.assert_array_equal(dev.reward, orc.reward, err_msg=f"reward {what}")


def random_connect_geometries(rng, count):
    out = []
    while len(out) < count:
        h, w = int(rng.integers(1, 16)), int(rng.integers(1, 17))
        if w * (h + 1) > 192:
            continue
        out.append((h, w, int(rng.integers(1, 8))))
    return out


@pytest.mark.parametrize("case", range(max(24, EXTRA)))
def test_connect_random_geometry(case):
    from simulator.batch import ConnectBatch

    rng = np.random.default_rng(1000 + case)
    (h, w, k), = random_connect_geometries(rng, 1)
    n = int(rng.integers(1, 3000))
    first = int(rng.integers(0, 1 << 40))
    seed = SEED ^ case
    dev = ConnectBatch(h, w, k, n)
    orc = oracle.ConnectOracle(h, w, k, n)
    dev.set_first_game(first)
    what = f"connect {h}x{w}x{k} n={n}"
    # a few sampled plies, a few caller-chosen ones (some illegal), then a capped and a full rollout
    for _ in range(int(rng.integers(0, 4))):
        dev.step_random(seed)
        orc.step_random(seed, first_game=first)
    for _ in range(int(rng.integers(0, 4))):
        cols = rng.integers(-1, w + 1, size=n).astype(np.int32)
        np.testing.assert_array_equal(dev.step_actions(cols), orc.step_actions(cols), err_msg=what)
    same(dev, orc, what + " after steps")
    np.testing.assert_array_equal(dev.legal, orc.legal(), err_msg=what)
    cap = int(rng.integers(0, h * w + 2))
    dev.rollout(seed, max_plies=cap)
    orc.rollout(seed, first_game=first, max_plies=cap)
    same(dev, orc, what + f" capped at {cap}")
    dev.rollout(seed)
    orc.rollout(seed, first_game=first)
    same(dev, orc, what + " finished")
    assert dev.has_ended.all()
    # from the initial state, with a fresh seed, into the same handle
    dev.rollout(seed + 1, from_initial=True)
    orc.reset()
    orc.rollout(seed + 1, first_game=first)
    same(dev, orc, what + " from initial")
    # reload the final boards and check that the device derives the same verdicts from the grids alone
    again = ConnectBatch(h, w, k, n)
    assert (again.write_state(orc.grid) == 0).all()
    same(again, orc, what + " reloaded")


def random_bounce_grid(rng):
    while True:
        h, w = int(rng.integers(3, 12)), int(rng.integers(1, 13))
        if h * w <= 64:
            break
    grid = np.zeros((h, w), dtype=np.int8)
    density = rng.uniform(0.05, 0.7)
    max_value = int(rng.choice([1, 2, 3, 3, 3, 5, 9, 15]))
    for y in range(1, h - 1):
        for x in range(w):
            if rng.random() < density:
                grid[y, x] = int(rng.integers(1, max_value + 1))
    return grid


@pytest.mark.parametrize("case", range(max(32, EXTRA)))
def test_bounce_random_grid(case):
    from simulator.batch import BounceBatch

    rng = np.random.default_rng(5000 + case)
    grid = random_bounce_grid(rng)
    n = int(rng.integers(1, 600))
    first = int(rng.integers(0, 1 << 40))
    seed = SEED ^ (case << 8)
    dev = BounceBatch(grid, n)
    orc = oracle.BounceOracle(grid, n)
    dev.set_first_game(first)
    what = f"bounce {grid.shape} case {case}"
    same(dev, orc, what + " after reset")
    np.testing.assert_array_equal(dev.action_count, orc.count_actions(), err_msg=what)
    for ply in range(int(rng.integers(0, 6))):
        dev.step_random(seed)
        orc.step_random(seed, first_game=first)
        same(dev, orc, what + f" step {ply}")
    np.testing.assert_array_equal(dev.action_count, orc.count_actions(), err_msg=what)
    # exhaustive target sets of a few boards
    masks = dev.targets
    width = dev.width
    for i in range(0, n, max(1, n // 7)):
        got = []
        row = int(masks[i, width])
        for x in range(width):
            m = int(masks[i, x])
            got += [((x, row), (c % width, c // width)) for c in range(64) if (m >> c) & 1]
        assert got == orc.actions(i), f"{what} board {i}"
    # caller-chosen moves: the oracle's own legal moves, some garbage, some skips
    moves = np.full((n, 4), -1, dtype=np.int32)
    for i in range(n):
        acts = orc.actions(i)
        r = rng.random()
        if acts and r < 0.6:
            (sx, sy), (tx, ty) = acts[rng.integers(len(acts))]
            moves[i] = [sx, sy, tx, ty]
        elif r < 0.8:
            moves[i] = rng.integers(0, 12, size=4)
    np.testing.assert_array_equal(dev.step_actions(moves), orc.step_actions(moves), err_msg=what)
    same(dev, orc, what + " after chosen moves")
    cap = int(rng.integers(0, 40))
    dev.rollout(seed, max_plies=cap)
    orc.rollout(seed, first_game=first, max_plies=cap)
    same(dev, orc, what + f" capped at {cap}")
    dev.rollout(seed, max_plies=600)
    orc.rollout(seed, first_game=first, max_plies=600)
    same(dev, orc, what + " rolled out")
    dev.rollout(seed + 3, max_plies=600, from_initial=True)
    orc.reset()
    orc.rollout(seed + 3, first_game=first, max_plies=600)
    same(dev, orc, what + " from initial")
    again = BounceBatch(grid, n)
    assert (again.write_state(orc.grid, orc.player, orc.winner, orc.plies) == 0).all()
    same(again, orc, what + " reloaded")


def random_piece_list_grid(rng):
    """A start grid the piece-list kernel takes: at most 8 columns and 16 pieces (values 1..15, a few cells each)."""
    while True:
        h, w = int(rng.integers(3, 12)), int(rng.integers(1, 9))
        if h * w <= 64:
            break
    grid = np.zeros((h, w), dtype=np.int8)
    cells = [(y, x) for y in range(1, h - 1) for x in range(w)]
    pieces = int(rng.integers(1, min(16, len(cells)) + 1))
    max_value = int(rng.choice([1, 2, 3, 3, 3, 4, 6, 15]))
    for k in rng.choice(len(cells), size=pieces, replace=False):
        y, x = cells[int(k)]
        grid[y, x] = int(rng.integers(1, max_value + 1))
    return grid


@pytest.mark.parametrize("case", range(max(24, EXTRA)))
def test_bounce_piece_list_random_grid(case):
    """K3p (one lane per board on the piece list; the 8-, 12- and 16-piece instantiations, the bulk + tail plan for caps
    beyond 768 plies, parked boards) against the oracle on random start grids, batch sizes around the wave / workgroup
    boundaries, random first-game offsets."""
    from simulator.batch import BounceBatch

    rng = np.random.default_rng(9000 + case)
    grid = random_piece_list_grid(rng)
    n = int(rng.choice([1, 63, 64, 65, 255, 257, 1000, 2500, 6000]))
    first = int(rng.integers(0, 1 << 40))
    cap = int(rng.choice([0, 1, 7, 60, 500, 769, 2000]))
    old = knobs.get("bounce_group")
    knobs["bounce_group"] = "1"
    try:
        dev = BounceBatch(grid, n)
        orc = oracle.BounceOracle(grid, n)
        dev.set_first_game(first)
        what = f"piece-list bounce {grid.shape} {int((grid > 0).sum())} pieces n={n} cap={cap} case {case}"
        dev.rollout(SEED ^ case, max_plies=cap, from_initial=True)
        total = orc.rollout(SEED ^ case, first_game=first, max_plies=cap)
        same(dev, orc, what)
        assert dev.steps == total, what
        dev.close()
    finally:
        if old is None:
            del knobs["bounce_group"]
        else:
            knobs["bounce_group"] = old

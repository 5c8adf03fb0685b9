#!/usr/bin/env python3
"""Round-6 soak: what the round added, on random inputs, against the oracle and against the kernels it replaced.
  1. Bounce, default board: the compile-time geometry (K3p + K3w instantiated on DefaultBounceGeom, K3p writing the tail's work
     list itself) against the run-time record (experiment bounce_static_geom=0) -- random seeds, batch sizes, launch hints,
     ply caps around the bulk caps, first-game offsets; rewards / plies / status / grids identical, and the oracle's on a sample;
  2. Connect under the strict RNG contract: random geometries, sizes, entry states, against the oracle.
    python3 tools/r6_soak.py [minutes]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "board-game-simulator-python_amd")]
from tests.knobs import knobs
import numpy as np
from oracle import oracle
from simulator.batch import BounceBatch, ConnectBatch

minutes = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
rng = np.random.default_rng(20261005)
g = np.zeros((9, 6), dtype=np.int8); g[1] = g[7] = [1, 2, 3, 3, 2, 1]
t_end = time.time() + 60 * minutes * 0.6
rounds = boards = 0
while time.time() < t_end:
    n = int(rng.choice([1 << 15, 40000, 1 << 16, 100003, 1 << 17, 1 << 18]))
    hint = int(rng.choice([1, 4, 8, 20]))
    if n < 131072 and hint < 4:
        hint = 4   # (K3p takes batches below 2^17 only with four or more launches in flight)
    cap = int(rng.choice([4096, 4096, 700, 161, 160, 159, 129, 128, 127, 81, 80, 79, 40, 9, 4, 3]))
    seed = int(rng.integers(1 << 62))
    first = int(rng.integers(1 << 40))
    out = []
    for static in ("1", "0"):
        knobs["bounce_static_geom"] = static
        b = BounceBatch(g, n)
        b.set_launches_in_flight(hint)
        b.set_first_game(first)
        b.rollout(seed, max_plies=cap, from_initial=True)
        out.append((b.reward.copy(), b.plies.copy(), b.winner.copy(), b.grid.copy(), b.steps))
        b.close()
    for k in range(4):
        assert np.array_equal(out[0][k], out[1][k]), ("static vs run-time geometry", n, hint, cap, seed, first, k)
    assert out[0][4] == out[1][4]
    m = min(n, 1 << 14)
    orc = oracle.BounceOracle(g, m)
    orc.rollout(seed, first_game=first, max_plies=cap)
    assert np.array_equal(out[0][0][:m], orc.reward) and np.array_equal(out[0][1][:m], orc.plies) and np.array_equal(out[0][3][:m], orc.grid), \
        ("oracle", n, hint, cap, seed, first)
    rounds += 1
    boards += n
print(f"bounce: {rounds} rounds, {boards} boards: static geometry == run-time record == oracle (first 2^14)", flush=True)
knobs.pop("bounce_static_geom", None)

t_end = time.time() + 60 * minutes * 0.4
rounds = boards = 0
while time.time() < t_end:
    h, w = int(rng.integers(1, 16)), int(rng.integers(1, 17))
    if w * (h + 1) > 192 or rng.integers(8) == 0:
        h, w = (6, 7) if rng.integers(2) else (12, 13)
    k = int(rng.integers(2, 6))
    n = int(rng.choice([64, 1000, 4999, 30000, 1 << 16]))
    seed, first = int(rng.integers(1 << 62)), int(rng.integers(1 << 40))
    dev = ConnectBatch(h, w, k, n)
    dev.set_rng_contract("per-ply")
    dev.set_first_game(first)
    orc = oracle.ConnectOracle(h, w, k, n, per_ply=True)
    how = int(rng.integers(3))
    cap = 2**31 - 1
    if how == 1:
        pre = int(rng.integers(1, 6))
        dev.step_random(seed ^ 9, plies=pre)
        for _ in range(pre):
            orc.step_random(seed ^ 9, first_game=first)
    if how == 2:
        cap = int(rng.integers(0, h * w + 2))
    dev.rollout(seed, max_plies=cap, from_initial=(how != 1))
    orc.rollout(seed, first_game=first, max_plies=cap)
    assert np.array_equal(dev.grid, orc.grid) and np.array_equal(dev.reward, orc.reward) and np.array_equal(dev.plies, orc.plies), \
        ("strict contract", h, w, k, n, how, cap, seed, first)
    dev.close()
    rounds += 1
    boards += n
print(f"connect, strict contract: {rounds} rounds, {boards} boards == oracle", flush=True)

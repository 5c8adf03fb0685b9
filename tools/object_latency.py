#!/usr/bin/env python3
"""Cost of the object API (`State.actions` -> `random.choice` -> `Action.sample_next_state`, the reference's README loop)
on the HIP path: microseconds per transition on one thread, and transitions per second from a pool of 8 threads (the
reference's arena runs 8 boards from a ThreadPoolExecutor, textual/examples/arena.py:53).  Prints one JSON object."""
import json, os, random, sys, time
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "board-game-simulator-python_amd")]
import numpy as np
from simulator.game.bounce import Config as BounceConfig
from simulator.game.connect import Config as ConnectConfig

grid = np.zeros((9, 6), dtype=np.int64)
grid[1] = grid[7] = [1, 2, 3, 3, 2, 1]


def play(config, games, cap, seed):
    rng = random.Random(seed)
    n = 0
    for _ in range(games):
        s = config.sample_initial_state()
        p = 0
        while not s.has_ended and p < cap:
            s = rng.choice(s.actions).sample_next_state()
            n += 1
            p += 1
    return n


out = {}
for name, config, games, cap in (("connect_6x7x4", ConnectConfig(6, 7, 4), 30, 1000), ("bounce_default", BounceConfig(grid), 6, 150)):
    play(config, 1, 20, 0)  # engine + kernels warm
    t0 = time.perf_counter()
    n = play(config, games, cap, 1)
    one = (time.perf_counter() - t0) / n
    with ThreadPoolExecutor(8) as pool:
        list(pool.map(lambda k: play(config, 1, 20, k), range(8)))  # one engine per thread
        t0 = time.perf_counter()
        counts = list(pool.map(lambda k: play(config, games, cap, 100 + k), range(8)))
        dt = time.perf_counter() - t0
    out[name] = {"us_per_transition_one_thread": one * 1e6, "transitions_per_s_one_thread": 1 / one,
                 "transitions_per_s_8_threads": sum(counts) / dt, "speedup_8_threads": sum(counts) / dt * one}
print(json.dumps(out, indent=1))

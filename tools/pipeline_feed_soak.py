#!/usr/bin/env python3
"""Soak of RolloutPipeline.run on the library's feeder thread (bgs_pipeline_feed / _release / _wait): many short-lived
pipelines of random depth and size, runs of random length, consumers that stop early, submit() and run() mixed on one
pipeline, every yielded step checked against the oracle's replay of its seed.

    python3 tools/pipeline_feed_soak.py [iterations]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "board-game-simulator-python_amd")]
import numpy as np
from oracle import oracle
from simulator.batch import BounceBatch, ConnectBatch
from simulator.pipeline import RolloutPipeline

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 120
rng = np.random.default_rng(20261005)
grid = np.zeros((9, 6), dtype=np.int8)
grid[1] = grid[7] = [1, 2, 3, 3, 2, 1]
t0 = time.perf_counter()
checked = steps = 0
for it in range(iters):
    bounce = it % 4 == 3
    n = int(rng.integers(64, 3000)) & ~1
    depth = int(rng.integers(1, 5))
    cls, args, cap = (BounceBatch, (grid,), 300) if bounce else (ConnectBatch, (6, 7, 4), 2**31 - 1)
    orc = oracle.BounceOracle(grid, n) if bounce else oracle.ConnectOracle(6, 7, 4, n)
    def expect(seed):
        orc.reset()
        orc.rollout(seed, first_game=0, max_plies=cap) if bounce else orc.rollout(seed, first_game=0)
        return np.asarray(orc.reward).copy()
    with RolloutPipeline(cls, args, n, depth=depth, max_plies=cap) as pipe:
        for rnd in range(int(rng.integers(1, 4))):
            count = int(rng.integers(0, 40))
            seeds = [int(s) for s in rng.integers(1, 2**62, size=count)]
            stop_at = int(rng.integers(0, count + 1)) if rng.random() < 0.4 else count
            first = None
            for k, (step, rewards) in enumerate(pipe.run(seeds)):
                first = step if first is None else first
                assert step == first + k
                if k % 7 == 0:
                    np.testing.assert_array_equal(rewards, expect(seeds[k]), err_msg=f"iteration {it}, round {rnd}, step {step}")
                    checked += 1
                steps += 1
                if k + 1 == stop_at:
                    break
            if rng.random() < 0.5:   # the one-step-at-a-time calls on the same pipeline, after a run
                seed = int(rng.integers(1, 2**62))
                step = pipe.submit(seed)
                np.testing.assert_array_equal(pipe.result(step), expect(seed), err_msg=f"iteration {it}: submit after run")
                checked += 1
print(f"{iters} pipelines, {steps} steps yielded, {checked} checked against the oracle ({time.perf_counter() - t0:.0f} s)")

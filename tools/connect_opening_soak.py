"""Soak of K2o against K2a: the same seeds on a batch played by the kernel without the opening stage
(rollout_opening=0) and on batches played by K2o with 1..4 opening blocks and different chunk sizes must give
identical rewards, boards and step counts."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "board-game-simulator-python_amd")]
from tests.knobs import knobs  # BGS_EXPERIMENT ("name=value;...") as a mapping
import numpy as np
import torch
from simulator.batch import ConnectBatch

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
seeds = int(sys.argv[2]) if len(sys.argv) > 2 else 200

def make(opening, chunk):
    knobs["rollout_opening"] = str(opening)
    if chunk:
        knobs["rollout_chunk"] = str(chunk)
    else:
        knobs.pop("rollout_chunk", None)
    return ConnectBatch(6, 7, 4, n, use_torch=True)

ref = make(0, 0)
variants = {"open3": make(3, 0), "open2/chunk256": make(2, 256), "open4/chunk1024": make(4, 1024), "open1/chunk64": make(1, 64),
            "open3/chunk4096": make(3, 4096)}
t0 = time.perf_counter()
for s in range(seeds):
    seed = 0x13579BDF02468ACE + 104729 * s
    ref.reset_steps(); ref.rollout(seed, from_initial=True)
    want, steps = ref.reward_copy_tensor(), ref.steps
    grid = ref.grid[: 1 << 14] if s % 40 == 0 else None
    for name, b in variants.items():
        b.reset_steps(); b.rollout(seed, from_initial=True)
        assert b.steps == steps, (name, s, b.steps, steps)
        assert torch.equal(b.reward_copy_tensor(), want), (name, s, "reward")
        if grid is not None:
            assert np.array_equal(b.grid[: 1 << 14], grid), (name, s, "grid")
print(f"{seeds} seeds x {len(variants)} K2o variants of {n} boards agree with K2a ({time.perf_counter() - t0:.0f} s)")

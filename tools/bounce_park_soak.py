"""Soak of the flat Bounce kernel's shared drain: the same seeds on a batch whose waves drain alone (bounce_park=0)
and on batches that park / adopt (threshold 32, several waves-per-launch settings) must give identical boards, plies,
rewards and step counts -- a lost or doubly played parked board would show up here."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "board-game-simulator-python_amd")]
from tests.knobs import knobs  # BGS_EXPERIMENT ("name=value;...") as a mapping
import numpy as np
import torch
from simulator.batch import BounceBatch

g = np.zeros((9, 6), dtype=np.int8); g[1] = g[7] = [1, 2, 3, 3, 2, 1]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 16
seeds = int(sys.argv[2]) if len(sys.argv) > 2 else 150

def make(park, waves):
    knobs["bounce_group"] = "1"
    knobs["bounce_park"] = str(park)
    if waves:
        knobs["bounce_flat_waves"] = str(waves)
    else:
        knobs.pop("bounce_flat_waves", None)
    return BounceBatch(g, n, use_torch=True)

ref = make(0, 0)
variants = {"park32": make(32, 0), "park32/waves64": make(32, 64), "park32/waves7": make(32, 7), "park5/waves300": make(5, 300)}
t0 = time.perf_counter()
for s in range(seeds):
    seed = 0xABCDEF0123 + 7919 * s
    ref.reset_steps(); ref.rollout(seed, max_plies=4096, from_initial=True)
    want = (ref.reward_copy_tensor(), torch.from_numpy(ref.plies.astype(np.int32)), ref.steps)
    grid = ref.grid if s % 25 == 0 else None
    for name, b in variants.items():
        b.reset_steps(); b.rollout(seed, max_plies=4096, from_initial=True)
        assert b.steps == want[2], (name, s, b.steps, want[2])
        assert torch.equal(b.reward_copy_tensor(), want[0]), (name, s, "reward")
        assert np.array_equal(b.plies.astype(np.int32), want[1].numpy()), (name, s, "plies")
        if grid is not None:
            assert np.array_equal(b.grid, grid), (name, s, "grid")
print(f"{seeds} seeds x {len(variants)} variants of {n} boards agree with the unshared drain ({time.perf_counter() - t0:.0f} s)")

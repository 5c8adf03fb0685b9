"""Soak of the piece-list Bounce kernel (K3p) against the flat cell-search kernel (K3f) on the device: the same seeds on
2^16 default boards through K3f (one launch, waves drain alone) and through K3p in several settings (default plan with
the tail pass, one launch, few waves, small parking threshold) must give identical rewards, plies, step counts and
(sampled) grids -- a board lost or played twice in the drain, or a wrong target set, would show up here."""
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
KEYS = ("bounce_group", "bounce_pieces", "bounce_plan", "bounce_park", "bounce_flat_waves", "bounce_pool")

def make(**env):
    for k in KEYS:
        knobs.pop(k, None)
    knobs["bounce_group"] = "1"
    knobs.update({k: str(v) for k, v in env.items()})
    return BounceBatch(g, n, use_torch=True)

ref = make(bounce_pieces=0, bounce_plan="single", bounce_park=0)
variants = {"K3p default (bulk + tail)": make(), "K3p single launch": make(bounce_plan="single"),
            "K3p 64 waves": make(bounce_flat_waves=64), "K3p park 5 / 300 waves": make(bounce_park=5, bounce_flat_waves=300),
            "K3p tail at 96": make(bounce_plan="96:1,0:8"),
            # round 4: the device-wide pool of parked boards is on by default (every variant above); off, and in the shapes that
            # stress it: many small workgroups' worth of waves, a parking threshold of 32 and of 3
            "K3p no device-wide pool": make(bounce_pool=0), "K3p pool, 1024 waves, park 32": make(bounce_flat_waves=1024, bounce_park=32),
            "K3p pool, 128 waves, park 3, single launch": make(bounce_flat_waves=128, bounce_park=3, bounce_plan="single")}
# the launch shapes of bounce_shape(): one launch at a time (the default above), 8 and 16 in flight
for hint in (8, 16):
    variants[f"K3p shape of {hint} in flight"] = make()
    variants[f"K3p shape of {hint} in flight"].set_launches_in_flight(hint)
t0 = time.perf_counter()
for s in range(seeds):
    seed = 0xABCDEF0123 + 7919 * s
    ref.reset_steps(); ref.rollout(seed, max_plies=4096, from_initial=True)
    want = (ref.reward_copy_tensor(), ref.plies.astype(np.int32), ref.steps)
    grid = ref.grid if s % 25 == 0 else None
    for name, b in variants.items():
        b.reset_steps(); b.rollout(seed, max_plies=4096, from_initial=True)
        assert b.steps == want[2], (name, s, b.steps, want[2])
        assert torch.equal(b.reward_copy_tensor(), want[0]), (name, s, "reward")
        assert np.array_equal(b.plies.astype(np.int32), want[1]), (name, s, "plies")
        if grid is not None:
            assert np.array_equal(b.grid, grid), (name, s, "grid")
print(f"{seeds} seeds x {len(variants)} K3p variants of {n} boards agree with K3f ({time.perf_counter() - t0:.0f} s)")

"""Soak of the Connect(12,13,5) rollout's opening launch: the same seeds through the LDS-staged kernel from the empty board
(rollout_opening=0), through opening launch + rollout kernel (default) and through the register kernel K2b must give
identical planes, status, rewards and step counts on 2^18 boards."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "board-game-simulator-python_amd")]
from tests.knobs import knobs  # BGS_EXPERIMENT ("name=value;...") as a mapping
import numpy as np
import torch
from simulator.batch import ConnectBatch
from simulator.game import _abi

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 18
seeds = int(sys.argv[2]) if len(sys.argv) > 2 else 100

def make(**env):
    for k in ("rollout_opening", "rollout_no_lds"):
        knobs.pop(k, None)
    knobs.update({k: str(v) for k, v in env.items()})
    return ConnectBatch(12, 13, 5, n, use_torch=True)

ref = make(rollout_opening=0)
variants = {"opening launch": make(), "register kernel": make(rollout_no_lds=1)}
t0 = time.perf_counter()
for s in range(seeds):
    seed = 0x5EED0000 + 104729 * s
    first = (s * 7919) << 20
    for b in [ref, *variants.values()]:
        b.set_first_game(first); b.reset_steps(); b.rollout(seed, from_initial=True)
    want = (ref._arena_view(_abi.BUF_PLANES).clone(), ref.status_tensor().clone(), ref.reward_copy_tensor(), ref.steps)
    for name, b in variants.items():
        assert b.steps == want[3], (name, s, b.steps, want[3])
        assert torch.equal(b._arena_view(_abi.BUF_PLANES), want[0]), (name, s, "planes")
        assert torch.equal(b.status_tensor(), want[1]), (name, s, "status")
        assert torch.equal(b.reward_copy_tensor(), want[2]), (name, s, "reward")
print(f"{seeds} seeds x {len(variants)} variants of {n} Connect(12,13,5) boards agree with the kernel from the empty board ({time.perf_counter() - t0:.0f} s)")

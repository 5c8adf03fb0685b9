#!/usr/bin/env python3
"""Policy-driven stepping: an external policy on the GPU picks every move.  Per ply: legal mask on
the device (bgs_export_device 'l') -> torch policy -> bgs_step_actions with a device action tensor.  No host round
trip inside the loop.  Prints env-steps/s for a uniform-random torch policy on Connect4(6,7,4), 2^20 boards."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "board-game-simulator-python_amd")]
import torch
from simulator.batch import ConnectBatch

n = 1 << 20
batch = ConnectBatch(6, 7, 4, n, use_torch=True)
legal = torch.empty((n, 7), dtype=torch.uint8, device="cuda")
gen = torch.Generator(device="cuda").manual_seed(1)

def play_all():
    batch.reset()
    plies = 0
    for _ in range(42):
        batch.legal_tensor(legal)
        # uniform over legal columns: random scores, illegal columns masked out; boards without a legal column skip
        scores = torch.rand((n, 7), device="cuda", generator=gen) * legal
        col = scores.argmax(dim=1).to(torch.int32)
        col = torch.where(legal.any(dim=1), col, torch.full_like(col, -1))
        batch.step_actions(col, want_status=False)
        plies += 1
    torch.cuda.synchronize()

play_all()
t0 = time.perf_counter()
play_all()
dt = time.perf_counter() - t0
print(f"policy loop: {batch.steps} env-steps in {dt*1e3:.2f} ms = {batch.steps/dt/1e9:.2f} G env-steps/s "
      f"(42 plies x [legal mask + torch policy + step_actions]); all ended: {bool(batch.has_ended.all())}")

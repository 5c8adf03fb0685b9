#!/usr/bin/env python3
"""Six step_random launches on 2^24 Connect4 boards (256 MiB of planes: larger than the Infinity Cache), for the
rocprofv3 HBM counters of the per-ply kernel K1."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "board-game-simulator-python_amd")]
import torch
from simulator.batch import ConnectBatch
b = ConnectBatch(6, 7, 4, 1 << 24, use_torch=True)
for i in range(6):
    b.step_random(0x0123456789ABCDEF)
torch.cuda.synchronize()
print("steps", b.steps)

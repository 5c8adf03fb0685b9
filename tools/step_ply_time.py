#!/usr/bin/env python3
"""Microseconds per launch of the per-ply kernel (bgs_step_random, one ply) on Connect4(6,7,4): 40 back-to-back launches
between two events, for a list of batch sizes.  BGS_STEP_NT_FROM sets the batch size from which the kernel uses
non-temporal accesses."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "board-game-simulator-python_amd")]
import torch
from simulator.batch import ConnectBatch

out = {}
for n in [int(a) for a in sys.argv[1:]] or [1 << 20, 1 << 22, 1 << 24]:
    b = ConnectBatch(6, 7, 4, n, use_torch=True)
    best = None
    for rep in range(3):
        b.reset()
        for _ in range(4):
            b.step_random(5)
        torch.cuda.synchronize()
        a, z = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for k in range(16):   # plies 4 .. 19: nearly every board still running
            b.step_random(5)
        z.record(); torch.cuda.synchronize()
        us = a.elapsed_time(z) * 1e3 / 16
        best = us if best is None or us < best else best
    out[n] = {"us_per_ply": best, "boards_per_s": n / best * 1e6, "GBps_at_25B": n * 25 / best * 1e-3}
    b.close()
print(json.dumps(out))

#!/usr/bin/env python3
"""Connect(12,13,5), 8 launches in flight: plain rollout vs rollout that writes its outcome codes (into DEVICE memory, no
sink): is the hand-over's 9 % the kernel variant or the sink?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "board-game-simulator-python_amd")]
from simulator.game import _abi
_abi.request_hardware_queues()
import torch
from simulator.batch import ConnectBatch
N, D, SEED = 1 << 18, 8, 0x0123456789ABCDEF
streams = [torch.cuda.Stream() for _ in range(D)]
batches, bufs = [], []
for s in streams:
    with torch.cuda.stream(s):
        b = ConnectBatch(12, 13, 5, N, use_torch=True); b.set_launches_in_flight(D); batches.append(b)
        bufs.append(torch.zeros((N + 63) // 64 * 16, dtype=torch.uint8, device="cuda"))
def run(codes, reps=320):
    for b in batches: b.reset_steps()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(reps):
        k = i % D
        if codes: batches[k].rollout_outcomes_tensor(bufs[k], SEED + i, from_initial=True)
        else: batches[k].rollout(SEED + i, from_initial=True)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    return sum(b.steps for b in batches) / dt
run(False, 64); run(True, 64)
for _ in range(3):
    print("plain %.1f G/s   with codes to device memory %.1f G/s" % (run(False) / 1e9, run(True) / 1e9))

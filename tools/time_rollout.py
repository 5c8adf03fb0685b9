#!/usr/bin/env python3
"""Time the fused rollout kernel for several launch geometries (BGS_ROLLOUT_WPS) on one GPU; checks that every
geometry produces identical boards (results must not depend on the launch geometry)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "board-game-simulator-python_amd")]
import numpy as np
import torch
from simulator.batch import ConnectBatch, BounceBatch

SEED = 0x0123456789ABCDEF
def run(make, label, wps_list, reps=20, **kw):
    ref = None
    for wps in wps_list:
        os.environ["BGS_ROLLOUT_WPS"] = str(wps)
        b = make()
        for i in range(3):
            b.rollout(SEED + i, from_initial=True, **kw)
        b.reset_steps()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(reps):
            b.rollout(SEED + i, from_initial=True, **kw)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        steps = b.steps / reps
        b.rollout(SEED, from_initial=True, **kw)
        sig = (b.reward.tobytes(), b.plies.tobytes())
        if ref is None:
            ref = sig
        print(f"{label} wps={wps}: {ms*1e3:9.1f} us/launch  {steps/ms/1e6:9.2f} G env-steps/s  same_as_first={sig == ref}", flush=True)
        b.close()

def run_pipelined(make, label, wps_list, streams_list, reps=24, **kw):
    """K launches round-robin over S batches on S streams (double / quad buffering)."""
    for wps in wps_list:
        os.environ["BGS_ROLLOUT_WPS"] = str(wps)
        for ns in streams_list:
            streams = [torch.cuda.Stream() for _ in range(ns)]
            batches = []
            for s_ in streams:
                with torch.cuda.stream(s_):
                    batches.append(make())
            for i in range(2 * ns):
                batches[i % ns].rollout(SEED + i, from_initial=True, **kw)
            torch.cuda.synchronize()
            for b in batches:
                b.reset_steps()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(reps):
                batches[i % ns].rollout(SEED + i, from_initial=True, **kw)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            steps = sum(b.steps for b in batches)
            print(f"{label} wps={wps} streams={ns}: {dt/reps*1e6:9.1f} us/step  {steps/dt/1e9:9.2f} G env-steps/s", flush=True)
            for b in batches:
                b.close()

which = sys.argv[1] if len(sys.argv) > 1 else "c4"
wps = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [8, 6, 4, 3, 2]
if which == "c4":
    lg = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    run(lambda: ConnectBatch(6, 7, 4, 1 << lg, use_torch=True), f"connect4 6x7x4 n=2^{lg}", wps)
elif which == "c4p":
    streams = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [1, 2, 4]
    run_pipelined(lambda: ConnectBatch(6, 7, 4, 1 << 20, use_torch=True), "connect4 2^20 pipelined", wps, streams)
elif which == "c5":
    run(lambda: ConnectBatch(12, 13, 5, 1 << 18, use_torch=True), "connect 12x13x5 n=2^18", wps)
elif which == "bounce":
    g = np.zeros((9, 6), dtype=np.int8); g[1] = g[7] = [1, 2, 3, 3, 2, 1]
    run(lambda: BounceBatch(g, 1 << 18, use_torch=True), "bounce 9x6 n=2^18", wps, reps=3, max_plies=4096)

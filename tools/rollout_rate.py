#!/usr/bin/env python3
"""Fused-rollout throughput of one BASELINE configuration on one GPU: one launch at a time (latency) and D batches in
flight on D streams (throughput), env-steps counted on the device.

    python tools/rollout_rate.py connect6x7 | connect12x13 | bounce  [--depth D] [--reps R] [--batch N]
Prints one JSON object."""
import argparse, json, os, sys, time
_depth = 3
if "--depth" in sys.argv:
    _depth = int(sys.argv[sys.argv.index("--depth") + 1])
# The HIP runtime multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4): more batches in
# flight than queues do not overlap.  Must be set before the runtime initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", str(max(4, min(2 * _depth, 32)) if _depth > 4 else 4))
# Bounce with many batches in flight: fewer, longer-lived waves per launch (less drain per batch; the launches fill the chip
# together).  Must be set before the library reads it at batch creation.
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "board-game-simulator-python_amd")]
if "--bounce-waves" in sys.argv:
    from tests.knobs import knobs  # BGS_EXPERIMENT ("name=value;...") as a mapping

    knobs["bounce_flat_waves"] = sys.argv[sys.argv.index("--bounce-waves") + 1]
import numpy as np
import torch
from simulator.batch import BounceBatch, ConnectBatch

SEED = 0x0123456789ABCDEF
ap = argparse.ArgumentParser()
ap.add_argument("config", choices=("connect6x7", "connect12x13", "bounce"))
ap.add_argument("--depth", type=int, default=3)
ap.add_argument("--reps", type=int, default=24)
ap.add_argument("--batch", type=int, default=0)
ap.add_argument("--max-plies", type=int, default=4096)
ap.add_argument("--hint", type=int, default=0, help="launches-in-flight hint for every run (0: the run's own depth); counters of the pipelined launch shape can then be taken one launch at a time")
ap.add_argument("--bounce-waves", type=int, default=0, help="BGS_EXPERIMENT bounce_flat_waves for this run (0: library default)")
args = ap.parse_args()

def make():
    if args.config == "connect6x7":
        return ConnectBatch(6, 7, 4, args.batch or 1 << 20, use_torch=True), {}
    if args.config == "connect12x13":
        return ConnectBatch(12, 13, 5, args.batch or 1 << 18, use_torch=True), {}
    g = np.zeros((9, 6), dtype=np.int8); g[1] = g[7] = [1, 2, 3, 3, 2, 1]
    return BounceBatch(g, args.batch or 1 << 18, use_torch=True), {"max_plies": args.max_plies}

streams = [torch.cuda.Stream() for _ in range(args.depth)]
batches = []
for s in streams:
    with torch.cuda.stream(s):
        b, kw = make()
        batches.append(b)

def run(depth, reps):
    for b in batches[:depth]:
        b.reset_steps()
        b.set_launches_in_flight(args.hint or depth)  # (what the rollout executor tells its batches)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(reps):
        batches[i % depth].rollout(SEED + i, from_initial=True, **kw)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    steps = sum(b.steps for b in batches[:depth])
    return {"s_per_batch": dt / reps, "env_steps_per_batch": steps / reps, "env_steps_per_s": steps / dt}

for b in batches:
    b.rollout(SEED, from_initial=True, **kw)
torch.cuda.synchronize()
out = {"config": args.config, "batch": batches[0].n, "one_launch_at_a_time": run(1, max(4, args.reps // 3)),
       f"{args.depth}_in_flight": run(args.depth, args.reps),
       "env": {k: v for k, v in os.environ.items() if k.startswith("BGS_") or k == "GPU_MAX_HW_QUEUES"}}
print(json.dumps(out))

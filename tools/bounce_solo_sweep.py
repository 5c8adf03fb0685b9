#!/usr/bin/env python3
"""Bounce default, 2^18 boards, ONE launch at a time (and D in flight): launch shape x bulk cap x park threshold, every
setting a fresh batch in this process (the experiment knobs are read when a batch is created).
    python tools/bounce_solo_sweep.py [--depth 1] [--reps 16]  name=v1,v2 name=v1,v2 ...     (BGS_EXPERIMENT names)"""
import argparse, itertools, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "board-game-simulator-python_amd")]
ap = argparse.ArgumentParser()
ap.add_argument("--depth", type=int, default=1)
ap.add_argument("--reps", type=int, default=16)
ap.add_argument("--hint", type=int, default=0)
ap.add_argument("--batch", type=int, default=1 << 18)
ap.add_argument("axes", nargs="*")
args = ap.parse_args()
import numpy as np
import torch
from tests.knobs import knobs
from simulator.batch import BounceBatch

SEED = 0x0123456789ABCDEF
g = np.zeros((9, 6), dtype=np.int8); g[1] = g[7] = [1, 2, 3, 3, 2, 1]
axes = [(a.split("=", 1)[0], a.split("=", 1)[1].split("|")) for a in args.axes]
streams = [torch.cuda.Stream() for _ in range(args.depth)]
for combo in itertools.product(*[vals for _, vals in axes]) if axes else [()]:
    for (name, _), v in zip(axes, combo):
        if v == "-":
            knobs.pop(name, None)
        else:
            knobs[name] = v
    batches = []
    for s in streams:
        with torch.cuda.stream(s):
            b = BounceBatch(g, args.batch, use_torch=True)
            b.set_launches_in_flight(args.hint or args.depth)
            batches.append(b)
    for b in batches:
        b.rollout(SEED, max_plies=4096, from_initial=True)
    torch.cuda.synchronize()
    best = None
    for rnd in range(3):
        for b in batches:
            b.reset_steps()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.reps):
            batches[i % args.depth].rollout(SEED + i, max_plies=4096, from_initial=True)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        steps = sum(b.steps for b in batches)
        rate = steps / dt
        if best is None or rate > best[0]:
            best = (rate, dt / args.reps * 1e3)
    print(json.dumps({"setting": dict(zip([n for n, _ in axes], combo)), "env_steps_per_s": best[0], "ms_per_launch": best[1]}), flush=True)
    for b in batches:
        b.close()

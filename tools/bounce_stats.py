#!/usr/bin/env python3
"""Loop statistics of the piece-list Bounce rollout (a library built with -DBGS_BOUNCE_STATS, BGS_LIBRARY=...): wave
iterations, iterations with a search, iterations while draining, closure-loop trips -- per launch of 2^18 boards."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "board-game-simulator-python_amd")]
import numpy as np, torch
from simulator.batch import BounceBatch
from simulator.game import _abi
g = np.zeros((9, 6), dtype=np.int8); g[1] = g[7] = [1, 2, 3, 3, 2, 1]
n = 1 << 18
b = BounceBatch(g, n, use_torch=True)
if os.environ.get("HINT"):
    b.set_launches_in_flight(int(os.environ["HINT"]))   # the launch shape of that many launches in flight
out = []
for i in range(3):
    b.reset_steps()
    b.rollout(0x0123456789ABCDEF + i, max_plies=int(os.environ.get("CAP", "4096")), from_initial=True)
    torch.cuda.synchronize()
    w = b.steps_tensor().cpu().numpy()
    steps = int(w[::8].sum())
    out.append({"env_steps": steps, "wave_iterations": int(w[1]), "with_search": int(w[2]), "draining": int(w[3]),
                "boards_per_iteration": steps / max(int(w[1]), 1),
                "searches_per_iteration": int(w[2]) / max(int(w[1]), 1)})
print(json.dumps(out, indent=1))

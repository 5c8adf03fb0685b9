import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "board-game-simulator-python_amd")]
import numpy as np, torch
from simulator.batch import BounceBatch
SEED = 0x0123456789ABCDEF
g = np.zeros((9, 6), dtype=np.int8); g[1] = g[7] = [1, 2, 3, 3, 2, 1]
for lg in (14, 16, 18):
    b = BounceBatch(g, 1 << lg, use_torch=True)
    for cap in (16, 32, 64, 128, 256, 4096):
        b.rollout(SEED, max_plies=cap, from_initial=True); torch.cuda.synchronize()
        b.reset_steps()
        t0 = time.perf_counter()
        for i in range(3):
            b.rollout(SEED + i, max_plies=cap, from_initial=True)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
        print(f"n=2^{lg} cap={cap}: {dt*1e6:9.1f} us  steps/launch={b.steps/3:.0f}", flush=True)
    pl = b.plies
    print("plies mean", pl.mean(), "max", pl.max(), "p99", np.percentile(pl, 99), "p99.9", np.percentile(pl, 99.9))
    b.close()

"""Throughput of the generic (reference-layout) kernels on geometries the bit-packed kernels do not cover."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "board-game-simulator-python_amd")]
from tests.knobs import knobs  # BGS_EXPERIMENT ("name=value;...") as a mapping
import numpy as np
import torch
from simulator.batch import BounceBatch, ConnectBatch

SEED = 0x0123456789ABCDEF
out = {}
def rate(b, reps=3, **kw):
    b.rollout(SEED, from_initial=True, **kw); b.reset_steps(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(reps):
        b.rollout(SEED + i, from_initial=True, **kw)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    return {"generic": b.generic, "boards": b.n, "s_per_batch": dt, "env_steps_per_s": b.steps / reps / dt}
out["connect_20x20x5_n2^16"] = rate(ConnectBatch(20, 20, 5, 1 << 16, use_torch=True))
out["connect_64x64x6_n2^12"] = rate(ConnectBatch(64, 64, 6, 1 << 12, use_torch=True))
g = np.zeros((10, 8), dtype=np.int8); g[1] = g[8] = [1, 2, 3, 4, 4, 3, 2, 1]
out["bounce_10x8_n2^14_cap512"] = rate(BounceBatch(g, 1 << 14, use_torch=True), max_plies=512)
knobs["force_generic"] = "1"
out["connect_6x7x4_generic_n2^18"] = rate(ConnectBatch(6, 7, 4, 1 << 18, use_torch=True))
print(json.dumps(out, indent=1))

import os, sys, time
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "board-game-simulator-python_amd")]
import numpy as np, torch
from simulator.batch import ConnectBatch
n = 1 << 26
b = ConnectBatch(6, 7, 4, n, use_torch=True)
b.rollout(1, from_initial=True); torch.cuda.synchronize()
b.reset_steps(); t0 = time.perf_counter()
b.rollout(0x0123456789ABCDEF, from_initial=True); torch.cuda.synchronize(); dt = time.perf_counter() - t0
steps = b.steps
pl = b.plies
print("n=2^26", dt*1e3, "ms", steps/dt/1e9, "G steps/s", "steps==sum plies", steps == int(pl.sum(dtype=np.int64)), "all ended", bool(b.has_ended.all()))
small = ConnectBatch(6, 7, 4, 4096); small.set_first_game((1 << 26) - 4096); small.rollout(0x0123456789ABCDEF, from_initial=True)
print("tail shard matches", bool((small.reward == b.reward[-4096:]).all()))

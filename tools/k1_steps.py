"""K1 (`bgs_step_random`, the HBM-bound per-ply kernel) on 2^24 Connect4(6,7,4) boards: plies 0..5 never end a game,
so every board steps and the algorithmic traffic is exact.  Prints the per-ply time of one-ply launches and of a
4-ply launch.  Profiled by tools/profile_kernel.sh (FETCH_SIZE / WRITE_SIZE / stats passes)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "board-game-simulator-python_amd")]
import torch
from simulator.batch import ConnectBatch
lg = int(os.environ.get("LOG2N", "24"))
n = (1 << lg) + int(os.environ.get("N_EXTRA", "0"))  # (N_EXTRA: is a power-of-two distance between the planes a channel conflict?)
b = ConnectBatch(6, 7, 4, n, use_torch=True)
b.step_random(1); b.reset(); torch.cuda.synchronize()
out = {"boards": n, "build_id": __import__("simulator.game._abi", fromlist=["x"]).build_id()}
for plies, reps in ((1, 6), (4, 1)):
    b.reset(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        b.step_random(1, plies=plies)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) * 1e-3 / reps
    out[f"{plies}_ply_launch"] = {"s_per_launch": t, "env_steps_per_s": n * plies / t,
                                  "bytes_per_launch": n * (16 + 1 + (8 if plies == 1 else 16)),
                                  "GBps": n * (16 + 1 + (8 if plies == 1 else 16)) / t / 1e9}
print(json.dumps(out))

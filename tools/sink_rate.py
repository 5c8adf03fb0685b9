"""Rank 0's side of the multi-GPU hand-over in isolation: gathered outcome codes of `--ranks` x 2^20 games (already on
the device) -> one copy to a pinned slot -> host expansion into int8[ranks * 2^20, 2], by thread count.  Prints the
time per step and the expansion rate: at 8 ranks the host array grows by 16 MiB per step."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "board-game-simulator-python_amd")]
from tests.knobs import knobs  # BGS_EXPERIMENT ("name=value;...") as a mapping
import numpy as np
import torch
from simulator.batch import RewardSink

ap = argparse.ArgumentParser()
ap.add_argument("--ranks", type=int, default=8)
ap.add_argument("--threads", type=int, nargs="+", default=[2, 4, 8, 12, 16, 24, 32])
args = ap.parse_args()
n = args.ranks << 20
codes = torch.randint(0, 256, (n // 4,), dtype=torch.uint8, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
out = []
for t in args.threads:
    sink = RewardSink(n, slots=4, threads=t)
    hosts = [np.full((n, 2), 7, dtype=np.int8) for _ in range(4)]
    tickets = []
    for i in range(4):
        tickets.append(sink.submit_packed(codes, n, hosts[i % 4], stream=stream))
    sink.wait(tickets[-1])
    reps = 40
    t0 = time.perf_counter()
    tickets = [None] * 4
    for i in range(reps):
        k = i % 4
        if tickets[k] is not None:
            sink.wait(tickets[k])
        tickets[k] = sink.submit_packed(codes, n, hosts[k], stream=stream)
    for k in range(4):
        if tickets[k] is not None:
            sink.wait(tickets[k])
    dt = (time.perf_counter() - t0) / reps
    out.append({"threads": t, "us_per_step": dt * 1e6, "host_GBps": 2 * n / dt / 1e9})
    sink.close()
print(json.dumps({"ranks": args.ranks, "games_per_step": n, "stream_stores": "no_stream_stores" not in knobs, "by_threads": out}))

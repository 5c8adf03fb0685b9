import os, sys, time
sys.path[:0] = ["/root/repo", "/root/repo/board-game-simulator-python_amd"]
os.environ.setdefault("BGS_ROLLOUT_WPS", "2")
import numpy as np, torch
from simulator.batch import ConnectBatch, RewardSink
D, H, N, K, W = 3, 6, 1 << 20, 20, 5
streams = [torch.cuda.Stream() for _ in range(D)]
batches = []
for s in streams:
    with torch.cuda.stream(s):
        batches.append(ConnectBatch(6, 7, 4, N, use_torch=True))
hosts = [np.full((N, 2), 5, dtype=np.int8) for _ in range(H)]
sink = RewardSink(N, slots=H, threads=6)
tickets = [None] * H
def step(i, handover):
    h = i % H
    if handover:
        if tickets[h] is not None:
            sink.wait(tickets[h]); tickets[h] = None
        tickets[h] = sink.rollout(batches[i % D], hosts[h], 100 + i, from_initial=True)
    else:
        batches[i % D].rollout(100 + i, from_initial=True)
def drain():
    for h in range(H):
        if tickets[h] is not None:
            sink.wait(tickets[h]); tickets[h] = None
for rep in range(6):
    for handover in (True, False):
        for i in range(W): step(i, handover)
        drain(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(K): step(W + i, handover)
        t1 = time.perf_counter()
        drain()
        t2 = time.perf_counter()
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        print(f"handover={handover}: enqueue {1e6*(t1-t0):.0f} us, drain {1e6*(t2-t1):.0f} us, sync {1e6*(t3-t2):.0f} us, total {1e6*(t3-t0):.0f} us = {1e6*(t3-t0)/K:.1f} us/step")

"""Soak: many create / use / destroy cycles of batches, sinks, events and pinned arrays (handle and memory leaks show
up as growing device memory or failures), then one long bench-like loop with every step checked against the first
occurrence of its seed."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "board-game-simulator-python_amd")]
import numpy as np
import torch
from simulator.batch import BounceBatch, ConnectBatch, GridSink, HostEvent, PinnedArray, RewardSink
from simulator.pipeline import RolloutExecutor

SEED = 0x0123456789ABCDEF
g = np.zeros((9, 6), dtype=np.int8); g[1] = g[7] = [1, 2, 3, 3, 2, 1]
t0 = time.perf_counter()
free0, marks = None, []
for it in range(160):
    if it in (10, 60, 110):  # the first iterations load code objects and grow the runtime's own pools
        torch.cuda.synchronize(); torch.cuda.empty_cache()
        marks.append(torch.cuda.mem_get_info()[0])
        free0 = free0 or marks[-1]
    n = 1000 + 37 * it
    c = ConnectBatch(6, 7, 4, n, use_torch=bool(it & 1))
    b = BounceBatch(g, 200 + it, use_torch=bool(it & 2))
    sink = RewardSink(n, slots=1 + it % 3, threads=1 + it % 4)
    host = np.empty((n, 2), dtype=np.int8)
    ticket = sink.rollout(c, host, SEED + it, from_initial=True)
    b.rollout(SEED + it, max_plies=64, from_initial=True)
    ev, pin = HostEvent(0), PinnedArray((n, 2), np.int8)
    sink.wait(ticket)
    c.read_reward_async(pin, ev); ev.synchronize()
    assert np.array_equal(pin.array, host) and np.array_equal(c.reward, host)
    # round 3: the native loop, a grid sink, a library stream
    gsink = GridSink(c, slots=2, threads=2)
    grids = [np.empty((n, 6, 7), dtype=np.int8) for _ in range(2)]
    exe = RolloutExecutor([c], sink=gsink, host_arrays=grids, seed0=SEED + it)
    exe.enqueue(3)
    exe.drain()
    assert np.array_equal(exe.last_host_array(), c.grid)
    for obj in (exe, gsink, sink, ev, pin, c, b):
        obj.close()
torch.cuda.synchronize()
torch.cuda.empty_cache()  # use_torch batches live in torch's caching allocator
free1 = torch.cuda.mem_get_info()[0]
marks.append(free1)
print(f"160 create/destroy cycles in {time.perf_counter() - t0:.1f} s; device memory free after 10/60/110/160: {[m >> 20 for m in marks]} MiB")
assert free0 - free1 < (64 << 20), f"device memory leaked: {(free0 - free1) >> 20} MiB"

# long loop: 20000 steps over 3 batches and a sink; seeds repeat with period 50, results must repeat bit for bit
N, D, H = 1 << 18, 3, 6
batches = [ConnectBatch(6, 7, 4, N) for _ in range(D)]
sink = RewardSink(N, slots=H, threads=4)
hosts = [np.empty((N, 2), dtype=np.int8) for _ in range(H)]
tickets, seeds, first = [None] * H, [None] * H, {}
t0 = time.perf_counter()
for i in range(20000):
    h = i % H
    if tickets[h] is not None:
        sink.wait(tickets[h])
        key = seeds[h]
        digest = int(hosts[h].view(np.uint16).astype(np.uint64).sum())
        assert first.setdefault(key, digest) == digest, f"step {i}: result of seed {key} changed"
    seeds[h] = i % 50
    tickets[h] = sink.rollout(batches[i % D], hosts[h], SEED + seeds[h], from_initial=True)
for h in range(H):
    if tickets[h] is not None:
        sink.wait(tickets[h])
dt = time.perf_counter() - t0
steps = sum(b.steps for b in batches)
print(f"20000 steps of 2^18 games in {dt:.2f} s = {steps / dt / 1e9:.1f} G env-steps/s, {len(first)} distinct seeds, all repeats identical")

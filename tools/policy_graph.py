#!/usr/bin/env python3
"""Policy-driven stepping captured in a HIP graph: one game's worth of plies -- legal mask (bgs_export_device 'l') -> torch
policy -> bgs_step_actions(device actions) -- recorded once with torch.cuda.graphs and replayed, so that the ~8 launches
per ply cost the host nothing.  Every libbgs call in the loop is a plain enqueue on the batch's stream (no allocation, no
synchronisation), which is what makes the capture possible.  Prints both rates for Connect4(6,7,4)."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "board-game-simulator-python_amd")]
import torch
from simulator.batch import ConnectBatch

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 16
stream = torch.cuda.Stream()
with torch.cuda.stream(stream):
    batch = ConnectBatch(6, 7, 4, n, use_torch=True)   # bound to `stream`
    legal = torch.empty((n, 7), dtype=torch.uint8, device="cuda")

    def plies(count):
        for _ in range(count):
            batch.legal_tensor(legal)
            scores = torch.rand((n, 7), device="cuda") * legal
            col = scores.argmax(dim=1).to(torch.int32)
            col = torch.where(legal.any(dim=1), col, torch.full_like(col, -1))
            batch.step_actions(col, want_status=False)

    def eager():
        batch.reset()
        plies(42)

    eager(); torch.cuda.synchronize()
    t0 = time.perf_counter(); eager(); torch.cuda.synchronize(); t_eager = time.perf_counter() - t0
    steps_eager = batch.steps
    assert bool(batch.has_ended.all())

    graph = torch.cuda.CUDAGraph()
    batch.reset(); torch.cuda.synchronize()
    with torch.cuda.graph(graph, stream=stream):
        plies(42)
    def replay():
        batch.reset()
        graph.replay()
    replay(); torch.cuda.synchronize()
    t0 = time.perf_counter(); replay(); torch.cuda.synchronize(); t_graph = time.perf_counter() - t0
    steps_graph = batch.steps
    assert bool(batch.has_ended.all())
print(json.dumps({"boards": n, "eager_ms": t_eager * 1e3, "graph_ms": t_graph * 1e3, "speedup": t_eager / t_graph,
                  "env_steps_per_s_eager": steps_eager / t_eager, "env_steps_per_s_graph": steps_graph / t_graph}))

#!/usr/bin/env python3
"""The policy-driven ply on Bounce (default 9x6 start): `env_step` (moves in, target masks + reward pairs + ended flags out,
finished boards restarted) with a device-side policy -- first movable column, a pseudo-random one of its targets -- and
the policy's own kernels timed alone, so that the library's share of a ply can be read off.
    python3 tools/policy_loop_bounce.py [boards ...]      -> one JSON object"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "board-game-simulator-python_amd")]
import numpy as np
import torch
from simulator.batch import BounceBatch

GRID = np.zeros((9, 6), dtype=np.int8)
GRID[1] = GRID[7] = [1, 2, 3, 3, 2, 1]
PLIES = 60


def policy(targets, salt):
    """int32[n, 4] moves from int64[n, 7] target masks (6 columns + the active row): the first movable column, and of its
    targets the lowest or the highest by a salt bit (boards that have ended get (-1, ...): they ignore their move)."""
    masks, row = targets[:, :6], targets[:, 6]
    movable = masks != 0
    sx = movable.to(torch.uint8).argmax(dim=1)
    m = masks.gather(1, sx[:, None]).squeeze(1)
    low = m & (-m)
    cell_low = torch.log2(low.clamp(min=1).double()).round().long()
    cell_high = torch.log2(m.clamp(min=1).double()).floor().long()   # (exact: cells < 54, a double holds 53 bits; the top bit decides)
    cell = torch.where(((sx + salt) & 1).bool(), cell_low, cell_high)
    ok = movable.any(dim=1)
    moves = torch.stack((sx, row, cell % 6, cell // 6), dim=1).to(torch.int32)
    return torch.where(ok[:, None], moves, torch.full_like(moves, -1))


def measure(n):
    with torch.cuda.stream(torch.cuda.Stream()):
        return measure_on_stream(n)


def measure_on_stream(n):
    batch = BounceBatch(GRID, n, use_torch=True)   # bound to the current torch stream
    targets = batch.targets_tensor()
    ended = torch.zeros(n, dtype=torch.uint8, device="cuda")
    reward = torch.zeros((n, 2), dtype=torch.int8, device="cuda")

    def plies(count, with_policy, fixed=None):
        for p in range(count):
            batch.env_step(policy(targets, p) if with_policy else fixed, targets, ended=ended, reward=reward)

    plies(10, True); torch.cuda.synchronize()
    batch.reset_steps()
    t0 = time.perf_counter(); plies(PLIES, True); torch.cuda.synchronize()
    whole = (time.perf_counter() - t0) / PLIES
    steps = batch.steps / PLIES
    # the policy alone on the observation as it stands
    t0 = time.perf_counter()
    for p in range(PLIES):
        moves = policy(targets, p)
    torch.cuda.synchronize()
    alone = (time.perf_counter() - t0) / PLIES
    # the separate calls of round 3 for the same ply: moves, then the target masks
    t0 = time.perf_counter()
    for p in range(PLIES):
        batch.step_actions(policy(targets, p), want_status=False)
        batch.targets_tensor(targets)
    torch.cuda.synchronize()
    separate = (time.perf_counter() - t0) / PLIES
    # the same loops replayed from a HIP graph: the GPU's time, free of the host's launch costs
    stream = torch.cuda.current_stream()
    def replayed(body):
        graph = torch.cuda.CUDAGraph()
        torch.cuda.synchronize()
        with torch.cuda.graph(graph, stream=stream):
            body()
        graph.replay(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            graph.replay()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / 3 / PLIES
    moves_buf = torch.empty((n, 4), dtype=torch.int32, device="cuda")
    def policy_only():
        for p in range(PLIES):
            moves_buf.copy_(policy(targets, p))
    g_whole = replayed(lambda: plies(PLIES, True))
    g_alone = replayed(policy_only)
    def two_calls():
        for p in range(PLIES):
            batch.step_actions(policy(targets, p), want_status=False)
            batch.targets_tensor(targets)
    g_two = replayed(two_calls)
    batch.close()
    return {"boards": n, "graph_us_per_ply_env_step_with_policy": g_whole * 1e6, "graph_us_per_ply_policy_alone": g_alone * 1e6,
            "graph_us_per_ply_library_share": (g_whole - g_alone) * 1e6, "graph_us_per_ply_two_calls_with_policy": g_two * 1e6, "us_per_ply_env_step_with_policy": whole * 1e6, "us_per_ply_policy_alone": alone * 1e6,
            "us_per_ply_library_share": (whole - alone) * 1e6, "us_per_ply_two_calls_with_policy": separate * 1e6,
            "env_steps_per_ply": steps, "env_steps_per_s_with_policy": steps / whole}


if __name__ == "__main__":
    sizes = [int(a) for a in sys.argv[1:]] or [1 << 16, 1 << 18]
    print(json.dumps({"policy_loop_bounce": [measure(n) for n in sizes]}, indent=1))

#!/usr/bin/env python3
"""Policy-driven stepping (N2): env-steps/s of the loop  observation -> device policy -> move  on Connect4(6,7,4).

Two forms of the library's part of a ply:
  two calls   bgs_export_device 'l' (legal mask), then bgs_step_actions            -- what round 3 had
  one call    bgs_step_actions_observe: the moves and the next legal mask in one pass over the batch (round 4)
  env step    bgs_env_step: the same pass plus reward pairs, ended flags and the restart of finished boards (a vector environment)
each eager and replayed from a HIP graph (one game's worth of plies captured once), with the same torch policy -- random
scores on the legal columns, argmax: four torch kernels over n x 7 floats, 65 us a ply at 2^20 boards, several times the
library's share -- and with NO policy (a fixed action tensor: the library's share alone; env-steps/s mean nothing there,
boards fill up and refuse moves: read the microseconds per ply).
    python3 tools/policy_loop.py [boards ...]      -> one JSON object per batch size"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "board-game-simulator-python_amd")]
import torch
from simulator.batch import ConnectBatch

PLIES = 42


def measure(n):
    stream = torch.cuda.Stream()
    out = {"boards": n, "plies_per_game": PLIES}
    with torch.cuda.stream(stream):
        batch = ConnectBatch(6, 7, 4, n, use_torch=True)   # bound to `stream`
        legal = torch.empty((n, 7), dtype=torch.uint8, device="cuda")
        fixed = torch.randint(0, 7, (n,), dtype=torch.int32, device="cuda")

        def policy():
            return (torch.rand((n, 7), device="cuda") + legal).argmax(dim=1).to(torch.int32)   # ended boards ignore their move

        def two_calls(with_policy):
            for _ in range(PLIES):
                batch.legal_tensor(legal)
                batch.step_actions(policy() if with_policy else fixed, want_status=False)

        def one_call(with_policy):
            batch.legal_tensor(legal)
            for _ in range(PLIES):
                batch.step_actions_observe(policy() if with_policy else fixed, legal)

        ended = torch.zeros(n, dtype=torch.uint8, device="cuda")
        reward = torch.zeros((n, 2), dtype=torch.int8, device="cuda")

        def env_step(with_policy):   # the vector-environment step: + reward pairs, ended flags, finished boards restarted
            batch.legal_tensor(legal)
            for _ in range(PLIES):
                batch.env_step(policy() if with_policy else fixed, legal, ended=ended, reward=reward)

        for name, loop in (("two_calls", two_calls), ("one_call", one_call), ("env_step", env_step)):
            for with_policy in (True, False):
                key = name + ("" if with_policy else "_no_policy")

                def game():
                    batch.reset()
                    loop(with_policy)

                game(); torch.cuda.synchronize()
                batch.reset_steps()
                reps = 5
                t0 = time.perf_counter()
                for _ in range(reps):
                    game()
                torch.cuda.synchronize()
                eager = (time.perf_counter() - t0) / reps
                steps = batch.steps / reps
                graph = torch.cuda.CUDAGraph()
                batch.reset(); torch.cuda.synchronize()
                with torch.cuda.graph(graph, stream=stream):
                    loop(with_policy)
                batch.reset(); graph.replay(); torch.cuda.synchronize()
                batch.reset_steps()
                t0 = time.perf_counter()
                for _ in range(reps):
                    batch.reset()
                    graph.replay()
                torch.cuda.synchronize()
                replay = (time.perf_counter() - t0) / reps
                steps_g = batch.steps / reps
                out[key] = {"eager_us_per_ply": eager / PLIES * 1e6, "graph_us_per_ply": replay / PLIES * 1e6,
                            "env_steps_per_s_eager": steps / eager, "env_steps_per_s_graph": steps_g / replay,
                            # what one ply of the LIBRARY moves per board, whatever the board's state: planes 16 B in + 8 B out
                            # (the mover's), status 1 B, action 4 B, legal 7 B out = 36 B; two calls read the planes and the
                            # status a second time: + 17 B.  (The policy's own traffic is not counted.)
                            "library_bytes_per_board_ply": {"two_calls": 53, "one_call": 36, "env_step": 39}[name],
                            "library_GBps_graph": n * {"two_calls": 53, "one_call": 36, "env_step": 39}[name] / (replay / PLIES) / 1e9}
        for tag in ("", "_no_policy"):
            out["one_call_over_two_calls_graph" + tag] = out["one_call" + tag]["env_steps_per_s_graph"] / out["two_calls" + tag]["env_steps_per_s_graph"]
            out["one_call_over_two_calls_eager" + tag] = out["one_call" + tag]["env_steps_per_s_eager"] / out["two_calls" + tag]["env_steps_per_s_eager"]
        batch.close()
    return out


if __name__ == "__main__":
    sizes = [int(a) for a in sys.argv[1:]] or [1 << 16, 1 << 20]
    print(json.dumps({"policy_loop": [measure(n) for n in sizes]}))

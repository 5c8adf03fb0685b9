"""Throughput of simulator.pipeline.RolloutPipeline (the library object) on the bench's workload, for comparison with
bench.py's own loop: Connect4(6,7,4), 2^20 boards, rewards into host arrays."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "board-game-simulator-python_amd")]
from simulator.batch import ConnectBatch
from simulator.pipeline import RolloutPipeline

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
with RolloutPipeline(ConnectBatch, (6, 7, 4), 1 << 20) as pipe:
    for _ in pipe.run(range(20)):
        pass
    before = pipe.env_steps
    t0 = time.perf_counter()
    checksum = 0
    for step, rewards in pipe.run(range(1000, 1000 + steps)):
        checksum += int(rewards[step % 1024, 0])  # (touch the result)
    dt = time.perf_counter() - t0
    print(f"{steps} steps of 2^20 games: {(pipe.env_steps - before) / dt / 1e9:.1f} G env-steps/s, {dt / steps * 1e6:.1f} us per step (checksum {checksum})")

"""RolloutPipeline.run on Connect4(6,7,4), 2^20 boards, 3 in flight, with 3 / 6 / 11 / 22 host arrays per stream: five regions of
400 steps each.  More arrays give the launches more room to run ahead of a consumer that the host deschedules, and cost the
hand-over threads their cache (round 5: 8.19-8.23 against 8.11-8.18 x 10^11 on a quiet host; no steadier on a busy one)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "board-game-simulator-python_amd")]
from simulator import pipeline
from simulator.batch import ConnectBatch
for aps in (3, 11, 3, 11, 6, 22):
    with pipeline.RolloutPipeline(ConnectBatch, (6, 7, 4), 1 << 20, arrays_per_stream=aps, depth=3) as pipe:
        for _ in pipe.run(range(40)): pass
        t_end = time.perf_counter() + 0.3
        while time.perf_counter() < t_end:
            for _ in pipe.run(range(100, 106)): pass
        rates = []
        for r in range(5):
            before = pipe.env_steps; t0 = time.perf_counter(); c = 0
            for step, rewards in pipe.run(range(1000 * (r + 1), 1000 * (r + 1) + 400)):
                c += int(rewards[step % 1024, 0])
            rates.append((pipe.env_steps - before) / (time.perf_counter() - t0))
        print("arrays_per_stream", aps, ["%.3e" % x for x in sorted(rates)], flush=True)

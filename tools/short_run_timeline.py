#!/usr/bin/env python3
"""Where the time of a short timed region goes: the headline workload (Connect 6x7x4, 2^20 boards, 3 in flight, rewards
to host) for K steps with every launch bracketed by events -- start and end of every launch relative to the first, and
the host's clock around enqueue / drain.  Prints one JSON object.   python tools/short_run_timeline.py [K] [repeats]
With sink_trace=1 the sink's threads report (stderr, microseconds of the same clock) when each delivery's codes were
seen and expanded; round 3, r3_drain.sh in the git history puts the two together for the last delivery of each repeat."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "board-game-simulator-python_amd")]
from tests.knobs import knobs  # BGS_EXPERIMENT ("name=value;...") as a mapping
import numpy as np
from simulator.game import _abi
_abi.request_hardware_queues()
import torch
if os.environ.get("BGS_SCHEDULE_SPIN"):  # (experiment: synchronising calls spin instead of sleeping on an interrupt)
    import ctypes
    print("hipSetDeviceFlags(spin) ->", ctypes.CDLL("libamdhip64.so").hipSetDeviceFlags(1), file=sys.stderr)
from simulator.batch import ConnectBatch, RewardSink
from simulator.pipeline import RolloutExecutor

K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
REPEATS = int(sys.argv[2]) if len(sys.argv) > 2 else 5
N, D, SEED = 1 << 20, 3, 0x0123456789ABCDEF
streams = [torch.cuda.Stream() for _ in range(D)]
batches = []
for s in streams:
    with torch.cuda.stream(s):
        batches.append(ConnectBatch(6, 7, 4, N, use_torch=True))
hosts = [np.zeros((N, 2), dtype=np.int8) for _ in range(12)]
sink = RewardSink(N, slots=12, threads=6)
exe = RolloutExecutor(batches, sink=sink, host_arrays=hosts, seed0=SEED)
exe.enqueue(60); exe.drain()   # warm
runs = []
for rep in range(REPEATS):
    exe.enqueue(5); exe.drain(); exe.kernel_ms()
    for b in batches: b.reset_steps()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    exe.enqueue(K, True, 1)
    t1 = time.perf_counter()
    exe.drain()
    t2 = time.perf_counter()
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    tl = exe.timeline()
    exe.kernel_ms()
    steps = sum(b.steps for b in batches)
    last_end = max(e for _, e in tl)
    if knobs.get("sink_trace"):
        print(f"host-trace rep {rep} t0 {t0 * 1e6:.1f} enqueue_returns {t1 * 1e6:.1f} drain_returns {t2 * 1e6:.1f} synchronized {t3 * 1e6:.1f}", file=sys.stderr)
    runs.append({"host_us": {"enqueue_returns": (t1 - t0) * 1e6, "drain_returns": (t2 - t0) * 1e6, "synchronized": (t3 - t0) * 1e6},
                 "device_us": {"last_kernel_ends_after_first_starts": last_end * 1e3,
                               "launches": [[round(a * 1e3, 1), round(z * 1e3, 1)] for a, z in tl]},
                 "env_steps": steps, "value": steps / (t3 - t0)})
best = max(runs, key=lambda r: r["value"])
print(json.dumps({"steps": K, "values": [r["value"] for r in runs], "best_run": best}, indent=1))

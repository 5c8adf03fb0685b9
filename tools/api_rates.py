#!/usr/bin/env python3
"""What each level of the API delivers (round-4 review, weak #5 / next #3): env-steps/s with every step's rewards in a host
array, for BASELINE configs 2, 3 and 4, through
  naive      `batch.rollout(seed, from_initial=True)` followed by `batch.reward` (one launch, one synchronous read, per step);
  pipeline   `for step, rewards in RolloutPipeline(...).run(seeds)`: the documented Python loop (README), which since round 5
             enqueues its launches in bursts through the native loop (bgs_pipeline_enqueue_seeds) and waits per step;
  executor   `RolloutExecutor.enqueue(K); drain()`: the whole region as one library call (what bench.py times).
One child process per (config, level): the hardware queues a deep Bounce pipeline needs are asked for before HIP starts.
Every level is timed over three regions and the median is reported (`values_of_3` holds all three), as bench.py does.

    python3 tools/api_rates.py > profiles/r05_api_rates.json"""
import json, os, subprocess, sys, time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "board-game-simulator-python_amd")]
SEED = 0x0123456789ABCDEF
CONFIGS = {"connect_6x7x4": ("connect", (6, 7, 4), 1 << 20, 2**31 - 1, 400), "connect_12x13x5": ("connect", (12, 13, 5), 1 << 18, 2**31 - 1, 320),
           "bounce_default": ("bounce", None, 1 << 18, 4096, 200)}


def child(name, level):
    import numpy as np

    kind, args, n, max_plies, steps = CONFIGS[name]
    from simulator import pipeline
    from simulator.batch import BounceBatch, ConnectBatch, RewardSink

    cls = ConnectBatch if kind == "connect" else BounceBatch
    if kind == "bounce":
        grid = np.zeros((9, 6), dtype=np.int8)
        grid[1] = grid[7] = [1, 2, 3, 3, 2, 1]
        args = (grid,)
    depth = pipeline.default_depth(cls, args)
    if level == "naive":
        b = cls(*args, n)
        steps = max(20, steps // 8)
        for s in range(3):
            b.rollout(SEED + s, max_plies=max_plies, from_initial=True)
            b.reward
        regions = []
        for rep_ in range(3):   # three timed regions, the median reported (bench.py's way: a busy host stalls a region now and then)
            b.reset_steps()
            t0 = time.perf_counter()
            for s in range(steps):
                b.rollout(SEED + 10 + s, max_plies=max_plies, from_initial=True)
                r = b.reward
            regions.append((b.steps, time.perf_counter() - t0))
    elif level == "pipeline":
        with pipeline.RolloutPipeline(cls, args, n, max_plies=max_plies) as pipe:
            for _ in pipe.run(range(4 * depth)):
                pass
            t_end = time.perf_counter() + 0.3   # (the device's power state climbs under load, as in bench.py)
            while time.perf_counter() < t_end:
                for _ in pipe.run(range(100, 100 + 2 * depth)):
                    pass
            regions = []
            for rep_ in range(3):
                before = pipe.env_steps
                t0 = time.perf_counter()
                check = 0
                for step, rewards in pipe.run(range(1000 * (rep_ + 1), 1000 * (rep_ + 1) + steps)):
                    check += int(rewards[step % 1024, 0])   # (touch the result)
                regions.append((pipe.env_steps - before, time.perf_counter() - t0))
            depth = pipe.depth
    else:
        import torch

        depth = pipeline.usable_depth(depth, False, "api_rates")
        streams = [torch.cuda.Stream() for _ in range(depth)]
        batches = []
        for s in streams:
            with torch.cuda.stream(s):
                batches.append(cls(*args, n, use_torch=True))
        per = 6 if kind == "bounce" else 3
        hosts = [np.zeros((n, 2), dtype=np.int8) for _ in range(per * depth)]
        sink = RewardSink(n, slots=per * depth, threads=6)
        exe = pipeline.RolloutExecutor(batches, sink=sink, host_arrays=hosts, seed0=SEED, max_plies=max_plies)
        t_end = time.perf_counter() + 0.3
        while time.perf_counter() < t_end:
            exe.enqueue(2 * depth)
            exe.drain()
        regions = []
        for rep_ in range(3):
            for b in batches:
                b.reset_steps()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            exe.enqueue(steps)
            exe.drain()
            torch.cuda.synchronize()
            regions.append((sum(b.steps for b in batches), time.perf_counter() - t0))
        exe.close()
        sink.close()
    rates = sorted(done / dt for done, dt in regions)
    done, dt = sorted(regions, key=lambda r: r[0] / r[1])[1]
    print(json.dumps({"value": done / dt, "values_of_3": rates, "unit": "env-steps/s", "steps": steps, "us_per_step": dt / steps * 1e6,
                      "in_flight": 1 if level == "naive" else depth}))


def main():
    if len(sys.argv) == 3:
        return child(sys.argv[1], sys.argv[2])
    from simulator.game import _abi

    out = {"what": "env-steps/s with every step's rewards in a host array, per API level (tools/api_rates.py)", "build_id": _abi.build_id(),
           "unit_ids": _abi.unit_ids(), "configs": {}}
    for name in CONFIGS:
        out["configs"][name] = {}
        for level in ("naive", "pipeline", "executor"):
            env = dict(os.environ)
            if name == "bounce_default" and level != "naive":
                env.setdefault("GPU_MAX_HW_QUEUES", "24")
            p = subprocess.run([sys.executable, os.path.abspath(__file__), name, level], env=env, capture_output=True, text=True, timeout=600)
            lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
            out["configs"][name][level] = json.loads(lines[-1]) if p.returncode == 0 and lines else {"error": p.stderr.strip()[-300:]}
        c = out["configs"][name]
        if all("value" in c[k] for k in ("pipeline", "executor")):
            c["pipeline_over_executor"] = c["pipeline"]["value"] / c["executor"]["value"]
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()

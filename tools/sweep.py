#!/usr/bin/env python3
"""Parametrised sweeps on the GPU box (round 4: one script instead of a directory of one-off shell loops).

    python3 tools/sweep.py bounce-block   [--reps N]     K3p workgroup size (bounce_block) x parking threshold, 20 in flight
    python3 tools/sweep.py bounce-park    [--reps N]     K3p parking threshold (bounce_pieces_park) with / without the device-wide pool
    python3 tools/sweep.py bounce-tail    [--reps N]     ply cap of the bulk pass (bounce_plan) x parking threshold, 20 in flight
    python3 tools/sweep.py bounce-depth   [--reps N]     Bounce batches in flight (8 .. 28) with the default launch shape
    python3 tools/sweep.py k2c-shape      [--reps N]     Connect(12,13,5): waves per SIMD per launch (BGS_ROLLOUT_WPS) x launches in flight
    python3 tools/sweep.py k2c-depth      [--reps N]     Connect(12,13,5) batches in flight (1 .. 12)
    python3 tools/sweep.py headline-depth [--reps N]     Connect4(6,7,4) batches in flight (1 .. 8)
    python3 tools/sweep.py bench --env NAME=v1,v2,... [--args "bench.py arguments"] [--repeat R]
                                                         bench.py with one environment variable (or, NAME starting with "--",
                                                         one command-line option) swept over values: value / median of 3 /
                                                         device-resident per point.  Covers what round 3 kept as one-off
                                                         scripts: BGS_BENCH_SLOT_FACTOR (slots), --host-threads, --prewarm-ms,
                                                         --inflight, BGS_ROLLOUT_WPS, BGS_BENCH_PAIRS, sink_spin_us, ... (a lower-case NAME is a switch of BGS_EXPERIMENT)
Every point is a child process (the library reads its knobs when a batch is created; GPU_MAX_HW_QUEUES when HIP starts);
prints one line per point and a JSON summary."""
import argparse, json, os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RATE = os.path.join(ROOT, "tools", "rollout_rate.py")


def with_knobs(env):
    """A child's environment: lower-case names are the library's A/B switches (BGS_EXPERIMENT), the others variables."""
    e = dict(os.environ)
    knobs = [f"{k}={v}" for k, v in (env or {}).items() if k.islower()]
    if knobs:
        e["BGS_EXPERIMENT"] = ";".join(([e["BGS_EXPERIMENT"]] if e.get("BGS_EXPERIMENT") else []) + knobs)
    e.update({k: str(v) for k, v in (env or {}).items() if not k.islower()})
    return e


def point(config, depth, reps, env=None, extra=()):
    e = with_knobs(env)
    out = subprocess.run([sys.executable, RATE, config, "--depth", str(depth), "--reps", str(reps), *extra], env=e, capture_output=True, text=True)
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    if out.returncode != 0 or not lines:
        return {"error": out.stderr.strip()[-300:]}
    d = json.loads(lines[-1])
    return {"solo": d["one_launch_at_a_time"]["env_steps_per_s"], "pipelined": d[f"{depth}_in_flight"]["env_steps_per_s"]}


def bench_point(env, args):
    e = with_knobs(env)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-other-configs", *args], env=e,
                         capture_output=True, text=True)
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    if out.returncode != 0 or not lines:
        return {"error": out.stderr.strip()[-300:]}
    d = json.loads(lines[-1])
    return {"value": d["value"], "median_of_3": d.get("value_median_of_3"), "device_resident": (d.get("device_resident") or {}).get("value"),
            "ms_per_step": d["ms_per_step"]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--env", default="", help="bench sweep: NAME=v1,v2,...")
    ap.add_argument("--args", default="", help="bench sweep: further bench.py arguments")
    ap.add_argument("--repeat", type=int, default=1)
    ap.add_argument("what", choices=("bench", "bounce-block", "bounce-park", "bounce-tail", "bounce-depth", "k2c-shape", "k2c-depth", "headline-depth"))
    ap.add_argument("--reps", type=int, default=0)
    args = ap.parse_args()
    rows = []
    if args.what == "bench":
        name, _, values = args.env.partition("=")
        if not name or not values:
            ap.error("bench needs --env NAME=v1,v2,...")
        for v in values.split(","):
            for _ in range(args.repeat):
                extra = args.args.split()
                r = bench_point({}, extra + [name, v]) if name.startswith("--") else bench_point({name: v}, extra)
                rows.append(dict({name: v}, **r))
                print(rows[-1], flush=True)
    elif args.what == "bounce-block":
        for block in (256, 512, 1024):
            for park in (16, 32):
                r = point("bounce", 20, args.reps or 120, {"bounce_block": block, "bounce_park": park})
                rows.append(dict({"block": block, "park": park}, **r))
                print(rows[-1], flush=True)
    elif args.what == "bounce-park":
        for pool in (1, 0):
            for park in ((32, 40, 48, 56, 63) if pool else (32, 48, 63)):
                r = point("bounce", 20, args.reps or 120, {"bounce_pool": pool, "bounce_pieces_park": park})
                rows.append(dict({"pool": pool, "park": park}, **r))
                print(rows[-1], flush=True)
    elif args.what == "bounce-tail":
        # the bulk pass's ply cap in front of the one-board-per-wave tail pass (lanes 64; round 3's tail: lanes 8)
        for plan in ("auto", "64:1,0:64", "96:1,0:64", "128:1,0:64", "160:1,0:64", "224:1,0:64", "160:1,0:8"):
            for park in (40,):
                env = {"bounce_pieces_park": park}
                if plan != "auto":
                    env["bounce_plan"] = plan
                rows.append(dict({"plan": plan, "park": park}, **point("bounce", 20, args.reps or 120, env)))
                print(rows[-1], flush=True)
    elif args.what == "bounce-depth":
        for depth in (8, 12, 16, 20, 24, 28):
            rows.append(dict({"depth": depth}, **point("bounce", depth, args.reps or 6 * depth)))
            print(rows[-1], flush=True)
    elif args.what == "k2c-shape":
        for depth in (1, 2, 4):
            for wps in (1, 2, 3, 4, 6, 8):
                rows.append(dict({"depth": depth, "wps": wps}, **point("connect12x13", depth, args.reps or 60 * depth, {"BGS_ROLLOUT_WPS": wps})))
                print(rows[-1], flush=True)
    elif args.what == "k2c-depth":
        for depth in (1, 2, 3, 4, 6, 8, 12):
            rows.append(dict({"depth": depth}, **point("connect12x13", depth, args.reps or 40 * depth)))
            print(rows[-1], flush=True)
    else:
        for depth in (1, 2, 3, 4, 6, 8):
            rows.append(dict({"depth": depth}, **point("connect6x7", depth, args.reps or 60 * depth, {"BGS_ROLLOUT_WPS": 2})))
            print(rows[-1], flush=True)
    print(json.dumps({"sweep": args.what, "rows": rows}))


if __name__ == "__main__":
    main()

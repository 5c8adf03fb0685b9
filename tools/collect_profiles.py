#!/usr/bin/env python3
"""gpurun_out/prof_* (tools/profile_all.sh) -> profiles/<round>_*.json (round: BGS_PROFILE_ROUND, default r06), including
profiles/<round>_rollout_counters.json, the per-launch PMC figures of the bench kernel that bench.py quotes when the id of
the kernel UNIT they were taken on (connect / bounce / generic: bgs_kernel_unit_id) matches the running library's.  Only the
tags whose passes exist under gpurun_out/ are (re)written: profile_all.sh skips the units that did not move."""
import json, os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "board-game-simulator-python_amd")]
OUT = os.path.join(ROOT, "profiles")
ROUND = os.environ.get("BGS_PROFILE_ROUND", "r06")
TAG_UNIT = {"bench": "connect", "k1": "connect", "k2c": "connect", "k2b": "connect", "bounce": "bounce", "bounce_solo": "bounce",
            "bounce_k3f": "bounce", "bounce8": "bounce"}


def summary(tag, want=""):
    return json.loads(subprocess.check_output([sys.executable, os.path.join(ROOT, "tools", "summarize_profile.py"), tag, want]))


def hbm_bytes(k):
    # rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB; gfx950 reports HALF the bytes of wide coalesced reads
    # (MI355X_MICROARCH.md, HBM): FETCH_SIZE is doubled, WRITE_SIZE taken as is
    return (2 * k.get("FETCH_SIZE", 0.0) + k.get("WRITE_SIZE", 0.0)) * 1024.0


def main():
    from simulator.game import _abi

    build = _abi.build_id()
    units = _abi.unit_ids()
    method = ("tools/profile_all.sh: rocprofv3 --kernel-trace --stats, then --pmc in separate passes (SQ_*; FETCH_SIZE; "
              "WRITE_SIZE), per-dispatch means; FETCH_SIZE doubled (gfx950 counts half of wide coalesced reads), KiB -> B")
    for tag, want, name in (("bench", "k_connect_rollout_opened", f"{ROUND}_bench_kernel.json"), ("k1", "step_random", f"{ROUND}_k1.json"),
                            ("k2c", "_lds", f"{ROUND}_k2c.json"), ("k2b", "k_connect_rollout_aligned_wide", f"{ROUND}_k2b.json"),
                            ("bounce", "k_bounce", f"{ROUND}_bounce.json"), ("bounce_solo", "k_bounce", f"{ROUND}_bounce_solo.json"), ("bounce_k3f", "k_bounce", f"{ROUND}_bounce_k3f.json"),
                            ("bounce8", "k_bounce_rollout", f"{ROUND}_bounce_lane_groups.json")):
        try:
            s = summary(tag, want)
        except Exception as exc:  # a tag that was not profiled in this pass
            print(f"skipping {tag}: {exc}", file=sys.stderr)
            continue
        if not s:
            continue
        for k in s.values():
            k["hbm_bytes_per_launch"] = hbm_bytes(k)
            if "SQ_THREAD_CYCLES_VALU" in k and k.get("SQ_ACTIVE_INST_VALU"):
                k["active_lanes_per_valu_instruction"] = k["SQ_THREAD_CYCLES_VALU"] / k["SQ_ACTIVE_INST_VALU"]
        # one rollout = every kernel of the tag, once per step (Bounce: bulk pass, compaction, tail pass): the per-step
        # totals are what bench.py quotes for the configuration
        per_step = max(k.get("dispatches", 1) for k in s.values())
        total = {"valu_wave_instructions_per_launch": sum(k.get("SQ_INSTS_VALU", 0.0) * k.get("dispatches", per_step) / per_step for k in s.values()),
                 "hbm_bytes_per_launch": sum(k["hbm_bytes_per_launch"] * k.get("dispatches", per_step) / per_step for k in s.values()),
                 "kernel_us_per_launch": sum(k.get("mean_us", 0.0) * k.get("dispatches", per_step) / per_step for k in s.values())}
        lanes = sum(k.get("SQ_THREAD_CYCLES_VALU", 0.0) * k.get("dispatches", 1) for k in s.values())
        insts = sum(k.get("SQ_ACTIVE_INST_VALU", 0.0) * k.get("dispatches", 1) for k in s.values())
        if insts:
            total["active_lanes_per_valu_instruction"] = lanes / insts
        extra = {}
        if tag == "bench":
            # the same kernel under `rocprofv3 --kernel-trace --stats -- python3 bench.py` (the bench's own command: pre-warm,
            # warm-up and 200 timed steps, three launches in flight throughout)
            try:
                full = summary("benchfull", want)
                if full:
                    k = next(iter(full.values()))
                    extra["same_command_as_bench"] = {
                        "command": "rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline --no-other-configs --no-device-resident --no-repeats",
                        "dispatches": k.get("dispatches"), "mean_us": k.get("mean_us"), "min_us": k.get("min_us"), "max_us": k.get("max_us")}
                    # what the bench itself read INSIDE that traced run: the tracer's per-dispatch work thins the overlap
                    # of the three launches in flight, so the run is slower and its kernels, sharing the chip less, shorter
                    log = os.path.join(ROOT, "gpurun_out", "prof_benchfull_stats.log")
                    if os.path.exists(log):
                        lines = [ln for ln in open(log) if ln.startswith("{")]
                        if lines:
                            line = json.loads(lines[-1])
                            extra["same_command_as_bench"]["bench_line_inside_the_traced_run"] = {
                                "value": line["value"], "ms_per_step": line["ms_per_step"],
                                "kernel_ms_per_launch_live": line["roofline"].get("kernel_ms_per_launch"),
                                "event_pairs": line["roofline"].get("event_pairs")}
            except Exception as exc:
                print(f"no benchfull pass: {exc}", file=sys.stderr)
        with open(os.path.join(OUT, name), "w") as fh:
            json.dump(dict({"build_id": build, "unit": TAG_UNIT[tag], "unit_id": units[TAG_UNIT[tag]], "method": method}, **total, **extra, kernels=s), fh, indent=1)
    try:
        misc = summary("misc")
    except Exception as exc:
        misc = None
        print(f"skipping misc: {exc}", file=sys.stderr)
    if misc:
        with open(os.path.join(OUT, f"{ROUND}_misc_kernel_stats.json"), "w") as fh:
            json.dump({"build_id": build, "unit_ids": units, "command": "rocprofv3 --kernel-trace --stats -- python3 tools/measure_all.py", "kernels": misc}, fh, indent=1)
    mfile = os.path.join(ROOT, "gpurun_out", "measure_all.json")
    if os.path.exists(mfile) and os.path.getsize(mfile):
        with open(mfile) as fh, open(os.path.join(OUT, f"{ROUND}_secondary_measurements.json"), "w") as out:
            out.write(fh.read())
    try:
        bench = summary("bench", "k_connect_rollout_opened")
    except Exception as exc:
        print(f"no bench pass in gpurun_out/ ({exc}): {ROUND}_rollout_counters.json left as it is", file=sys.stderr)
        return
    if not bench:
        return
    k = next(iter(bench.values()))
    with open(os.path.join(ROOT, "gpurun_out", "valu_mix.json")) as fh:
        mix = json.load(fh)
    # dynamic mix: the opening block runs once per 64 games, the loop body makes up the rest of the VALU instructions
    openings = (1 << 20) / 64.0
    n_open = openings * mix["opening_block"]["valu_instructions"]
    n_loop = max(k["SQ_INSTS_VALU"] - n_open, 0.0)
    mix_cpi = (n_open * mix["opening_block"]["mix_cycles_per_instruction"] +
               n_loop * mix["loop_body"]["mix_cycles_per_instruction"]) / (n_open + n_loop)
    mix["weights"] = {"opening_block_instructions_per_launch": n_open, "loop_and_other_instructions_per_launch": n_loop}
    counters = {
        "build_id": build,
        "unit": "connect",
        "unit_id": units["connect"],
        "kernel": next(iter(bench)),
        "dispatches": k.get("dispatches"),
        "mean_us": k.get("mean_us"),
        "valu_wave_instructions_per_launch": k["SQ_INSTS_VALU"],
        "salu_wave_instructions_per_launch": k["SQ_INSTS_SALU"],
        "active_lanes_per_valu_instruction": k["SQ_THREAD_CYCLES_VALU"] / k["SQ_ACTIVE_INST_VALU"],
        "FETCH_SIZE_KiB": k["FETCH_SIZE"],
        "WRITE_SIZE_KiB": k["WRITE_SIZE"],
        "hbm_bytes_per_launch": hbm_bytes(k),
        "mix_cycles_per_instruction": mix_cpi,
        "mix": mix,
        "method": method + "; command: python3 bench.py --steps 10 --warmup 2 --prewarm-ms 0 --no-cpu-baseline --no-device-resident --no-other-configs --no-repeats; "
        "mix_cycles_per_instruction from tools/valu_mix.py",
    }
    with open(os.path.join(OUT, f"{ROUND}_rollout_counters.json"), "w") as fh:
        json.dump(counters, fh, indent=1)
    print(json.dumps({kk: counters[kk] for kk in ("build_id", "unit_id", "valu_wave_instructions_per_launch", "hbm_bytes_per_launch",
                                                  "active_lanes_per_valu_instruction", "mean_us")}))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""gpurun_out/busy_* (tools/busy_counters.sh) -> profiles/<round>_valu_busy.json (BGS_PROFILE_ROUND, default r06): per case and kernel the mean counter values
per dispatch and what they say about the vector issue pipe.

Units (rocprofv3 -L on gfx950): SQ_ACTIVE_INST_VALU, SQ_ACTIVE_INST_ANY, SQ_WAVE_CYCLES, SQ_WAIT_*, SQ_BUSY_CU_CYCLES are
quad-cycles summed over the chip; SQ_ACTIVE_INST_VALU2 = quad-cycles in which a SIMD issued TWO VALU instructions;
GRBM_GUI_ACTIVE = cycles the chip was busy, summed over the 8 XCDs.  rocprofv3's own derived metric is
VALUBusy = SQ_ACTIVE_INST_VALU / CU_NUM / GRBM_GUI_ACTIVE(per XCD): 1.0 = every SIMD of every CU has one VALU instruction
in its pipe in every quad-cycle."""
import csv, glob, json, os, sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "board-game-simulator-python_amd")]
CUS, SIMDS, XCDS = 256, 1024, 8


def case(tag, want):
    acc = defaultdict(lambda: defaultdict(list))
    dur = defaultdict(list)
    for f in glob.glob(os.path.join(ROOT, "gpurun_out", f"busy_{tag}", "**", "*counter_collection.csv"), recursive=True):
        seen = set()
        for row in csv.DictReader(open(f)):
            acc[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
            key = (row["Kernel_Name"], row.get("Dispatch_Id"))
            if key not in seen and row.get("Start_Timestamp") and row.get("End_Timestamp"):
                seen.add(key)
                dur[row["Kernel_Name"]].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
    out = {}
    for k, cs in acc.items():
        if want not in k:
            continue
        m = {c: sum(v) / len(v) for c, v in cs.items()}
        m["dispatches"] = len(next(iter(cs.values())))
        if dur[k]:
            m["mean_us"] = sum(dur[k]) / len(dur[k]) / 1e3
        cycles = m.get("GRBM_GUI_ACTIVE", 0.0) / XCDS           # cycles of the dispatch
        if cycles and m.get("SQ_ACTIVE_INST_VALU"):
            quads = cycles / 4.0
            m["derived"] = {
                "cycles_per_dispatch": cycles,
                "clock_GHz_if_mean_us_is_the_wall_time": cycles / (m["mean_us"] * 1e3) if m.get("mean_us") else None,
                "valu_busy_frac": m["SQ_ACTIVE_INST_VALU"] / SIMDS / quads,
                "valu_instr_per_simd_per_quad_cycle": m["SQ_INSTS_VALU"] / SIMDS / quads,
                "frac_of_quads_with_two_valu_issued": m.get("SQ_ACTIVE_INST_VALU2", 0.0) / SIMDS / quads,
                "active_inst_valu_over_insts_valu": m["SQ_ACTIVE_INST_VALU"] / m["SQ_INSTS_VALU"],
                "cu_busy_frac": m.get("SQ_BUSY_CU_CYCLES", 0.0) / CUS / quads if m.get("SQ_BUSY_CU_CYCLES") else None,
                "waves_resident_per_simd": m.get("SQ_WAVE_CYCLES", 0.0) / SIMDS / quads,
                "wave_time_split": {
                    "issuing_or_executing (SQ_ACTIVE_INST_ANY)": m.get("SQ_ACTIVE_INST_ANY", 0.0) / m["SQ_WAVE_CYCLES"],
                    "waiting_for_an_issue_slot (SQ_WAIT_INST_ANY)": m.get("SQ_WAIT_INST_ANY", 0.0) / m["SQ_WAVE_CYCLES"],
                    "parked (SQ_WAIT_ANY: s_waitcnt, barriers)": m.get("SQ_WAIT_ANY", 0.0) / m["SQ_WAVE_CYCLES"],
                } if m.get("SQ_WAVE_CYCLES") else None,
            }
        out[k.split("(")[0][:120]] = m
    return out


def main():
    from simulator.game import _abi

    cases = {
        "k2o_solo": ("k_connect_rollout_opened", "one launch of 2^20 Connect4 games, 2 waves per SIMD (the bench's launch, alone)"),
        "k2o_3deep": ("k_connect_rollout_opened", "one launch of 3 x 2^20 games, 6 waves per SIMD, 512 games per wave: the chip as three bench launches in flight fill it"),
        "k2c_solo": ("_lds", "Connect(12,13,5), one launch of 2^18 boards"),
        "k2c_8deep": ("_lds", "Connect(12,13,5), 8 x 2^18 boards in one launch, 8 waves per SIMD, 256 games per wave"),
        "k3p_solo": ("k_bounce_rollout_pieces", "Bounce default, one launch of 2^18 boards in the shape of 20 in flight"),
        "k3p_8x": ("k_bounce_rollout_pieces", "Bounce default, 8 x 2^18 boards in one launch"),
    }
    units = _abi.unit_ids()
    case_unit = {"k2o_solo": "connect", "k2o_3deep": "connect", "k2c_solo": "connect", "k2c_8deep": "connect", "k3p_solo": "bounce", "k3p_8x": "bounce"}
    rnd = os.environ.get("BGS_PROFILE_ROUND", "r06")
    path = os.path.join(ROOT, "profiles", f"{rnd}_valu_busy.json")
    # cases that were not re-taken in this pass (their unit did not move: busy_counters.sh skipped them) are carried over
    # from the newest file that holds them for the running unit id
    carried = {}
    for old in sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_valu_busy.json")), reverse=True):
        with open(old) as fh:
            prev = json.load(fh)
        for tag, entry in prev.get("cases", {}).items():
            if tag not in carried and entry.get("unit_id") and entry["unit_id"] == units.get(entry.get("unit")):
                carried[tag] = entry
    out = {"build_id": _abi.build_id(),
           "method": "tools/busy_counters.sh: rocprofv3 --kernel-trace --pmc (one pass, 8 SQ + 2 GRBM counters), per-dispatch means; "
                     "dispatches are serialised by the counter collection, so pipelined cases are counted as one launch of N x the boards "
                     "with N x the waves per SIMD",
           "cases": {}}
    for tag, (want, what) in cases.items():
        c = case(tag, want)
        if c:
            out["cases"][tag] = {"what": what, "unit": case_unit[tag], "unit_id": units[case_unit[tag]], "kernels": c}
        elif tag in carried:
            out["cases"][tag] = carried[tag]
    with open(path, "w") as fh:
        json.dump(out, fh, indent=1)
    for tag, c in out["cases"].items():
        for k, m in c["kernels"].items():
            d = m.get("derived") or {}
            print(tag, k[:40], "us %.1f" % m.get("mean_us", 0), {kk: (round(v, 3) if isinstance(v, float) else v) for kk, v in d.items() if kk != "wave_time_split"},
                  {kk[:12]: round(v, 3) for kk, v in (d.get("wave_time_split") or {}).items()})


if __name__ == "__main__":
    main()

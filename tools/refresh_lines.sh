#!/bin/bash
# The committed lines of a round, one call on the GPU box (R = the round's tag, default r05): the driver-style bench line at
# 200 and 20 steps, what each API level delivers (three runs), the object API's latency, and -- with DIST=1 -- the N > 1
# rehearsals of tools/r5_dist.sh.  Copy gpurun_out/${R}_*.json into profiles/ afterwards (the counters: tools/profile_all.sh,
# then tools/collect_profiles.py and tools/busy_counters.py on the merged gpurun_out/).
R=${R:-r05}
mkdir -p gpurun_out
timeout -k 10 400 python bench.py 2> gpurun_out/${R}_bench.err | grep "^{" > gpurun_out/${R}_bench.json
timeout -k 10 400 python bench.py --steps 20 --warmup 5 2> gpurun_out/${R}_bench_steps20.err | grep "^{" > gpurun_out/${R}_bench_steps20.json
for i in 1 2 3; do timeout -k 10 300 python tools/api_rates.py > gpurun_out/${R}_api_rates_$i.json 2> gpurun_out/${R}_api_rates_$i.err; done
timeout -k 10 200 python tools/object_latency.py > gpurun_out/${R}_object_latency.json 2> gpurun_out/${R}_object_latency.err
[ "${DIST:-0}" = "1" ] && timeout -k 10 900 bash tools/r5_dist.sh > gpurun_out/${R}_dist.log 2>&1
python3 - "$R" <<'PY'
import json, sys
R = sys.argv[1]
for f in (f"{R}_bench", f"{R}_bench_steps20"):
    d = json.load(open("gpurun_out/" + f + ".json"))
    r = d["roofline"]
    print(f, "value %.3e" % d["value"], ["%.3e" % v for v in d["values_of_3"]], "solo %.3e" % d["solo"]["value"], "frac", round(r["frac"] or 0, 3), "busy", r.get("valu_busy_frac"))
    for k, v in d["other_configs"].items():
        print("   ", k, "%.3e" % v["value"], "dev %.3e" % v["device_resident"], v["parity_with_oracle"], "frac", round(v["valu_issue"]["frac"] or 0, 3))
for i in (1, 2, 3):
    d = json.load(open(f"gpurun_out/{R}_api_rates_{i}.json"))
    print("api", i, {k: round(v.get("pipeline_over_executor", 0), 3) for k, v in d["configs"].items()})
PY

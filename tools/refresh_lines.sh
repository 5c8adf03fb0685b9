#!/bin/bash
# The committed lines of a round, one call on the GPU box: the driver-style bench line (200 and 20 steps) and the
# distributed lines of tools/r4_dist.sh; copy gpurun_out/r4_bench*.json / r4_dist_*.json into profiles/ afterwards.
timeout -k 10 400 python bench.py 2> gpurun_out/r4_bench.err | grep "^{" > gpurun_out/r4_bench.json; timeout -k 10 400 python bench.py --steps 20 --warmup 5 2> gpurun_out/r4_bench20.err | grep "^{" > gpurun_out/r4_bench20.json; timeout -k 10 500 tools/r4_dist.sh > gpurun_out/r4_dist.log 2>&1; tail -4 gpurun_out/r4_dist.log | cut -c1-330
python3 - <<'PY'
import json
for f in ("r4_bench", "r4_bench20"):
    d = json.load(open("gpurun_out/" + f + ".json"))
    r = d["roofline"]
    print(f, "value %.3e" % d["value"], ["%.3e" % v for v in d["values_of_3"]], "solo %.3e" % d["solo"]["value"], "frac", round(r["frac"], 3), "busy", r.get("valu_busy_frac"))
    for k, v in d["other_configs"].items():
        print("   ", k, "%.3e" % v["value"], "dev %.3e" % v["device_resident"], v["parity_with_oracle"], "frac", round(v["valu_issue"]["frac"], 3))
PY

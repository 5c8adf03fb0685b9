#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_pipeline.py -x -q -m gpu > gpurun_out/r3g_tests.log 2>&1
echo "tests rc=$?"; tail -15 gpurun_out/r3g_tests.log
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > gpurun_out/r3g_bench20.json 2> gpurun_out/r3g_bench20.err
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r3g_bench20.json") if l.startswith("{")][-1])
print("%.3e"%d["value"], d["values_of_3"], json.dumps(d.get("grids_to_host"),indent=0))
for k,v in d["other_configs"].items(): print(k, "%.3e"%v.get("value",0), v.get("solo"), v.get("parity_with_oracle"), v.get("error"))
PY

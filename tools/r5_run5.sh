#!/bin/bash
# round 5, GPU run 5: the full GPU suite on the current build, K2o launch-shape sweeps, the API-level rates
set -u
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r5_tests3.log 2>&1; echo "gpu suite rc=$?"; tail -3 gpurun_out/r5_tests3.log
python tools/sweep.py bench --env BGS_ROLLOUT_WPS=2 --args "--inflight 3" --repeat 2 2>&1 | grep "^{'" 
python tools/sweep.py bench --env BGS_ROLLOUT_WPS=1 --args "--inflight 6" --repeat 2 2>&1 | grep "^{'"
python tools/sweep.py bench --env BGS_ROLLOUT_WPS=1 --args "--inflight 5" 2>&1 | grep "^{'"
python tools/sweep.py bench --env BGS_ROLLOUT_WPS=2 --args "--inflight 4" 2>&1 | grep "^{'"
python tools/sweep.py bench --env BGS_ROLLOUT_WPS=3 --args "--inflight 2" 2>&1 | grep "^{'"
python tools/sweep.py bench --env BGS_ROLLOUT_WPS=2 --args "--inflight 3 --steps 20 --warmup 5" --repeat 3 2>&1 | grep "^{'"
bash tools/count_valu.sh k2o python3 bench.py --steps 10 --warmup 2 --prewarm-ms 0 --no-cpu-baseline --no-device-resident --no-other-configs --no-repeats | head -2
timeout -k 10 400 python tools/api_rates.py > gpurun_out/r5_api_rates.json 2> gpurun_out/r5_api_rates.err; python - <<'PY'
import json
d=json.load(open("gpurun_out/r5_api_rates.json"))
for k,v in d["configs"].items():
    print(k, {l:(("%.3e"%x["value"]) if "value" in x else x) for l,x in v.items() if isinstance(x,dict)}, v.get("pipeline_over_executor"))
PY

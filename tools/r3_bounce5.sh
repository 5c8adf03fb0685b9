#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
P=$(pwd)/board-game-simulator-python_amd
for lib in libbgs.so libbgs_prio2.so; do
for cfg in "384:1,0:8 512" "384:1,0:8 0" "single 512" "single 0"; do
set -- $cfg
BGS_LIBRARY=$P/$lib BGS_BOUNCE_PLAN=$1 BGS_BOUNCE_FLAT_WAVES=$2 timeout -k 10 300 python tools/rollout_rate.py bounce --depth 16 --reps 64 > gpurun_out/r3b5_${lib}_$1_$2.json 2>> gpurun_out/r3b5.err
done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r3b5_*.json")):
    try:
        d=json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f, {k:("%.3e"%v["env_steps_per_s"], "%.3f ms"%(v["s_per_batch"]*1e3)) for k,v in d.items() if isinstance(v,dict) and "env_steps_per_s" in v})
    except Exception as e:
        print(f, "ERR", e)
PY

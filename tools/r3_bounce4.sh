#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
i=0
for plan in "256:1,0:8" "384:1,0:8" "512:1,0:8" "768:1,0:8"; do
for wv in 0 512 1024; do
i=$((i+1))
BGS_BOUNCE_PLAN=$plan BGS_BOUNCE_FLAT_WAVES=$wv timeout -k 10 300 python tools/rollout_rate.py bounce --depth 16 --reps 64 > gpurun_out/r3b4_${i}.json 2>> gpurun_out/r3b4.err
done
done
BGS_BOUNCE_PLAN="384:1,0:8" BGS_BOUNCE_FLAT_WAVES=512 timeout -k 10 300 python tools/rollout_rate.py bounce --depth 24 --reps 96 > gpurun_out/r3b4_d24.json 2>> gpurun_out/r3b4.err
BGS_BOUNCE_PLAN="384:1,0:8" BGS_BOUNCE_PIECES=0 timeout -k 10 300 python tools/rollout_rate.py bounce --depth 16 --reps 64 > gpurun_out/r3b4_k3f.json 2>> gpurun_out/r3b4.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r3b4_*.json")):
    try:
        d=json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f, d["env"].get("BGS_BOUNCE_PLAN"), d["env"].get("BGS_BOUNCE_FLAT_WAVES"), {k:("%.3e"%v["env_steps_per_s"], "%.3f ms"%(v["s_per_batch"]*1e3)) for k,v in d.items() if isinstance(v,dict) and "env_steps_per_s" in v})
    except Exception as e:
        print(f, "ERR", e)
PY

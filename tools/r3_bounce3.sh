#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
for mp in 512 4096; do
for cfg in "1 256 0" "1 256 512" "1 1024 0" "0 256 0"; do
set -- $cfg
BGS_BOUNCE_PIECES=$1 BGS_BOUNCE_BLOCK=$2 BGS_BOUNCE_FLAT_WAVES=$3 timeout -k 10 300 python tools/rollout_rate.py bounce --depth 16 --reps 64 --max-plies $mp > gpurun_out/r3b3_mp${mp}_p$1_b$2_w$3.json 2>> gpurun_out/r3b3.err
done
done
GPU_MAX_HW_QUEUES=64 BGS_BOUNCE_PIECES=1 timeout -k 10 300 python tools/rollout_rate.py bounce --depth 32 --reps 96 > gpurun_out/r3b3_d32_p1.json 2>> gpurun_out/r3b3.err
GPU_MAX_HW_QUEUES=64 BGS_BOUNCE_PIECES=0 timeout -k 10 300 python tools/rollout_rate.py bounce --depth 32 --reps 96 > gpurun_out/r3b3_d32_p0.json 2>> gpurun_out/r3b3.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r3b3_*.json")):
    try:
        d=json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f, {k:("%.3e"%v["env_steps_per_s"], "%.3f ms"%(v["s_per_batch"]*1e3)) for k,v in d.items() if isinstance(v,dict) and "env_steps_per_s" in v})
    except Exception as e:
        print(f, "ERR", e)
PY

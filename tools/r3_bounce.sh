#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "bounce" > gpurun_out/r3b_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r3b_tests.log
tail -5 gpurun_out/r3b_tests.log
for p in 1 0; do
BGS_BOUNCE_PIECES=$p timeout -k 10 300 python tools/rollout_rate.py bounce --depth 16 --reps 48 > gpurun_out/r3b_rate_p$p.json 2> gpurun_out/r3b_rate_p$p.err
BGS_BOUNCE_PIECES=$p timeout -k 10 300 python tools/rollout_rate.py bounce --depth 1 --reps 9 > gpurun_out/r3b_solo_p$p.json 2>> gpurun_out/r3b_rate_p$p.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r3b_*.json")):
    try:
        d=json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f, {k:("%.3e"%v["env_steps_per_s"], "%.3f ms"%(v["s_per_batch"]*1e3)) for k,v in d.items() if isinstance(v,dict) and "env_steps_per_s" in v})
    except Exception as e:
        print(f, "ERR", e)
PY

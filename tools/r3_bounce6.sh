#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -m gpu > gpurun_out/r3b6_tests.log 2>&1
echo "tests rc=$?"; tail -2 gpurun_out/r3b6_tests.log
timeout -k 10 300 python tools/rollout_rate.py bounce --depth 16 --reps 64 > gpurun_out/r3b6_rate.json 2>> gpurun_out/r3b6.err
timeout -k 10 300 python bench.py --only bounce_default --steps 32 > gpurun_out/r3b6_only.json 2>> gpurun_out/r3b6.err
bash tools/profile_kernel.sh bounce python3 tools/rollout_rate.py bounce --depth 1 --reps 6 > /dev/null 2>&1
python3 tools/summarize_profile.py bounce k_bounce | grep -v "WAIT\|BUSY"
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r3b6_*.json")):
    try:
        d=json.loads([l for l in open(f) if l.startswith("{")][-1])
        if "value" in d: print(f, "%.3e"%d["value"], d["ms_per_step"], d["solo"], d["parity_with_oracle"])
        else: print(f, {k:("%.3e"%v["env_steps_per_s"], "%.3f ms"%(v["s_per_batch"]*1e3)) for k,v in d.items() if isinstance(v,dict) and "env_steps_per_s" in v})
    except Exception as e:
        print(f, "ERR", e)
PY

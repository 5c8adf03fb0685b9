#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "connect" > gpurun_out/r3k_tests.log 2>&1
echo "tests rc=$?"; tail -3 gpurun_out/r3k_tests.log
timeout -k 10 300 python tools/rollout_rate.py connect12x13 --depth 3 --reps 60 > gpurun_out/r3k_rate.json 2> gpurun_out/r3k.err
bash tools/profile_kernel.sh k2c python3 tools/rollout_rate.py connect12x13 --depth 1 --reps 9 > /dev/null 2>&1
python3 tools/summarize_profile.py k2c k_connect_rollout_lds | grep "INSTS_VALU\|THREAD_CYCLES\|mean_us"
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r3k_rate.json") if l.startswith("{")][-1])
print({k:("%.3e"%v["env_steps_per_s"], "%.3f ms"%(v["s_per_batch"]*1e3)) for k,v in d.items() if isinstance(v,dict) and "env_steps_per_s" in v})
PY

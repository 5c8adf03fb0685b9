#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "bounce" > gpurun_out/r3b7_tests.log 2>&1
echo "tests rc=$?"; tail -2 gpurun_out/r3b7_tests.log
BGS_LIBRARY=$PWD/board-game-simulator-python_amd/libbgs_stats.so python tools/bounce_stats.py 2>/dev/null | head -10
timeout -k 10 300 python tools/rollout_rate.py bounce --depth 16 --reps 64 > gpurun_out/r3b7_rate.json 2>> gpurun_out/r3b7.err
bash tools/profile_kernel.sh bounce python3 tools/rollout_rate.py bounce --depth 1 --reps 6 > /dev/null 2>&1
python3 tools/summarize_profile.py bounce k_bounce_rollout_pieces | grep "INSTS_VALU\|THREAD_CYCLES\|mean_us"
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r3b7_*.json")):
    d=json.loads([l for l in open(f) if l.startswith("{")][-1])
    print(f, {k:("%.3e"%v["env_steps_per_s"], "%.3f ms"%(v["s_per_batch"]*1e3)) for k,v in d.items() if isinstance(v,dict) and "env_steps_per_s" in v})
PY

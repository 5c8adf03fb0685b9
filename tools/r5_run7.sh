#!/bin/bash
set -u
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -k "bounce" -x -q > gpurun_out/r5_book_tests2.log 2>&1; echo "bounce tests rc=$?"; tail -3 gpurun_out/r5_book_tests2.log
for rep in 1 2 3; do for D in 20 8; do
  out=$(python tools/rollout_rate.py bounce --depth $D --reps 120 2>/dev/null | grep '^{' | tail -1)
  python - "$out" $D <<'PY'
import json, sys
d = json.loads(sys.argv[1])
print("depth", sys.argv[2], "solo %.3e" % d["one_launch_at_a_time"]["env_steps_per_s"], "pipelined %.4e" % d[f"{sys.argv[2]}_in_flight"]["env_steps_per_s"])
PY
done; done
bash tools/count_valu.sh book4b python3 tools/rollout_rate.py bounce --depth 1 --reps 6 --hint 20 | head -3

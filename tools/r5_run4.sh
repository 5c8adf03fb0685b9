#!/bin/bash
# round 5, GPU run 4: the opening book's parity tests, its rate (book off / depth 3 / depth 4), K2o tuning sweeps
set -u
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -k "bounce" -x -q > gpurun_out/r5_book_tests.log 2>&1; echo "book tests rc=$?"; tail -3 gpurun_out/r5_book_tests.log
for B in 0 3 4; do
  for D in 20 8; do
    BGS_BOUNCE_BOOK=$B timeout -k 10 200 python tools/rollout_rate.py bounce --depth $D --reps 120 > gpurun_out/r5_book_${B}_d$D.json 2> gpurun_out/r5_book_${B}_d$D.err
    python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/r5_book_${B}_d$D.json") if l.startswith("{")][-1])
print("book $B depth $D", {k:(v.get("env_steps_per_s") if isinstance(v,dict) else v) for k,v in d.items() if "flight" in k or "launch" in k})
PY
  done
done
BGS_BOUNCE_BOOK=0 bash tools/count_valu.sh book0 python3 tools/rollout_rate.py bounce --depth 1 --reps 6 --hint 20 | tail -4
BGS_BOUNCE_BOOK=4 bash tools/count_valu.sh book4 python3 tools/rollout_rate.py bounce --depth 1 --reps 6 --hint 20 | tail -4
bash tools/count_valu.sh k2o python3 bench.py --steps 10 --warmup 2 --prewarm-ms 0 --no-cpu-baseline --no-device-resident --no-other-configs --no-repeats | tail -3
python tools/sweep.py bench --env BGS_ROLLOUT_OPENING=2,3,4 --repeat 2 2>&1 | tail -8
python tools/sweep.py bench --env BGS_ROLLOUT_CHUNK=256,512,1024 2>&1 | tail -5

#!/bin/bash
# Bounce: which plan wins at which number of launches in flight
mkdir -p gpurun_out
out=gpurun_out/bounce_depth.txt; : > $out
for depth in 4 8 16; do
 for cfg in "auto 0" "64:1,4096:8 2048" "64:1,4096:8 1024" "128:1,4096:8 1024" "192:1,4096:8 512" "384:1,4096:8 1024"; do
  set -- $cfg
  BGS_BOUNCE_PLAN=$1 timeout -k 10 120 python tools/rollout_rate.py bounce --depth $depth --reps 48 --bounce-waves $2 > gpurun_out/_solo.json 2>/dev/null || { echo "$cfg failed" >> $out; continue; }
  python - "$1" "$2" "$depth" >> $out <<'PY'
import json, sys
d = json.load(open("gpurun_out/_solo.json"))
k = [k for k in d if k.endswith("_in_flight")][0]
print(f"depth {sys.argv[3]:3s} plan {sys.argv[1]:18s} waves {sys.argv[2]:5s} solo {d['one_launch_at_a_time']['env_steps_per_s']/1e9:5.2f}  {k} {d[k]['env_steps_per_s']/1e9:5.2f} G/s")
PY
  tail -1 $out
 done
done

#!/bin/bash
# Bounce, ONE launch at a time: bulk-pass ply cap x waves per launch (the pipelined optimum is 384 plies, 512 waves)
mkdir -p gpurun_out
out=gpurun_out/bounce_solo.txt; : > $out
for plan in auto "32:1,4096:8" "64:1,4096:8" "96:1,4096:8" "128:1,4096:8" "192:1,4096:8" "48:1,384:1,4096:8" "32:1,128:1,4096:8"; do
  for waves in 0 1024 2048 4096; do
    BGS_BOUNCE_PLAN=$plan timeout -k 10 120 python tools/rollout_rate.py bounce --depth 2 --reps 12 --bounce-waves $waves > gpurun_out/_solo.json 2>/dev/null || { echo "$plan $waves failed" >> $out; continue; }
    python - "$plan" "$waves" >> $out <<'PY'
import json, sys
d = json.load(open("gpurun_out/_solo.json"))
solo = d["one_launch_at_a_time"]; two = d.get("2_in_flight", {})
print(f"plan {sys.argv[1]:22s} waves {sys.argv[2]:5s} solo {solo['s_per_batch']*1e3:6.2f} ms {solo['env_steps_per_s']/1e9:5.2f} G/s   2 in flight {two.get('env_steps_per_s',0)/1e9:5.2f} G/s")
PY
    tail -1 $out
  done
done

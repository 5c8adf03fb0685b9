#!/bin/bash
# games per wave x opening blocks x batches in flight for the Connect 6x7x4 rollout: one launch at a time | D in flight
for o in ${OPENINGS:-2 4}; do for c in ${CHUNKS:-256 512 1024 2048 4096}; do for d in ${DEPTHS:-3 4}; do
  r=$(BGS_ROLLOUT_OPENING=$o BGS_ROLLOUT_CHUNK=$c timeout -k 10 120 python3 tools/rollout_rate.py connect6x7 --depth $d --reps ${REPS:-180} 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=[x for x in d if x.endswith('in_flight')][0]
print('%.1f | %.1f' % (d['one_launch_at_a_time']['env_steps_per_s']/1e9, d[k]['env_steps_per_s']/1e9))")
  echo "opening $o chunk $c depth $d: $r"
done; done; done

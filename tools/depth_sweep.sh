#!/bin/bash
# rollout rate of Connect 6x7x4 at several pipeline depths for each "NAME=VALUE,..." setting in SETTINGS
for setting in ${SETTINGS}; do for d in ${DEPTHS:-3 4 6 8}; do
  ( IFS=,; for kv in $setting; do export "$kv"; done
    echo "$setting depth $d: $(python3 tools/rollout_rate.py connect6x7 --depth $d --reps ${REPS:-240} 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); k=[x for x in d if x.endswith('in_flight')][0]
print('%.1f | %.1f' % (d['one_launch_at_a_time']['env_steps_per_s']/1e9, d[k]['env_steps_per_s']/1e9))")" )
done; done

#!/bin/bash
# Connect, ONE launch at a time: games per wave (BGS_ROLLOUT_CHUNK)
for cfg in connect12x13 connect6x7; do
 for chunk in 0 64 128 192 256 384 512; do
  BGS_ROLLOUT_CHUNK=$chunk timeout -k 10 120 python tools/rollout_rate.py $cfg --depth 3 --reps 45 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=[k for k in d if k.endswith('_in_flight')][0]
print('$cfg chunk $chunk  one at a time %.1f G/s (%.1f us)   %s %.1f G/s' % (d['one_launch_at_a_time']['env_steps_per_s']/1e9, d['one_launch_at_a_time']['s_per_batch']*1e6, k, d[k]['env_steps_per_s']/1e9))"
 done
done

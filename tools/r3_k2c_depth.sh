#!/bin/bash
# K2c: launches in flight x games per wave (three waves per SIMD issue 0.43 of peak: would more waves help?)
export GPU_MAX_HW_QUEUES=32
for depth in 6 8 10 12 16; do
 for chunk in 0 384 512; do
  BGS_ROLLOUT_CHUNK=$chunk timeout -k 10 120 python tools/rollout_rate.py connect12x13 --depth $depth --reps $((depth * 20)) 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=[k for k in d if k.endswith('_in_flight')][0]
print('depth $depth chunk $chunk  %s %.1f G/s  queues %s' % (k, d[k]['env_steps_per_s']/1e9, d['env'].get('GPU_MAX_HW_QUEUES')))"
 done
done
# and the headline kernel with more queues / depth
for depth in 3 4 6; do
  timeout -k 10 120 python tools/rollout_rate.py connect6x7 --depth $depth --reps $((depth * 30)) 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=[k for k in d if k.endswith('_in_flight')][0]
print('connect6x7 depth $depth  %s %.1f G/s  queues %s' % (k, d[k]['env_steps_per_s']/1e9, d['env'].get('GPU_MAX_HW_QUEUES')))"
done

#!/bin/bash
# Connect(12,13,5), 2^18 boards: launches in flight x games per wave (rollout_chunk; 0 = the library's choice), device rate.
for d in 6 8 10 12; do for c in 0 128 192 256 384 512; do
  BGS_EXPERIMENT="rollout_chunk=$c" GPU_MAX_HW_QUEUES=16 python3 tools/rollout_rate.py connect12x13 --depth $d --reps $((d*20)) 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=[x for x in d if x.endswith('_in_flight')][0]
print('depth $d chunk $c:', k, '%.4g' % d[k]['env_steps_per_s'])"
done; done

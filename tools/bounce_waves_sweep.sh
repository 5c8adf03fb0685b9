#!/bin/bash
# Bounce 9x6, 2^18 boards: waves per launch x batches in flight (one launch at a time | D in flight, G env-steps/s)
for w in ${WAVES:-2048 1024 512 256}; do for d in ${DEPTHS:-8 16}; do
  echo "waves $w depth $d: $(BGS_BOUNCE_FLAT_WAVES=$w timeout -k 10 300 python3 tools/rollout_rate.py bounce --depth $d --reps ${REPS:-48} 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); k=[x for x in d if x.endswith('in_flight')][0]
print('%.2f | %.2f' % (d['one_launch_at_a_time']['env_steps_per_s']/1e9, d[k]['env_steps_per_s']/1e9))")"
done; done

#!/bin/bash
# Bounce 9x6, 2^18 boards: parking threshold x waves per launch (one launch at a time | DEPTH in flight, G env-steps/s)
for p in ${PARKS:-0 32}; do for w in ${WAVES:-2048 512}; do
  echo "park $p waves $w: $(BGS_BOUNCE_PARK=$p timeout -k 10 300 python3 tools/rollout_rate.py bounce --depth ${DEPTH:-16} --reps ${REPS:-192} --bounce-waves $w 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); k=[x for x in d if x.endswith('in_flight')][0]
print('%.2f | %.2f' % (d['one_launch_at_a_time']['env_steps_per_s']/1e9, d[k]['env_steps_per_s']/1e9))")"
done; done

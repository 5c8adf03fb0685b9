#!/bin/bash
# Bounce at 20 launches in flight: bulk cap x waves per launch around the shape bounce_shape() picks
mkdir -p gpurun_out
for cfg in "160:1,4096:8 512" "160:1,4096:8 448" "160:1,4096:8 384" "160:1,4096:8 320" "160:1,4096:8 256" "128:1,4096:8 384" "192:1,4096:8 384" "160:1,4096:8 384" "160:1,4096:8 512"; do
  set -- $cfg
  BGS_BOUNCE_PLAN=$1 timeout -k 10 120 python tools/rollout_rate.py bounce --depth 20 --reps 160 --bounce-waves $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=[k for k in d if k.endswith('_in_flight')][0]
print('plan $1 waves $2: %s %.2f G/s' % (k, d[k]['env_steps_per_s']/1e9))"
done

#!/bin/bash
# Bounce default, D launches in flight (default 8, HIP's default hardware queues): the bulk pass's ply cap x the launch shape
# of a hint (1: 128 boards a wave, 8: 256, 20: 512), K3w behind it.   bash tools/bounce_shape_probe.sh [D] [reps]
D=${1:-8}; R=${2:-2}
run() { h=$1; plan=$2; BGS_EXPERIMENT="bounce_plan=$plan" python3 tools/rollout_rate.py bounce --depth $D --reps $((D*20)) --hint $h 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=[x for x in d if x.endswith('_in_flight')][0]
print('hint $h plan $plan: %.4g' % d[k]['env_steps_per_s'])"; }
for i in $(seq $R); do
  for h in 8 20 1; do
    run $h auto
    for cap in 64 96 128 160 224; do run $h "$cap:1,4096:64"; done
  done
done

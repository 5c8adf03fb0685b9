#!/bin/bash
# A/B of one environment knob on a rollout_rate.py configuration, alternating:  bash tools/ab_env.sh NAME v1 v2 <config> <depth> [reps]
name=$1; a=$2; b=$3; cfg=$4; depth=$5; reps=${6:-3}
for i in $(seq $reps); do for v in $a $b; do
  env $name=$v python3 tools/rollout_rate.py $cfg --depth $depth --reps $((8*depth)) 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=[x for x in d if x.endswith('_in_flight')][0]
print('$name=$v', 'solo %.4g' % d['one_launch_at_a_time']['env_steps_per_s'], k, '%.4g' % d[k]['env_steps_per_s'])"
done; done

#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "bounce" > gpurun_out/bounce_tests.log 2>&1 || { tail -30 gpurun_out/bounce_tests.log; exit 1; }
tail -2 gpurun_out/bounce_tests.log
for depth in 1 2 4 8 16; do
  timeout -k 10 120 python tools/rollout_rate.py bounce --depth $depth --reps $((depth * 6 + 6)) 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=[k for k in d if k.endswith('_in_flight')][0]
print('depth $depth  one at a time %.2f G/s   %s %.2f G/s' % (d['one_launch_at_a_time']['env_steps_per_s']/1e9, k, d[k]['env_steps_per_s']/1e9))"
done

for cfg in connect6x7 connect12x13; do for w in 1 2 3 4 6 8; do
  BGS_ROLLOUT_WPS=$w python3 tools/rollout_rate.py $cfg --depth 1 --reps 20 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=[x for x in d if x.endswith('_in_flight')][0]
print('$cfg wps $w', 'one at a time %.4g  (%.1f us)' % (d['one_launch_at_a_time']['env_steps_per_s'], d['one_launch_at_a_time']['s_per_batch']*1e6), k, '%.4g' % d[k]['env_steps_per_s'])"
done; done

for rep in 1 2; do
for pol in 0:0 8:24 4:28 8:56 16:48 4:60 2:30; do
  BGS_EXPERIMENT="bounce_memo_policy=$pol" python3 tools/rollout_rate.py bounce --depth 8 --reps 160 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=[x for x in d if x.endswith('_in_flight')][0]
print('policy $pol', 'solo %.4g' % d['one_launch_at_a_time']['env_steps_per_s'], k, '%.4g' % d[k]['env_steps_per_s'])"
done; done

#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
for c in 128 256 384 512 1024; do
BGS_ROLLOUT_CHUNK=$c timeout -k 10 300 python tools/rollout_rate.py connect12x13 --depth 3 --reps 60 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('chunk', $c, {k:('%.3e'%v['env_steps_per_s'], '%.3f ms'%(v['s_per_batch']*1e3)) for k,v in d.items() if isinstance(v,dict) and 'env_steps_per_s' in v})"
done
for d in 2 4 6; do
timeout -k 10 300 python tools/rollout_rate.py connect12x13 --depth $d --reps 60 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('depth', $d, {k:('%.3e'%v['env_steps_per_s'], '%.3f ms'%(v['s_per_batch']*1e3)) for k,v in d.items() if isinstance(v,dict) and 'env_steps_per_s' in v})"
done

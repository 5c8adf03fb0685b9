#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
for cfg in "0 16" "384 16" "768 16" "1024 16" "0 12" "0 20" "0 24"; do
set -- $cfg
BGS_BOUNCE_FLAT_WAVES=$1 timeout -k 10 300 python tools/rollout_rate.py bounce --depth $2 --reps $((4*$2)) 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('waves $1 depth $2', {k:('%.3e'%v['env_steps_per_s'], '%.3f ms'%(v['s_per_batch']*1e3)) for k,v in d.items() if isinstance(v,dict) and 'env_steps_per_s' in v})"
done

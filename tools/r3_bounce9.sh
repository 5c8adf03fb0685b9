#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
for lib in libbgs.so libbgs_w5.so libbgs_w6.so libbgs.so; do
BGS_LIBRARY=$PWD/board-game-simulator-python_amd/$lib timeout -k 10 300 python tools/rollout_rate.py bounce --depth 16 --reps 64 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$lib', {k:('%.3e'%v['env_steps_per_s'], '%.3f ms'%(v['s_per_batch']*1e3)) for k,v in d.items() if isinstance(v,dict) and 'env_steps_per_s' in v})"
done

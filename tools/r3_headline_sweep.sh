#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
B="--no-cpu-baseline --no-other-configs --no-repeats --no-device-resident"
for d in 2 3 4 6; do for w in 1 2 3 4; do
BGS_ROLLOUT_WPS=$w python bench.py --inflight $d $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('inflight $d wps $w', '%.3e'%d['value'], '%.4f'%d['ms_per_step'])"
done; done
for c in 256 1024; do
BGS_ROLLOUT_CHUNK=$c python bench.py $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('chunk $c', '%.3e'%d['value'], '%.4f'%d['ms_per_step'])"
done

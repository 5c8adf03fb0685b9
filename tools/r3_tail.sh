#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
B="--no-cpu-baseline --no-other-configs --steps 20 --warmup 5"
for t in 0 3 4 6 0 4; do
BGS_PIPELINE_TAIL_WPS=$t python bench.py $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('tail wps $t', '%.3e'%d['value'], ['%.3e'%v for v in d['values_of_3']], 'dev %.3e'%d['device_resident']['value'])"
done

#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
B="--no-cpu-baseline --no-other-configs --steps 20 --warmup 5"
for p in 32 5 2; do
for i in 1 2; do
BGS_BENCH_PAIRS=$p python bench.py $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('pairs', $p, '%.3e'%d['value'], ['%.3e'%v for v in d['values_of_3']], 'dev %.3e'%d['device_resident']['value'], d['roofline']['event_pairs'])"
done
done

#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
for f in 2 3 4 2; do
BGS_BENCH_OTHER_SLOT_FACTOR=$f python bench.py --only bounce_default --steps 48 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('factor $f', '%.3e'%d['value'], 'dev %.3e'%d['device_resident'], d['parity_with_oracle'])"
done

#!/bin/bash
# host arrays / sink slots per stream (BGS_BENCH_SLOT_FACTOR) vs the bench value, 20-step and 200-step runs
for f in ${FACTORS:-2 3}; do for args in "--steps 20 --warmup 5" "--steps 200 --warmup 20"; do echo -n "slot factor $f, $args: "; for i in 1 2 3 4 5; do BGS_BENCH_SLOT_FACTOR=$f python3 bench.py $args --no-cpu-baseline --no-device-resident 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.0f' % (d['value']/1e9), end=' ')"; done; echo; done; done

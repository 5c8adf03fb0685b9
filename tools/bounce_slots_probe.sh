#!/bin/bash
# bench.py's Bounce configuration against the number of sink slots / host arrays per stream (deliveries complete in ticket
# order; a step's duration varies with its longest games).
for f in 2 3 5 8; do for i in 1 2; do
  BGS_BENCH_OTHER_SLOT_FACTOR=$f python3 bench.py --only ${CFG:-bounce_default} --steps ${STEPS:-200} 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('slots per stream $f', '%.4g'%d['value'], 'device %.4g'%d['device_resident'], d['parity_with_oracle'])"
done; done

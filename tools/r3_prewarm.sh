#!/bin/bash
# the first 20-step region against its two repeats, by length of the pre-warm
for round in 1 2; do
for ms in 150 400 800 1500; do
  for i in 1 2 3; do
  python bench.py --steps 20 --warmup 5 --prewarm-ms $ms --no-cpu-baseline --no-other-configs --no-device-resident 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('prewarm $ms ms: of3 %s' % ([round(v/1e9,1) for v in d.get('values_of_3',[])]))"
  done
done
done

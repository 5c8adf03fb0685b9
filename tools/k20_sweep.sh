#!/bin/bash
# fixed cost of short bench runs: value / total time for a few argument sets (ARGS_LIST separated by ';')
IFS=';' read -ra SETS <<< "${ARGS_LIST:---steps 20 --warmup 5;--steps 40 --warmup 5;--steps 80 --warmup 5}"
for args in "${SETS[@]}"; do
  python3 bench.py $args --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$args', '| value %.1f G  ms/step %.4f  total %.3f ms | device %.1f G' % (d['value']/1e9, d['ms_per_step'], d['ms_per_step']*d['steps'], d.get('device_resident',{}).get('value',0)/1e9))"
done

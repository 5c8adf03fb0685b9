#!/bin/bash
# how the sink's waiters wait (sleep / poll the event / spin before sleeping) vs the bench value, 20- and 200-step runs
for cfg in "BGS_SINK_POLL=0 BGS_SINK_SPIN_US=0" "BGS_SINK_POLL=1 BGS_SINK_SPIN_US=0" "BGS_SINK_POLL=1 BGS_SINK_SPIN_US=100" "BGS_SINK_POLL=1 BGS_SINK_SPIN_US=1000"; do
  for args in "--steps 20 --warmup 5" "--steps 200 --warmup 20"; do
    echo -n "$cfg, $args: "
    for i in 1 2 3 4 5; do env $cfg python3 bench.py $args --no-cpu-baseline --no-device-resident 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.0f' % (d['value']/1e9), end=' ')"; done; echo
  done
done

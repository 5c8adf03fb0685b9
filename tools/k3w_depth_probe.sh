#!/bin/bash
# Bounce, launches in flight x launch shape (the --hint: 1 = {64 plies, 128 boards a wave}, 8 = {128, 256}, 20 = {160, 512}) with
# the K3w tail pass; run with GPU_MAX_HW_QUEUES=24.
run() { d=$1; h=$2; shift; shift; env "$@" python3 tools/rollout_rate.py bounce --depth $d --reps $((d*40)) --hint $h | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); k=[x for x in d if x.endswith('_in_flight')][0]; print('depth $d shape-of $h $*: %.3e' % d[k]['env_steps_per_s'])"; }
for d in 2 4 6 8 12 16 20; do for h in 1 8 20; do run $d $h BGS_X=1; done; done

#!/bin/bash
# the N > 1 code path on a one-GPU box, 20-step regions (what a driver's scaling run times)
D="RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 BGS_FORCE_DIST=1"
for g in shm rccl shm rccl; do env $D MASTER_PORT=$((29500 + RANDOM % 400)) python bench.py --gpus 1 --gather $g --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$g 20 steps: %.1f G/s of3 %s verified %s' % (d['value']/1e9, [round(v/1e9,1) for v in d.get('values_of_3',[])], d['config'].get('gathered_rewards_verified')))"; done
BGS_DIST_BACKEND=gloo python bench.py --gpus 3 --steps 40 --batch 262144 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('gloo3: %.1f G/s %s verified %s' % (d['value']/1e9, [round(v/1e9,1) for v in d.get('values_of_3',[])], d['config'].get('gathered_rewards_verified')))"
timeout -k 10 600 python -m pytest tests -x -q -m gpu -k "handover or pipeline" 2>&1 | tail -2

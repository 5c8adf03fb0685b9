#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
D="RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 BGS_FORCE_DIST=1"
for p in 32 2 32 2; do
env $D MASTER_PORT=$((29500 + RANDOM % 400)) BGS_BENCH_PAIRS=$p python bench.py --gpus 1 --gather rccl --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('pairs $p', '%.3e'%d['value'], ['%.3e'%v for v in d['values_of_3']], 'dev %.3e'%d['device_resident']['value'])"
done

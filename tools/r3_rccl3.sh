#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
D="RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 BGS_FORCE_DIST=1 BGS_BENCH_TRACE=1"
for cfg in "1 4" "1 8" "3 4" "6 4" "1 3"; do
set -- $cfg
env $D MASTER_PORT=$((29500 + RANDOM % 400)) BGS_GATHER_BATCH=$1 BGS_BENCH_SLOT_FACTOR=$2 python bench.py --gpus 1 --gather rccl --no-cpu-baseline --no-repeats --no-device-resident 2> gpurun_out/r3r3.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('batch $1 factor $2', '%.3e'%d['value'])"
grep trace gpurun_out/r3r3.err | tail -1
done

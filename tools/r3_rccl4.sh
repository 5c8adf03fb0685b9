#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
D="RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 BGS_FORCE_DIST=1 BGS_BENCH_TRACE=1"
for v in "" "BGS_GATHER_PROBE_INLINE=1"; do
env $D MASTER_PORT=$((29500 + RANDOM % 400)) $v python bench.py --gpus 1 --gather rccl --no-cpu-baseline --no-repeats --no-device-resident 2> gpurun_out/r3r4.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('[$v]', '%.3e'%d['value'], d['config']['gathered_rewards_verified'])"
grep trace gpurun_out/r3r4.err | tail -1
done

#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
D="RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 BGS_FORCE_DIST=1"
stat() { grep -E "nr_throttled|throttled_usec|usage_usec" /sys/fs/cgroup/cpu.stat | tr '\n' ' '; echo; }
for g in shm rccl rccl; do
echo "before $g: $(stat)"
env $D MASTER_PORT=$((29500 + RANDOM % 400)) python bench.py --gpus 1 --gather $g --no-cpu-baseline --no-repeats --no-device-resident 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$g', '%.3e'%d['value'])"
echo "after  $g: $(stat)"
done
echo "threads while running rccl:"
env $D MASTER_PORT=29777 python bench.py --gpus 1 --gather rccl --no-cpu-baseline --steps 4000 --no-repeats --no-device-resident > /dev/null 2>&1 &
pid=$!
sleep 12
top -H -b -n 1 -p $pid | head -30
wait $pid

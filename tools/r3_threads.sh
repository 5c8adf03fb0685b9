#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
B="--no-cpu-baseline --no-other-configs --steps 20 --warmup 5"
for t in 6 10 14 6 10 14; do
BGS_BENCH_TRACE=1 python bench.py $B --host-threads $t 2> gpurun_out/r3t.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('threads $t', '%.3e'%d['value'], ['%.3e'%v for v in d['values_of_3']], 'dev %.3e'%d['device_resident']['value'])"
grep trace gpurun_out/r3t.err | head -1
done

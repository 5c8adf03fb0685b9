#!/bin/bash
# does what ran before a 20-step bench change its result?
k20() { BGS_BENCH_TRACE=1 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>gpurun_out/k20.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('   K=20: value %.1f G  total %.3f ms | device %.1f G' % (d['value']/1e9, d['ms_per_step']*d['steps'], d['device_resident']['value']/1e9))"; grep trace gpurun_out/k20.err | head -1; }
echo "alone"; k20; k20
echo "after a 200-step bench without cpu baseline"; python3 bench.py --no-cpu-baseline > /dev/null 2>&1; k20
echo "after a 200-step bench with cpu baseline"; python3 bench.py > /dev/null 2>&1; k20; k20
echo "after 20 s of 16 busy CPU threads"; python3 -c "
import multiprocessing as mp, time
def burn(_):
    t=time.time(); x=0
    while time.time()-t<20: x+=1
    return x
with mp.Pool(16) as p: p.map(burn, range(16))"; k20; k20

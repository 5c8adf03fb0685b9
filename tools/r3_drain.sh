#!/bin/bash
# the end of a short run: every stream synchronised by a helper thread from the start of the drain, or by the draining thread at its end
mkdir -p gpurun_out
for knob in "X=1" "BGS_DRAIN_SERIAL_SYNC=1" "X=1" "BGS_DRAIN_SERIAL_SYNC=1"; do
  echo "== $knob"
  env $knob BGS_SINK_TRACE=1 python tools/short_run_timeline.py 20 5 > gpurun_out/_t.json 2> gpurun_out/sink_trace.txt
  python - <<'PY'
import re
lines=open('gpurun_out/sink_trace.txt').read().splitlines()
land={}; exp={}
for l in lines:
    m=re.match(r'sink-trace ticket (\d+) landed ([\d.]+)',l)
    if m: land[int(m.group(1))]=float(m.group(2))
    m=re.match(r'sink-trace ticket (\d+) worker (\d+) expand ([\d.]+) \.\. ([\d.]+)',l)
    if m: exp.setdefault(int(m.group(1)),[]).append(float(m.group(4)))
for h in [l for l in lines if l.startswith('host-trace')][1:]:
    m=re.search(r't0 ([\d.]+) enqueue_returns ([\d.]+) drain_returns ([\d.]+) synchronized ([\d.]+)', h)
    t0,te,td,ts=map(float,m.groups())
    last=max(t for t in land if land[t]<=td+1)
    print('  landed %.1f expanded %.1f drain returns %.1f synchronized %.1f' % (land[last]-t0, max(exp[last])-t0, td-t0, ts-t0))
PY
  for i in 1 2 3; do env $knob python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --no-device-resident 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('  bench 20: of3 %s' % ([round(v/1e9,1) for v in d.get('values_of_3',[])]))"; done
done

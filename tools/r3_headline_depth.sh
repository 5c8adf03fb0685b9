#!/bin/bash
# headline (Connect 6x7x4, 2^20): batches in flight with 32 hardware queues, 200-step and 20-step regions
mkdir -p gpurun_out
out=gpurun_out/headline_depth.txt; : > $out
for depth in 3 4 5 6 8 9 12; do
 for steps in "200 10" "20 5"; do
  set -- $steps
  python bench.py --steps $1 --warmup $2 --inflight $depth --no-cpu-baseline --no-other-configs --no-device-resident > gpurun_out/_hd.json 2>/dev/null
  python - $depth $1 >> $out <<'PY'
import json, sys
d = json.loads([l for l in open("gpurun_out/_hd.json") if l.startswith("{")][-1])
print(f"inflight {sys.argv[1]:3s} steps {sys.argv[2]:4s} value {d['value']/1e9:6.1f} G/s  of3 {[round(v/1e9,1) for v in d.get('values_of_3',[])]}  queues {d['config'].get('gpu_max_hw_queues')}")
PY
  tail -1 $out
 done
done

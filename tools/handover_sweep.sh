#!/bin/bash
# Hand-over mode x batches in flight x host threads, one bench.py run each (no CPU baseline); one JSON line per run
# into gpurun_out/handover_sweep.jsonl.  Usage on the GPU box: bash tools/handover_sweep.sh
set -u
out=gpurun_out/handover_sweep.jsonl
mkdir -p gpurun_out
: > $out
run() {
  python bench.py --steps 300 --warmup 20 --no-cpu-baseline "$@" 2>>gpurun_out/handover_sweep.err | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({'args': '$*', 'value': d['value'], 'ms_per_step': d['ms_per_step'], 'kernel_ms': d['roofline']['kernel_ms_per_launch'], 'device_resident': (d.get('device_resident') or {}).get('value')}))" >> $out
  tail -1 $out
}
for rep in 1 2; do
  run --handover none --inflight 2 --no-device-resident
  run --handover none --inflight 3 --no-device-resident
  run --handover pairs --inflight 2 --no-device-resident
  for d in 2 3 4; do
    for t in 1 2 3 4 6; do
      run --handover codes --inflight $d --host-threads $t --no-device-resident
    done
  done
done
run --handover codes --inflight 3 --host-threads 3

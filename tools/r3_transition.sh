#!/bin/bash
# object API round trip: staged copies vs host-mapped blocks vs one replayed HIP graph (BGS_TRANSITION)
set -e
mkdir -p gpurun_out
for mode in staged mapped graph fused; do
  echo "== $mode"
  BGS_TRANSITION=$mode timeout -k 10 200 python tools/object_latency.py > gpurun_out/object_latency_$mode.json
  cat gpurun_out/object_latency_$mode.json
done
timeout -k 10 600 python -m pytest tests -x -q -m gpu -k "dropin or generic or object or transition or json or thread or branching" > gpurun_out/transition_tests.log 2>&1 || { tail -30 gpurun_out/transition_tests.log; exit 1; }
tail -3 gpurun_out/transition_tests.log
BGS_TRANSITION=mapped timeout -k 10 600 python -m pytest tests -x -q -m gpu -k "dropin or generic" > gpurun_out/transition_tests_mapped.log 2>&1 || { tail -30 gpurun_out/transition_tests_mapped.log; exit 1; }
tail -3 gpurun_out/transition_tests_mapped.log

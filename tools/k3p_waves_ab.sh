#!/bin/bash
# A/B of K3p's occupancy cap (amdgpu_waves_per_eu): the library in the tree against a variant built with
#   make -C csrc EXTRA_CXXFLAGS=-DBGS_K3P_WAVES=5 OUT=<repo>/gpurun_lib/libbgs_w5.so     (in a scratch copy of csrc/)
# alternating, Bounce default, 2^18 boards, 20 and 8 launches in flight (tools/rollout_rate.py).
set -u
for rep in 1 2 3; do
  for lib in "" gpurun_lib/libbgs_w5.so; do
    for D in 20 8; do
      if [ -n "$lib" ]; then export BGS_LIBRARY=$PWD/$lib; else unset BGS_LIBRARY; fi
      out=$(python tools/rollout_rate.py bounce --depth $D --reps 120 2>/dev/null | grep '^{' | tail -1)
      python - "$out" "${lib:-tree}" $D <<'PY'
import json, sys
d = json.loads(sys.argv[1])
print(sys.argv[2], "depth", sys.argv[3], "solo %.3e" % d["one_launch_at_a_time"]["env_steps_per_s"], "pipelined %.4e" % d[f"{sys.argv[3]}_in_flight"]["env_steps_per_s"])
PY
    done
  done
done

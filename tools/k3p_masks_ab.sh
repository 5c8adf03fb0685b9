#!/bin/bash
# A/B on one box: K3p with the board's cell masks in vector registers (the tree) against scalar registers
# (gpurun_lib/libbgs_nomask.so: make EXTRA_CXXFLAGS=-DBGS_K3P_VECTOR_MASKS=0 in a scratch copy of csrc/), each with the
# parking threshold at 32 and at 40; Bounce default, 2^18 boards, 20 and 8 launches in flight.
set -u
for rep in 1 2 3; do
  for lib in "" gpurun_lib/libbgs_nomask.so; do
    for park in 32 40; do
      for D in 20 8; do
        if [ -n "$lib" ]; then export BGS_LIBRARY=$PWD/$lib; else unset BGS_LIBRARY; fi
        out=$(BGS_EXPERIMENT="bounce_pieces_park=$park" python tools/rollout_rate.py bounce --depth $D --reps 120 2>/dev/null | grep '^{' | tail -1)
        python - "$out" "${lib:-vector-masks}" $D $park <<'PY'
import json, sys
d = json.loads(sys.argv[1])
print(sys.argv[2], "park", sys.argv[4], "depth", sys.argv[3], "solo %.3e" % d["one_launch_at_a_time"]["env_steps_per_s"], "pipelined %.4e" % d[f"{sys.argv[3]}_in_flight"]["env_steps_per_s"])
PY
      done
    done
  done
done

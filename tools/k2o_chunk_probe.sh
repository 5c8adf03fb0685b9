#!/bin/bash
# K2o (Connect4 6x7x4, 2^20 boards): VALU instructions per launch against the games a wave owns (BGS_EXPERIMENT rollout_chunk).
for c in 128 256 512 1024 2048 4096; do
  BGS_EXPERIMENT="rollout_chunk=$c" bash tools/count_valu.sh k2o_c$c python3 tools/rollout_rate.py connect6x7 --depth 1 --reps 9 | grep rollout_opened | sed "s/^/chunk $c: /" | cut -c1-30,130-260
done

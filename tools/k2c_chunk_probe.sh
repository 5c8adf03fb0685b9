#!/bin/bash
# K2c (Connect 12x13x5, 2^18 boards): VALU instructions per launch against the games a wave owns (BGS_EXPERIMENT rollout_chunk) -- how
# much of a launch's instruction count is the drain of its waves.  (Round 4: 64 / 128 / 256 / 512 / 1024 / 4096 games a wave
# = 47.5 / 35.2 / 28.4 / 25.1 / 23.4 / 22.1 M instructions.)
for c in 64 128 256 512 1024 2048 4096; do
  BGS_EXPERIMENT="rollout_chunk=$c" bash tools/count_valu.sh k2c_c$c python3 tools/rollout_rate.py connect12x13 --depth 1 --reps 9 | grep rollout_lds | sed "s/^/static chunk $c: /"
done

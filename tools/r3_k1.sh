#!/bin/bash
# K1s (per-ply kernel, HBM-bound, 2^24 boards): does the power-of-two distance between the two planes matter?
for extra in 0 64 1024 4096 65536 1000000; do
  N_EXTRA=$extra python tools/k1_steps.py | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('2^24 + $extra boards: 1 ply %.1f us %.0f GB/s   4 plies %.1f us' % (d['1_ply_launch']['s_per_launch']*1e6, d['1_ply_launch']['GBps'], d['4_ply_launch']['s_per_launch']*1e6))"
done

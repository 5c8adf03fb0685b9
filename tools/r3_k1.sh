#!/bin/bash
# K1s (per-ply kernel, HBM-bound): time per ply at 2^24 and 2^20 boards
for lg in 24 24 22 20; do
  LOG2N=$lg python tools/k1_steps.py | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('2^$lg boards: 1 ply %.1f us %.0f GB/s   4 plies %.1f us' % (d['1_ply_launch']['s_per_launch']*1e6, d['1_ply_launch']['GBps'], d['4_ply_launch']['s_per_launch']*1e6))"
done
timeout -k 10 300 python -m pytest tests -x -q -m gpu -k "step_random or lockstep or lock_step or connect_parity or test_connect" 2>&1 | tail -2

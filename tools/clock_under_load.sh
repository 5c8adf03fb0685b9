#!/bin/bash
# the shader clock while the bench kernel runs (the VALU-issue peak in DESIGN assumes 2.4 GHz)
python tools/rollout_rate.py connect6x7 --depth 3 --reps 900000 > /dev/null 2>&1 &
pid=$!
sleep 9
for i in 1 2 3 4 5; do rocm-smi --showclocks 2>/dev/null | grep -i "sclk\|fclk\|mclk" | head -3; rocm-smi --showuse --showpower 2>/dev/null | grep -i "GPU use\|Graphics Package Power\|Average" | head -2; sleep 0.4; done
kill $pid 2>/dev/null; wait $pid 2>/dev/null
echo "-- idle"; sleep 2; rocm-smi --showclocks 2>/dev/null | grep -i "sclk" | head -1

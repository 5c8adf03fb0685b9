#!/bin/bash
# A/B of rollout knobs with counters: for each "NAME=VALUE,..." setting in SETTINGS (space separated), kernel stats + SQ pass
set -u
R=${GRAFT_REPO_ROOT}
cd /tmp && export TMPDIR=/tmp
i=0
for setting in ${SETTINGS}; do
  i=$((i+1))
  ( IFS=,; for kv in $setting; do export "$kv"; done
    rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_ab${i}_stats -- python3 $R/tools/rollout_rate.py connect6x7 --depth 1 --reps 12 > $R/gpurun_out/ab${i}_stats.log 2>&1 || echo fail stats $i
    rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_WAIT_INST_LDS --output-format csv -d $R/gpurun_out/prof_ab${i}_sq -- python3 $R/tools/rollout_rate.py connect6x7 --depth 1 --reps 12 > $R/gpurun_out/ab${i}_sq.log 2>&1 || echo fail sq $i
    echo "$i $setting: $(python3 $R/tools/rollout_rate.py connect6x7 --depth 3 --reps 180 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%.1f | %.1f' % (d['one_launch_at_a_time']['env_steps_per_s']/1e9, d['3_in_flight']['env_steps_per_s']/1e9))")" )
done

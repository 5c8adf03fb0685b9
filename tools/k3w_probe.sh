#!/bin/bash
# K3w (one board per wave, the last pass of the Bounce rollout) against the 8-lane tail it replaces: the default 2^18-board
# batch one launch at a time and 20 in flight; explicit plans (BGS_EXPERIMENT=bounce_plan=cap:lanes,...; lanes 64 = K3w) for tuning.
#   bash tools/k3w_probe.sh [VAR=value ...]      each argument is one more case (an environment setting)
solo() { env "$@" python3 tools/rollout_rate.py bounce --depth 1 --reps 60 --hint 1 | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('   one launch at a time %.3e' % d['one_launch_at_a_time']['env_steps_per_s'])"; }
deep() { env "$@" python3 tools/rollout_rate.py bounce --depth 20 --reps 300 --hint 20 | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('   20 in flight         %.3e' % d['20_in_flight']['env_steps_per_s'])"; }
for setting in BGS_DEFAULT=1 BGS_EXPERIMENT=bounce_wave_pass=0 "$@"; do
    echo "== $setting"
    solo $setting
    deep $setting
done

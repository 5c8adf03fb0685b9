#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
BGS_BOUNCE_PIECES=1 bash tools/profile_kernel.sh bp1 python3 tools/rollout_rate.py bounce --depth 1 --reps 6
BGS_BOUNCE_PIECES=0 bash tools/profile_kernel.sh bp0 python3 tools/rollout_rate.py bounce --depth 1 --reps 6
python3 tools/summarize_profile.py bp1 k_bounce_rollout
python3 tools/summarize_profile.py bp0 k_bounce_rollout

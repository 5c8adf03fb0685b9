#!/bin/bash
# Profiles of every kernel of the path, one command (on the GPU box):  bash tools/profile_all.sh
# Each kernel gets rocprofv3 --kernel-trace --stats, an SQ counter pass and separate FETCH_SIZE / WRITE_SIZE passes
# (tools/profile_kernel.sh: the program directly after "--"); tools/collect_profiles.py turns the CSVs into
# profiles/<round>_*.json and the counters file bench.py reads (keyed by the library's build id).
# Round 5: a pass is taken only when the kernel UNIT it describes moved since the newest file under profiles/ was written
# (tools/needs_profile.py; BGS_PROFILE_FORCE=1 takes everything): a Bounce-only edit re-takes the Bounce passes only.
set -u
cd ${GRAFT_REPO_ROOT:-$(pwd)}
need() { python3 tools/needs_profile.py stem "$1"; }
if need rollout_counters; then
# K2a: the bench itself (hand-over included), 12 timed launches in the counter passes
bash tools/profile_kernel.sh bench python3 bench.py --steps 10 --warmup 2 --prewarm-ms 0 --no-cpu-baseline --no-device-resident --no-other-configs --no-repeats
# ... and rocprofv3 --kernel-trace --stats of the bench's OWN command (200 timed steps behind the pre-warm): the mean
# duration of the bench kernel there is what the line's live kernel_ms_per_launch has to agree with (the 12-launch
# counter passes above are half ramp and tail)
( cd /tmp && export TMPDIR=/tmp && rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_benchfull_stats && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_benchfull_stats -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-other-configs --no-device-resident --no-repeats > $GRAFT_REPO_ROOT/gpurun_out/prof_benchfull_stats.log 2>&1 ) || echo "pass failed: benchfull stats"
fi
# K1s: the HBM-bound per-ply kernel at 2^24 boards
need k1 && bash tools/profile_kernel.sh k1 python3 tools/k1_steps.py
# K2c: Connect(12,13,5), 2^18 boards
need k2c && bash tools/profile_kernel.sh k2c python3 tools/rollout_rate.py connect12x13 --depth 1 --reps 9
# K3: Bounce 9x6, 2^18 boards, max_plies 4096: the piece-list kernel + tail pass (default at this size), the flat cell
# search it replaces (K3f, one launch) and lane-group mode
# (the launch shape follows the launches-in-flight hint: "bounce" = the shape of 20 in flight, which bench.py's
# other_configs runs, counted one launch at a time; "bounce_solo" = the shape of a launch that is alone)
need bounce && bash tools/profile_kernel.sh bounce python3 tools/rollout_rate.py bounce --depth 1 --reps 6 --hint 20
need bounce_solo && bash tools/profile_kernel.sh bounce_solo python3 tools/rollout_rate.py bounce --depth 1 --reps 6 --hint 1
if [ "${BGS_PROFILE_FULL:-0}" = "1" ]; then   # the kernels the defaults replaced, for the instruction-count comparisons
need bounce_k3f && BGS_EXPERIMENT="bounce_pieces=0;bounce_plan=single" bash tools/profile_kernel.sh bounce_k3f python3 tools/rollout_rate.py bounce --depth 1 --reps 6 --hint 20
need bounce_lane_groups && BGS_EXPERIMENT="bounce_group=8" bash tools/profile_kernel.sh bounce8 python3 tools/rollout_rate.py bounce --depth 1 --reps 6 --hint 20
# K2b: the register kernel K2c replaced on 12x13x5
need k2b && BGS_EXPERIMENT="rollout_no_lds=1" bash tools/profile_kernel.sh k2b python3 tools/rollout_rate.py connect12x13 --depth 1 --reps 9
fi
# K4 and the rest (reset, unpack, legal, ...): kernel stats only
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT}
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_misc_stats -- python3 $R/tools/measure_all.py > $R/gpurun_out/measure_all.json 2> $R/gpurun_out/measure_all.err
cd $R
python3 tools/valu_mix.py > gpurun_out/valu_mix.json
python3 tools/collect_profiles.py
# the hardware busy counters (VALU busy, dual issue, wave-time split) of the rollout kernels
bash tools/busy_counters.sh
python3 tools/busy_counters.py

#!/bin/bash
# rocprofv3 passes over one tool command: kernel stats, SQ counters, FETCH_SIZE, WRITE_SIZE (separate passes; the
# program directly after "--").  usage: bash tools/profile_kernel.sh <tag> python3 tools/rollout_rate.py bounce ...
# Results: gpurun_out/prof_<tag>_{stats,sq,fetch,write}/ ; summarise with tools/summarize_profile.py <tag>
set -u
# a pass that fails (rocprofv3 itself occasionally aborts at start-up) is reported and skipped, the others still run
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
# BGS_EXPERIMENT only reaches the TEST build of the library (csrc/Makefile: libbgs_test.so, same kernel objects); without it the product library is measured
if [ -n "${BGS_EXPERIMENT:-}" ]; then export BGS_LIBRARY=${BGS_LIBRARY:-$R/board-game-simulator-python_amd/libbgs_test.so}; fi
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_${tag}_stats $R/gpurun_out/prof_${tag}_sq $R/gpurun_out/prof_${tag}_fetch $R/gpurun_out/prof_${tag}_write
prog=$1; shift
args=()
for a in "$@"; do case "$a" in /*) args+=("$a");; tools/*|bench.py) args+=("$R/$a");; *) args+=("$a");; esac; done
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${tag}_stats -- $prog "${args[@]}" > $R/gpurun_out/prof_${tag}_stats.log 2>&1 || echo "pass failed: $tag stats"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $R/gpurun_out/prof_${tag}_sq -- $prog "${args[@]}" > $R/gpurun_out/prof_${tag}_sq.log 2>&1 || echo "pass failed: $tag sq"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_${tag}_fetch -- $prog "${args[@]}" > $R/gpurun_out/prof_${tag}_fetch.log 2>&1 || echo "pass failed: $tag fetch"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_${tag}_write -- $prog "${args[@]}" > $R/gpurun_out/prof_${tag}_write.log 2>&1 || echo "pass failed: $tag write"
echo "profiled $tag"

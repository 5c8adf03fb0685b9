#!/bin/bash
# VALU busy / issue counters of the rollout kernels (round 4, review item 3): one rocprofv3 --pmc pass per case with
#   SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VALU2 SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY
#   GRBM_GUI_ACTIVE GRBM_COUNT
# rocprofv3 collects counters per dispatch and serialises the dispatches, so "N launches in flight" cannot be counted
# as such: each pipelined case is counted as ONE launch of N times the boards with N times the waves per SIMD (the same
# games per wave, the same waves resident per SIMD as N launches sharing the chip).
# usage (GPU box): bash tools/busy_counters.sh ; then python3 tools/busy_counters.py -> profiles/r06_valu_busy.json
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
# BGS_EXPERIMENT only reaches the TEST build of the library (csrc/Makefile: libbgs_test.so, same kernel objects); without it the product library is measured
if [ -n "${BGS_EXPERIMENT:-}" ]; then export BGS_LIBRARY=${BGS_LIBRARY:-$R/board-game-simulator-python_amd/libbgs_test.so}; fi
cd /tmp && export TMPDIR=/tmp
PMC="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VALU2 SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE GRBM_COUNT"
pass() {  # pass <tag> <program> <args...>   (environment of the caller); skipped when the case's kernel unit did not move
  local tag=$1; shift
  if ! ( cd $R && python3 tools/needs_profile.py case $tag ); then rm -rf $R/gpurun_out/busy_$tag; return; fi
  # (a case that sets BGS_EXPERIMENT runs on the test build of the library: the product has no such switches)
  if [ -n "${BGS_EXPERIMENT:-}" ]; then export BGS_LIBRARY=$R/board-game-simulator-python_amd/libbgs_test.so; else unset BGS_LIBRARY; fi
  rm -rf $R/gpurun_out/busy_$tag
  rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $R/gpurun_out/busy_$tag -- "$@" > $R/gpurun_out/busy_$tag.log 2>&1 || echo "pass failed: $tag"
  echo "counted $tag"
}
BENCH="--steps 10 --warmup 2 --prewarm-ms 0 --no-cpu-baseline --no-device-resident --no-other-configs --no-repeats"
# K2o, one launch of 2^20 games at 2 waves per SIMD (the bench's launch, alone on the chip)
pass k2o_solo python3 $R/bench.py $BENCH
# K2o, the chip as three such launches fill it: one launch of 3 x 2^20 games at 6 waves per SIMD (512 games per wave)
BGS_ROLLOUT_WPS=6 pass k2o_3deep python3 $R/bench.py $BENCH --batch 3145728 --inflight 1
# K2c (12x13x5): one launch of 2^18 boards, and 8 launches' worth in one
pass k2c_solo python3 $R/tools/rollout_rate.py connect12x13 --depth 1 --reps 9
BGS_ROLLOUT_WPS=8 BGS_EXPERIMENT="rollout_chunk=256" pass k2c_8deep python3 $R/tools/rollout_rate.py connect12x13 --depth 1 --reps 9 --batch 2097152
# K3p (Bounce): the launch shape of 20 in flight, one launch; and 8 launches' worth of boards in one launch
pass k3p_solo python3 $R/tools/rollout_rate.py bounce --depth 1 --reps 6 --hint 20
pass k3p_8x python3 $R/tools/rollout_rate.py bounce --depth 1 --reps 6 --hint 20 --batch 2097152

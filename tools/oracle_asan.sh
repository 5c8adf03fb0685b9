#!/bin/bash
# The CPU oracle under AddressSanitizer + UndefinedBehaviorSanitizer (CPU only; GPU sanitizers are not available on
# the pool): builds oracle/liboracle_asan.so and runs the oracle's own test suites against it -- the golden fixtures of
# the reference and the host-expansion tests.  usage: bash tools/oracle_asan.sh
set -e
cd "$(dirname "$0")/.."
make -C oracle liboracle_asan.so
ASAN=$(gcc -print-file-name=libasan.so)
LD_PRELOAD=$ASAN ASAN_OPTIONS=detect_leaks=0 BGS_ORACLE_LIBRARY=$PWD/oracle/liboracle_asan.so OMP_NUM_THREADS=2 \
    python -m pytest tests/test_oracle_golden.py tests/test_host_expand.py -q -x -p no:cacheprovider
echo "oracle ASan/UBSan run: clean"

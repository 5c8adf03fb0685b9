#!/bin/bash
# A/B of two builds of libbgs.so on one box: `tools/ab_lib.sh OLD.so ROUNDS -- command...` alternates the in-tree library and
# OLD.so (BGS_LIBRARY) over the command and prints the last line of every run.
OLD=$1; N=$2; shift 3
ids() { python3 -c 'import sys; sys.path[:0] = ["board-game-simulator-python_amd"]; from simulator.game import _abi; print(_abi.unit_ids())'; }
echo "new: $(ids)"; echo "old: $(BGS_LIBRARY=$OLD ids)"
for i in $(seq $N); do
  a=$(timeout -k 10 300 "$@" 2>/dev/null | tail -1) || exit 1
  b=$(BGS_LIBRARY=$OLD timeout -k 10 300 "$@" 2>/dev/null | tail -1) || exit 1
  echo "run $i new: $a"; echo "run $i old: $b"
done

#!/bin/bash
# Round 5: BASELINE config 5's N > 1 run rehearsed at the world sizes the box lets a run go -- 2, 4 and 6 processes sharing
# the one GPU (the box allows at most 6 processes on the card; the gather's N = 8 arithmetic runs as 4 processes x 2 ranks in
# tests/test_gpu_gather_peers.py) -- over gloo + the tests' stand-in for RCCL, FULL-SIZE shards (2^20 games per rank), both
# hand-overs, each verified; then the full-size 8-rank gather (tests/gather_peer.py full, 4 processes x 2 ranks).  Writes the
# JSON lines to gpurun_out/r5_dist_*.json.  The rates measure the stand-in (and N processes on one GPU), not RCCL.
set -e
mkdir -p gpurun_out
for N in 2 4 6; do
  BGS_DIST_BACKEND=gloo BGS_RCCL_LIB=$PWD/tests/c/libfake_rccl.so OMP_NUM_THREADS=2 \
    timeout -k 10 400 python bench.py --gpus $N --steps 20 --warmup 5 --host-threads 2 2> gpurun_out/r5_dist_${N}_ranks_standin.err | grep '^{' > gpurun_out/r5_dist_${N}_ranks_standin.json
  echo "N=$N done"
done
D=$(mktemp -d)
for G in 0,1 2,3 4,5 6,7; do
  BGS_RCCL_LIB=$PWD/tests/c/libfake_rccl.so PEER_GAMES=1048576 PEER_STEPS=14 OMP_NUM_THREADS=4 \
    timeout -k 10 500 python tests/gather_peer.py $D $G 8 full > gpurun_out/r5_gather_8_ranks_full_$G.log 2>&1 &
done
wait
grep -h FULL_OK gpurun_out/r5_gather_8_ranks_full_*.log | sort > gpurun_out/r5_gather_8_ranks_full.txt || true
cat gpurun_out/r5_gather_8_ranks_full.txt
python - <<'PY'
import json
for n in (2, 4, 6):
    d = json.load(open(f"gpurun_out/r5_dist_{n}_ranks_standin.json"))
    print(n, "value %.3e" % d["value"],
          {k: ("%.3e" % d[k]["value"], d[k]["gathered_rewards_verified"], (d[k].get("gather_info") or {}).get("ranks")) for k in d if k.startswith("gather_") and "error" not in d[k]},
          {k: d[k] for k in d if k.startswith("gather_") and "error" in d[k]})
PY

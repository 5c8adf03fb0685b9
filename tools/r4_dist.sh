#!/bin/bash
# The N > 1 loops on the one-GPU box: (1) the plain one-GPU line, (2) both hand-overs over the real RCCL library with a
# one-rank world, (3) both hand-overs with 3 ranks sharing the GPU over the tests' stand-in transport.  Writes the JSON
# lines (only the lines: library banners stay in the logs) to gpurun_out/r4_dist_*.json.
set -e
mkdir -p gpurun_out
PORT=${PORT:-29517}
python bench.py --no-other-configs --no-cpu-baseline 2> gpurun_out/r4_dist_plain.err | grep '^{' > gpurun_out/r4_dist_plain.json
RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=$PORT BGS_FORCE_DIST=1 \
  python bench.py --gpus 1 --gather both 2> gpurun_out/r4_dist_one_rank.err | grep '^{' > gpurun_out/r4_dist_one_rank.json
RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=$((PORT+1)) BGS_FORCE_DIST=1 \
  python bench.py --gpus 1 --gather both --steps 20 --warmup 5 2> gpurun_out/r4_dist_one_rank_steps20.err | grep '^{' > gpurun_out/r4_dist_one_rank_steps20.json
BGS_DIST_BACKEND=gloo BGS_RCCL_LIB=$PWD/tests/c/libfake_rccl.so OMP_NUM_THREADS=4 \
  python bench.py --gpus 3 --steps 60 --warmup 5 2> gpurun_out/r4_dist_three_ranks_standin.err | grep '^{' > gpurun_out/r4_dist_three_ranks_standin.json
python - <<'PY'
import json
for name in ("plain", "one_rank", "one_rank_steps20", "three_ranks_standin"):
    d = json.load(open(f"gpurun_out/r4_dist_{name}.json"))
    print(name, "value %.3e" % d["value"], [("%.3e" % v) for v in (d.get("values_of_3") or [])],
          {k: ("%.3e" % d[k]["value"], d[k]["gathered_rewards_verified"], [("%.3e" % v) for v in d[k]["values_of_3"]]) for k in d if k.startswith("gather_") and "error" not in d[k]},
          {k: d[k] for k in d if k.startswith("gather_") and "error" in d[k]})
PY

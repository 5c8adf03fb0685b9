#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
D="RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 BGS_FORCE_DIST=1"
env $D MASTER_PORT=$((29500 + RANDOM % 400)) python bench.py --gpus 1 --gather rccl --no-cpu-baseline > gpurun_out/r03_dist_rccl.json 2> gpurun_out/r03_dist_rccl.err
env $D MASTER_PORT=$((29500 + RANDOM % 400)) BGS_GATHER_BATCH=1 python bench.py --gpus 1 --gather rccl --no-cpu-baseline > gpurun_out/r03_dist_rccl_batch1.json 2> gpurun_out/r03_dist_rccl_batch1.err
env $D MASTER_PORT=$((29500 + RANDOM % 400)) python bench.py --gpus 1 --gather shm --no-cpu-baseline > gpurun_out/r03_dist_shm.json 2> gpurun_out/r03_dist_shm.err
timeout -k 10 600 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_handover.py -x -q -m gpu -k "gather or rccl or bench" 2>&1 | tail -2
python - <<'PY'
import json
for f in ("r03_dist_shm","r03_dist_rccl","r03_dist_rccl_batch1"):
    d=json.loads([l for l in open(f"gpurun_out/{f}.json") if l.startswith("{")][-1])
    print(f, "%.3e"%d["value"], d.get("values_of_3"), d["config"].get("gathered_rewards_verified"))
PY

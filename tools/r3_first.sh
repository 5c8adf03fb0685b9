#!/bin/bash
# round 3, first GPU pass: new tests, bench modes, trace of a 20-step region
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_handover.py -x -q -m gpu > gpurun_out/r3_first_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r3_first_tests.log
tail -5 gpurun_out/r3_first_tests.log
for i in 1 2 3; do
BGS_BENCH_TRACE=1 timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > gpurun_out/r3_b20_$i.json 2> gpurun_out/r3_b20_$i.err
done
timeout -k 10 300 python bench.py --no-other-configs > gpurun_out/r3_b200.json 2> gpurun_out/r3_b200.err
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > gpurun_out/r3_b20_full.json 2> gpurun_out/r3_b20_full.err
for g in shm rccl; do
RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29511 BGS_FORCE_DIST=1 timeout -k 10 300 python bench.py --gpus 1 --gather $g --no-cpu-baseline > gpurun_out/r3_dist_$g.json 2> gpurun_out/r3_dist_$g.err
done
BGS_GATHER_DIRECT=1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29512 BGS_FORCE_DIST=1 timeout -k 10 300 python bench.py --gpus 1 --gather rccl --no-cpu-baseline > gpurun_out/r3_dist_rccl_direct.json 2> gpurun_out/r3_dist_rccl_direct.err
GPU_MAX_HW_QUEUES=32 timeout -k 10 300 python bench.py --no-other-configs --no-cpu-baseline > gpurun_out/r3_b200_q32.json 2> gpurun_out/r3_b200_q32.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r3_*.json")):
    try:
        d=json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f, "%.3e"%d["value"], d["ms_per_step"], d.get("values_of_3"), (d.get("device_resident") or {}).get("value"), d["config"].get("gather"))
    except Exception as e:
        print(f, "ERR", e)
PY
grep -h trace gpurun_out/r3_b20_*.err | head -20

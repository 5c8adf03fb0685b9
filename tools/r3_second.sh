#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_pipeline.py -x -q -m gpu > gpurun_out/r3_second_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r3_second_tests.log
tail -3 gpurun_out/r3_second_tests.log
B="--no-cpu-baseline --no-other-configs"
for i in 1 2 3; do
BGS_BENCH_TRACE=1 timeout -k 10 300 python bench.py --steps 20 --warmup 5 $B > gpurun_out/r3s_b20_$i.json 2> gpurun_out/r3s_b20_$i.err
done
BGS_BENCH_TRACE=1 timeout -k 10 300 python bench.py --steps 20 --warmup 5 --prewarm-ms 0 $B > gpurun_out/r3s_b20_nopre.json 2> gpurun_out/r3s_b20_nopre.err
timeout -k 10 300 python bench.py $B > gpurun_out/r3s_b200.json 2> gpurun_out/r3s_b200.err
for g in shm rccl; do
RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29511 BGS_FORCE_DIST=1 timeout -k 10 300 python bench.py --gpus 1 --gather $g $B > gpurun_out/r3s_dist_$g.json 2> gpurun_out/r3s_dist_$g.err
RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29513 BGS_FORCE_DIST=1 timeout -k 10 300 python bench.py --gpus 1 --gather $g --steps 20 --warmup 5 $B > gpurun_out/r3s_dist20_$g.json 2> gpurun_out/r3s_dist20_$g.err
done
BGS_GATHER_DIRECT=1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29512 BGS_FORCE_DIST=1 timeout -k 10 300 python bench.py --gpus 1 --gather rccl $B > gpurun_out/r3s_dist_rccl_direct.json 2> gpurun_out/r3s_dist_rccl_direct.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r3s_*.json")):
    try:
        d=json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f, "%.3e"%d["value"], "%.4f"%d["ms_per_step"], ["%.3e"%v for v in d.get("values_of_3") or []], "dev %.3e"%(d.get("device_resident") or {}).get("value",0), d["config"].get("gather"), d["config"]["prewarm"])
    except Exception as e:
        print(f, "ERR", e)
PY
grep -h trace gpurun_out/r3s_b20_*.err | head -20

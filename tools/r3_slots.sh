#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
B="--no-cpu-baseline --no-other-configs"
for f in 2 3 4 6; do
BGS_BENCH_SLOT_FACTOR=$f timeout -k 10 300 python bench.py --steps 20 --warmup 5 $B > gpurun_out/r3f_b20_f$f.json 2> gpurun_out/r3f_b20_f$f.err
BGS_BENCH_SLOT_FACTOR=$f timeout -k 10 300 python bench.py $B > gpurun_out/r3f_b200_f$f.json 2> gpurun_out/r3f_b200_f$f.err
done
D="RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 BGS_FORCE_DIST=1"
for f in 3 4 6; do
env $D MASTER_PORT=$((29500 + RANDOM % 400)) BGS_BENCH_SLOT_FACTOR=$f timeout -k 10 300 python bench.py --gpus 1 --gather shm $B > gpurun_out/r3f_shm_f$f.json 2> gpurun_out/r3f_shm_f$f.err
env $D MASTER_PORT=$((29500 + RANDOM % 400)) BGS_BENCH_SLOT_FACTOR=$f timeout -k 10 300 python bench.py --gpus 1 --gather shm --steps 20 --warmup 5 $B > gpurun_out/r3f_shm20_f$f.json 2> gpurun_out/r3f_shm20_f$f.err
env $D MASTER_PORT=$((29500 + RANDOM % 400)) BGS_BENCH_SLOT_FACTOR=$f timeout -k 10 300 python bench.py --gpus 1 --gather rccl $B > gpurun_out/r3f_rccl_f$f.json 2> gpurun_out/r3f_rccl_f$f.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r3f_*.json")):
    try:
        d=json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f, "%.3e"%d["value"], "%.4f"%d["ms_per_step"], ["%.3e"%v for v in d.get("values_of_3") or []], "dev %.3e"%(d.get("device_resident") or {}).get("value",0), d["config"].get("gather"))
    except Exception as e:
        print(f, "ERR", e)
PY
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r3f_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r3f_tests.log; tail -4 gpurun_out/r3f_tests.log

#!/bin/bash
# Round 3: every number DESIGN.md quotes, in one pass on the GPU box.  Counters first (profiles/r03_rollout_counters.json
# must carry this build's id before bench.py is run for the record), then the bench lines and the secondary measurements.
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
if [ -z "${SKIP_PROFILES:-}" ]; then  # (SKIP_PROFILES=1: the kernels have not changed since the last counter pass)
bash tools/profile_all.sh > gpurun_out/r3_profile_all.log 2>&1
echo "profile_all done"; tail -2 gpurun_out/r3_profile_all.log
fi
python bench.py > gpurun_out/r03_bench.json 2> gpurun_out/r03_bench.err
python bench.py --steps 20 --warmup 5 > gpurun_out/r03_bench_steps20.json 2> gpurun_out/r03_bench_steps20.err
D="RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 BGS_FORCE_DIST=1"
for g in shm rccl; do
env $D MASTER_PORT=$((29500 + RANDOM % 400)) python bench.py --gpus 1 --gather $g --no-cpu-baseline > gpurun_out/r03_dist_$g.json 2> gpurun_out/r03_dist_$g.err
done
env $D MASTER_PORT=$((29500 + RANDOM % 400)) BGS_GATHER_BATCH=6 BGS_BENCH_SLOT_FACTOR=4 python bench.py --gpus 1 --gather rccl --no-cpu-baseline > gpurun_out/r03_dist_rccl_batch6.json 2> gpurun_out/r03_dist_rccl_batch6.err
BGS_DIST_BACKEND=gloo python bench.py --gpus 3 --steps 40 --batch 262144 > gpurun_out/r03_selfstart_gloo3.json 2> gpurun_out/r03_selfstart_gloo3.err
BGS_DIST_BACKEND=gloo python bench.py --gpus 6 --steps 40 --batch 131072 --host-threads 2 > gpurun_out/r03_selfstart_gloo6.json 2> gpurun_out/r03_selfstart_gloo6.err
python tools/object_latency.py > gpurun_out/r03_object_latency.json 2>/dev/null
( echo "["; python tools/rollout_rate.py connect6x7 --depth 3 --reps 90 2>/dev/null; echo ","; python tools/rollout_rate.py connect12x13 --depth 8 --reps 160 2>/dev/null; echo ","; python tools/rollout_rate.py bounce --depth 20 --reps 160 2>/dev/null; echo ","; BGS_BOUNCE_PIECES=0 BGS_BOUNCE_PLAN=single python tools/rollout_rate.py bounce --depth 20 --reps 160 2>/dev/null; echo "]" ) > gpurun_out/r03_rollout_rates.json
python - <<'PY'
import json
for f in ("r03_bench","r03_bench_steps20","r03_dist_shm","r03_dist_rccl","r03_dist_rccl_batch6","r03_selfstart_gloo3","r03_selfstart_gloo6"):
    try:
        d=json.loads([l for l in open(f"gpurun_out/{f}.json") if l.startswith("{")][-1])
        print(f, "%.3e"%d["value"], d["ms_per_step"], d.get("values_of_3"), d["roofline"].get("frac"), d["config"].get("gather"), d["config"].get("gathered_rewards_verified"))
    except Exception as e: print(f, "ERR", e)
PY

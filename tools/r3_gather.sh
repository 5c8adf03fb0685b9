#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
B="--no-cpu-baseline --no-other-configs --no-repeats"
D="RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 BGS_FORCE_DIST=1"
run() { # name, env...
  name=$1; shift
  env $D MASTER_PORT=$((29500 + RANDOM % 400)) "$@" timeout -k 10 300 python bench.py --gpus 1 --gather rccl $B > gpurun_out/r3g_$name.json 2> gpurun_out/r3g_$name.err
}
run direct_b1 BGS_GATHER_DIRECT=1 BGS_GATHER_BATCH=1
run direct_b3 BGS_GATHER_DIRECT=1 BGS_GATHER_BATCH=3
run direct_b6 BGS_GATHER_DIRECT=1 BGS_GATHER_BATCH=6 BGS_BENCH_SLOT_FACTOR=4
run copy_b3 BGS_GATHER_DIRECT=0 BGS_GATHER_BATCH=3
run direct_b3_ch1 BGS_GATHER_DIRECT=1 BGS_GATHER_BATCH=3 NCCL_MAX_NCHANNELS=1 NCCL_MIN_NCHANNELS=1
run direct_b1_ch1 BGS_GATHER_DIRECT=1 BGS_GATHER_BATCH=1 NCCL_MAX_NCHANNELS=1 NCCL_MIN_NCHANNELS=1
env $D MASTER_PORT=29911 timeout -k 10 300 python bench.py --gpus 1 --gather shm $B > gpurun_out/r3g_shm.json 2> gpurun_out/r3g_shm.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r3g_*.json")):
    try:
        d=json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f, "%.3e"%d["value"], "%.4f"%d["ms_per_step"], "dev %.3e"%(d.get("device_resident") or {}).get("value",0), d["config"].get("gather"), d["config"]["gathered_rewards_verified"])
    except Exception as e:
        print(f, "ERR", e)
PY
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29933 BGS_FORCE_DIST=1 BGS_GATHER_BATCH=1
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_gather -- python3 $R/bench.py --gpus 1 --gather rccl --steps 60 --warmup 3 --prewarm-ms 0 --no-cpu-baseline --no-other-configs --no-repeats --no-device-resident > $R/gpurun_out/r3g_prof.log 2>&1
find $R/gpurun_out/prof_gather -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'cut -c1-200 {} | head -12'

#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
for blk in 256 512 1024; do
BGS_BOUNCE_BLOCK=$blk timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "bounce_piece_list or bounce_flat_rollout_shared or bounce_rollout_kernel" > gpurun_out/r3b2_tests_$blk.log 2>&1
echo "tests blk=$blk rc=$?"; tail -2 gpurun_out/r3b2_tests_$blk.log
done
for blk in 256 512 1024; do
for wv in 0 1024 512; do
BGS_BOUNCE_BLOCK=$blk BGS_BOUNCE_FLAT_WAVES=$wv timeout -k 10 300 python tools/rollout_rate.py bounce --depth 16 --reps 48 > gpurun_out/r3b2_rate_b${blk}_w$wv.json 2> gpurun_out/r3b2_rate.err
done
done
BGS_BOUNCE_PIECES=0 timeout -k 10 300 python tools/rollout_rate.py bounce --depth 16 --reps 48 > gpurun_out/r3b2_rate_k3f.json 2>> gpurun_out/r3b2_rate.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r3b2_*.json")):
    try:
        d=json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f, {k:("%.3e"%v["env_steps_per_s"], "%.3f ms"%(v["s_per_batch"]*1e3)) for k,v in d.items() if isinstance(v,dict) and "env_steps_per_s" in v})
    except Exception as e:
        print(f, "ERR", e)
PY

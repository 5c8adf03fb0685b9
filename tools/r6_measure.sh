#!/bin/bash
# Round 6: the lines and rehearsals behind profiles/r06_*.json, one gpurun call:  bash tools/r6_measure.sh
#   bench at 200 and at the driver's 20 steps, the strict RNG contract as `value`, BASELINE config 5 rehearsed at 2 / 4 / 6 processes
#   sharing the GPU over the tests' stand-in for RCCL (the RCCL gather carries `value` since this round), the API levels.
set -u
mkdir -p gpurun_out
python bench.py > gpurun_out/r6_bench.json 2> gpurun_out/r6_bench.err; echo "bench 200: rc $?"
python bench.py --steps 20 --warmup 5 > gpurun_out/r6_bench_steps20.json 2> gpurun_out/r6_bench_steps20.err; echo "bench 20: rc $?"
python bench.py --rng per-ply --no-other-configs > gpurun_out/r6_bench_per_ply.json 2> gpurun_out/r6_bench_per_ply.err; echo "bench per-ply: rc $?"
for N in 2 4 6; do
  BGS_DIST_BACKEND=gloo BGS_RCCL_LIB=$PWD/tests/c/libfake_rccl.so OMP_NUM_THREADS=2 \
    timeout -k 10 400 python bench.py --gpus $N --steps 20 --warmup 5 --host-threads 2 2> gpurun_out/r6_dist_${N}_ranks_standin.err | grep '^{' > gpurun_out/r6_dist_${N}_ranks_standin.json
  echo "N=$N rc ${PIPESTATUS[0]}"
done
timeout -k 10 600 python tools/api_rates.py > gpurun_out/r6_api_rates.json 2> gpurun_out/r6_api_rates.err; echo "api rates: rc $?"
python - <<'PY'
import json
for name in ("r6_bench", "r6_bench_steps20", "r6_bench_per_ply"):
    d = json.loads(open(f"gpurun_out/{name}.json").read().strip().splitlines()[-1])
    print(name, "value %.4g" % d["value"], "median %.4g" % (d["value_median_of_3"] or 0), "frac", d["roofline"].get("frac"), "rng_other %.4g" % d["rng_other"]["value"])
    for k, v in (d.get("other_configs") or {}).items():
        print("   ", k, "%.4g" % v["value"], "solo %.4g" % v["solo"]["value"], v["parity_with_oracle"], v["valu_issue"].get("frac"), v.get("eight_in_flight"))
for n in (2, 4, 6):
    d = json.loads(open(f"gpurun_out/r6_dist_{n}_ranks_standin.json").read().strip().splitlines()[-1])
    print(n, "value %.3e" % d["value"], d["config"]["gather"], d["rccl_ranks"], d.get("failed_handovers"))
PY

#!/bin/bash
# K2c: where do the cycles go?  LDS and wait counters, one launch at a time
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/k2c_lds_a $R/gpurun_out/k2c_lds_b
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d $R/gpurun_out/k2c_lds_a -- python3 $R/tools/rollout_rate.py connect12x13 --depth 1 --reps 6 > $R/gpurun_out/k2c_lds_a.log 2>&1 || echo "pass a failed"
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY --output-format csv -d $R/gpurun_out/k2c_lds_b -- python3 $R/tools/rollout_rate.py connect12x13 --depth 1 --reps 6 > $R/gpurun_out/k2c_lds_b.log 2>&1 || echo "pass b failed"
cd $R
python3 - <<'PY'
import csv, glob, collections
for tag in ("a", "b"):
    for f in glob.glob(f"gpurun_out/k2c_lds_{tag}/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            if "_lds" in r["Kernel_Name"]:
                acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, c in acc.items():
            print(k)
            for name, v in sorted(c.items()):
                print(f"   {name:24s} {sum(v)/len(v):16.0f}")
PY

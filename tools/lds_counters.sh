#!/bin/bash
# LDS-side SQ counters of one command (rocprofv3 --pmc, the program directly after "--"): is a kernel's LDS pipe, its bank
# conflicts or its atomics what it waits for?   bash tools/lds_counters.sh TAG -- program args...
TAG=$1; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/lds_$TAG
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_LDS_ATOMIC_RETURN SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/lds_$TAG -- "$@" > $R/gpurun_out/lds_$TAG.log 2>&1 || echo "pass failed: $TAG"
python3 - $R/gpurun_out/lds_$TAG $TAG <<'PY'
import csv, glob, sys, collections, re
d, tag = sys.argv[1:3]
def short(name):
    m = re.search(r"(k_[a-z0-9_]+)(<[^(]*>)?\(", name)
    return (m.group(1) + (m.group(2) or ""))[-60:] if m else name[:60]
rows = collections.defaultdict(lambda: collections.defaultdict(list)); dur = collections.defaultdict(list)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, c in rows.items():
    if max(dur.get(k, [0])) < 20: continue
    print(tag, k, "n", len(dur[k]), "us %.1f" % (sum(dur[k]) / len(dur[k])), {n: "%.4g" % (sum(v) / len(v)) for n, v in sorted(c.items())})
PY

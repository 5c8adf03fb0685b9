#!/bin/bash
# SQ counters of one command under several builds of the library: `tools/ab_counters.sh TAG LIB.so -- program args...`
# (rocprofv3 --pmc, the program directly after "--"); prints per-kernel means of duration, VALU instructions and cycles.
TAG=$1; LIB=$2; shift 3
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/abc_$TAG
BGS_LIBRARY=$LIB rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU2 SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU --output-format csv -d $R/gpurun_out/abc_$TAG -- "$@" > $R/gpurun_out/abc_$TAG.log 2>&1 || echo "pass failed: $TAG"
python3 - $R/gpurun_out/abc_$TAG $TAG <<'PY'
import csv, glob, sys, collections
d, tag = sys.argv[1:3]
def short(name):   # "void bgs::(anonymous namespace)::k_x<...>(args)" -> "k_x<...>"
    import re
    m = re.search(r"(k_[a-z0-9_]+)(<[^(]*>)?\(", name)
    return (m.group(1) + (m.group(2) or ""))[-60:] if m else name[:60]
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = collections.defaultdict(list)
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, c in rows.items():
    if max(dur.get(k, [0])) < 20: continue
    print(tag, k, "durations", [round(x) for x in dur[k]])
    print(tag, k, "n", len(dur[k]), "us %.1f" % (sum(dur[k]) / len(dur[k])), {n: "%.4g" % (sum(v) / len(v)) for n, v in sorted(c.items())})
PY

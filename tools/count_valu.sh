#!/bin/bash
# Per-kernel VALU wave-instructions, active lanes and duration of one command (one rocprofv3 --pmc pass).
# usage (GPU box): bash tools/count_valu.sh <tag> <program> <args...>     -> gpurun_out/valu_<tag>.txt
set -u
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
# BGS_EXPERIMENT only reaches the TEST build of the library (csrc/Makefile: libbgs_test.so, same kernel objects); without it the product library is measured
if [ -n "${BGS_EXPERIMENT:-}" ]; then export BGS_LIBRARY=${BGS_LIBRARY:-$R/board-game-simulator-python_amd/libbgs_test.so}; fi
here=$(pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/valu_$tag
args=()
for a in "$@"; do case "$a" in tools/*|bench.py) args+=("$R/$a");; *) args+=("$a");; esac; done
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAVES --output-format csv -d $R/gpurun_out/valu_$tag -- "${args[@]}" > $R/gpurun_out/valu_$tag.log 2>&1 || echo "pass failed: $tag"
cd $here
python3 - "$R" "$tag" <<'PY' | tee $R/gpurun_out/valu_$tag.txt
import csv, glob, os, sys
from collections import defaultdict
root, tag = sys.argv[1], sys.argv[2]
acc = defaultdict(lambda: defaultdict(list)); dur = defaultdict(list)
for f in glob.glob(os.path.join(root, "gpurun_out", f"valu_{tag}", "**", "*counter_collection.csv"), recursive=True):
    seen = set()
    for row in csv.DictReader(open(f)):
        import re as _re; _m = _re.search(r"k_[a-z0-9_]+(<[^>]*>)?", row["Kernel_Name"]); k = _m.group(0) if _m else row["Kernel_Name"][:60]
        acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
        key = (k, row.get("Dispatch_Id"))
        if key not in seen and row.get("Start_Timestamp"):
            seen.add(key); dur[k].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
for k, cs in sorted(acc.items(), key=lambda kv: -sum(kv[1].get("SQ_INSTS_VALU", [0]))):
    n = len(cs["SQ_INSTS_VALU"]); valu = sum(cs["SQ_INSTS_VALU"]) / n
    if valu < 1e4: continue
    lanes = sum(cs["SQ_THREAD_CYCLES_VALU"]) / max(sum(cs["SQ_ACTIVE_INST_VALU"]), 1)
    print(f"{tag} {k}: dispatches {n} VALU/launch {valu:.4g} SALU/launch {sum(cs['SQ_INSTS_SALU'])/n:.4g} lanes {lanes:.1f} mean_us {sum(dur[k])/max(len(dur[k]),1)/1e3:.1f}")
PY

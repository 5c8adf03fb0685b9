#!/bin/bash
# Per-kernel time of one tool command (rocprofv3 --kernel-trace --stats, csv): the ten largest kernels by total time.
# usage: bash tools/kernel_stats.sh <tag> python3 tools/<tool>.py args...      -> gpurun_out/prof_<tag>/ + a table on stdout
set -u
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_${tag}
prog=$1; shift
args=()
for a in "$@"; do case "$a" in tools/*|bench.py) args+=("$R/$a");; *) args+=("$a");; esac; done
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${tag} -- $prog "${args[@]}" > $R/gpurun_out/prof_${tag}.log 2>&1 || echo "pass failed"
f=$(find $R/gpurun_out/prof_${tag} -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:12]:
    print(f'{r["Name"][:100]:100s} calls {r["Calls"]:>6s}  total {float(r["TotalDurationNs"]) / 1e6:9.3f} ms  mean {float(r["AverageNs"]) / 1e3:9.2f} us')
PY

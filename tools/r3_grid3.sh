#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
run() {
python - <<'PY'
import sys, os, json
sys.path[:0]=[os.getcwd(), os.path.join(os.getcwd(),"board-game-simulator-python_amd")]
import bench
r=bench.grids_to_host(40)
print({k:v for k,v in os.environ.items() if k.startswith("BGS_")}, "%.3e"%r["value"], "%.3f ms"%r["ms_per_step"], "%.1f GB/s"%r["pcie_GBps"], r["host_grids_equal_device_grids"])
PY
}
BGS_GRID_THREADS=12 BGS_GRID_NO_EXPAND=1 BGS_GRID_DIRECT=1 run
BGS_GRID_THREADS=12 BGS_GRID_DIRECT=1 run
BGS_GRID_THREADS=12 BGS_GRID_DIRECT=1 run
BGS_GRID_THREADS=12 run

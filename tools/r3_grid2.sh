#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
for t in 6 8 12 16; do
BGS_GRID_THREADS=$t python - <<'PY'
import sys, os, json
sys.path[:0]=[os.getcwd(), os.path.join(os.getcwd(),"board-game-simulator-python_amd")]
import bench
r=bench.grids_to_host(40)
print(os.environ["BGS_GRID_THREADS"], "%.3e"%r["value"], "%.3f ms"%r["ms_per_step"], "%.1f GB/s"%r["pcie_GBps"], r["host_grids_equal_device_grids"])
PY
done
timeout -k 10 600 python -m pytest tests/test_gpu_pipeline.py -x -q -m gpu -k grid 2>&1 | tail -2

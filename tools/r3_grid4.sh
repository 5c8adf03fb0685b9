#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
run() {
python - <<'PY'
import sys, os, ctypes
sys.path[:0]=[os.getcwd(), os.path.join(os.getcwd(),"board-game-simulator-python_amd")]
from simulator.game import _abi
_abi._more_hardware_queues()
if os.environ.get("BIND"):
    got = ctypes.c_int(0); _abi.check(_abi.lib().bgs_bind_host_thread(0, ctypes.byref(got)))
import bench
vals=[]
for i in range(4):
    r=bench.grids_to_host(40); vals.append(r["value"])
print("BIND" if os.environ.get("BIND") else "free", ["%.3e"%v for v in vals])
PY
}
run; BIND=1 run; run; BIND=1 run
numactl -H 2>/dev/null | head -5; cat /sys/bus/pci/devices/*/numa_node 2>/dev/null | sort | uniq -c | head

#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_dropin_connect.py tests/test_dropin_bounce.py tests/test_gpu_textual.py -x -q -m gpu > gpurun_out/r3o_tests.log 2>&1
echo "tests rc=$?"; tail -3 gpurun_out/r3o_tests.log
timeout -k 10 300 python tools/object_latency.py > gpurun_out/r3_object_latency.json 2> gpurun_out/r3o.err; cat gpurun_out/r3_object_latency.json

set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/k1_stats -- python3 $R/tools/k1_steps.py > $R/gpurun_out/k1_stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/k1_fetch -- python3 $R/tools/k1_steps.py > $R/gpurun_out/k1_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/k1_write -- python3 $R/tools/k1_steps.py > $R/gpurun_out/k1_write.log 2>&1

set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/p4_stats -- python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/p4_stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/p4_fetch -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $R/gpurun_out/p4_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/p4_write -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $R/gpurun_out/p4_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/p4_sq -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $R/gpurun_out/p4_sq.log 2>&1
cd $R
python bench.py > gpurun_out/bench_final.log 2>&1
tail -1 gpurun_out/bench_final.log | cut -c1-200

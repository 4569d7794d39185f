bash tools/gpu_run.sh r06_q1_sf100 prof pmc:FETCH_SIZE pmc:WRITE_SIZE
mkdir -p gpurun_out
timeout 500 ./databend_amd/host/block_sweep --out gpurun_out/r06_block_size_sweep.json 2> gpurun_out/r06_block_size_sweep.err; tail -3 gpurun_out/r06_block_size_sweep.err
export TMPDIR=/tmp; R=$PWD
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_sweep -o p -- $R/databend_amd/host/block_sweep --only-q1 --rows 33554432 > /dev/null 2> $R/gpurun_out/sweep_prof.err)
f=$(find gpurun_out/prof_sweep -name "*kernel_stats.csv" | head -1); cp $f gpurun_out/r06_block_sweep_kernel_stats.csv
python tools/trace_summary.py $(find gpurun_out/prof_sweep -name "*kernel_trace.csv" | head -1) 24 > gpurun_out/r06_block_sweep_trace_summary.txt; cat gpurun_out/r06_block_sweep_trace_summary.txt
rm -rf gpurun_out/prof_sweep
timeout 600 python -m pytest tests/test_gpu_parquet_device.py tests/test_gpu_comm.py tests/test_host_cpp.py -m gpu -q -x --timeout 300 2>&1 | tail -3

mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parquet_device.py tests/test_gpu_parquet.py -m gpu -q -x --timeout 300 2>&1 | grep -E "passed|failed|error" | tail -3
DBHIP_FAGG_JIT=0 timeout 300 python tools/probes/interp_sf100.py > gpurun_out/r06_q1_sf100_interpreter.json 2> gpurun_out/interp.err; tail -c 600 gpurun_out/r06_q1_sf100_interpreter.json
bash tools/gpu_run.sh r06_microbench_groupby "mb:--only groupby,gblayouts,sort,join"
cp gpurun_out/mb_r06_microbench_groupby.json gpurun_out/r06_microbench_groupby.json

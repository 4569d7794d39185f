# List chunks on the device after the level streams moved to LDS: the scan-side GPU tests, the 6 M-row List<Int64> chunk per codec, its kernel stats.   bash tools/probes/r06_list_step.sh <tag>
TAG=${1:-lst}; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parquet.py tests/test_gpu_parquet_device.py -m gpu -x -q > gpurun_out/pytest_$TAG.log 2>&1; tail -3 gpurun_out/pytest_$TAG.log
python tools/probes/pq_list_rate.py > gpurun_out/${TAG}_pq_list_rate.json 2> gpurun_out/${TAG}_pq_list_rate.err; cut -c1-900 gpurun_out/${TAG}_pq_list_rate.json
bash tools/gpu_run.sh $TAG "pyprof:tools/probes/pq_list_rate.py --codecs none" | tail -8

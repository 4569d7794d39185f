# One GPU step of the ZSTD work: the scan-side GPU tests, then the shipped build / producer-alone pair, then the per-wave cycle counters
# of the experiments build (DBHIP_PQ_ZSTD_X=32).   bash tools/probes/r06_zstd_step.sh <tag>
TAG=${1:-zstep}; R=$PWD; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parquet.py tests/test_gpu_parquet_device.py -m gpu -x -q > gpurun_out/pytest_$TAG.log 2>&1; tail -3 gpurun_out/pytest_$TAG.log
bash tools/probes/r06_zstd_ab.sh $TAG
DBHIP_LIBRARY=$R/databend_amd/libdbhip_exp.so DBHIP_PQ_ZSTD_X=32 timeout 300 python tools/pq_scan_probe.py --codec zstd --reps 1 2>&1 | grep "zstd2 page" | sort | head -8

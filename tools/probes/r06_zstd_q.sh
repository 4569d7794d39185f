# ZSTD: shipped build (queue of 64 commands, 4 KiB ring: 8 pages per CU) against the experiments build with a queue of 256 (EXP_DEFS=-DDBHIP_ZQ_CAP=256) at
# 4 / 8 KiB rings, after the scan-side GPU tests.   bash tools/probes/r06_zstd_q.sh <tag>
TAG=${1:-zq}; R=$PWD; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parquet.py tests/test_gpu_parquet_device.py -m gpu -x -q > gpurun_out/pytest_$TAG.log 2>&1; tail -3 gpurun_out/pytest_$TAG.log
run() { n=$1; shift; env "$@" python tools/pq_scan_probe.py --codec zstd --reps 7 > gpurun_out/${TAG}_$n.json 2>&1; echo $n $(tail -1 gpurun_out/${TAG}_$n.json | cut -c230-320); }
run ship A=1
run q256_r4096 DBHIP_LIBRARY=$R/databend_amd/libdbhip_exp.so DBHIP_PQ_ZSTD_RING=4096
run q256_r8192 DBHIP_LIBRARY=$R/databend_amd/libdbhip_exp.so DBHIP_PQ_ZSTD_RING=8192
run ship_again A=1

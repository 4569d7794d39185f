# LZ4 / Snappy ring size sweep in the experiments build, after the scan-side GPU tests on the shipped build.   bash tools/probes/r06_lz_ring.sh <tag>
TAG=${1:-lzring}; R=$PWD; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parquet.py tests/test_gpu_parquet_device.py -m gpu -x -q > gpurun_out/pytest_$TAG.log 2>&1; tail -2 gpurun_out/pytest_$TAG.log
export DBHIP_LIBRARY=$R/databend_amd/libdbhip_exp.so
for c in lz4 snappy; do for r in 8192 4096 2048; do
  DBHIP_PQ_LZ_RING=$r python tools/pq_scan_probe.py --codec $c --reps 5 > gpurun_out/${TAG}_${c}_$r.json 2>&1; echo $c ring $r $(tail -1 gpurun_out/${TAG}_${c}_$r.json | cut -c230-300)
done; done

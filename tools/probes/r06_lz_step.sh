# One GPU step of the byte-parallel replay for LZ4 / Snappy: the scan-side GPU tests, then the 7-column set under every codec.   bash tools/probes/r06_lz_step.sh <tag>
TAG=${1:-lzstep}; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parquet.py tests/test_gpu_parquet_device.py -m gpu -x -q > gpurun_out/pytest_$TAG.log 2>&1; tail -3 gpurun_out/pytest_$TAG.log
for c in none lz4 snappy zstd; do
  timeout 300 python tools/pq_scan_probe.py --codec $c --reps 5 > gpurun_out/${TAG}_$c.json 2>&1; echo $c $(tail -1 gpurun_out/${TAG}_$c.json | cut -c230-330)
done

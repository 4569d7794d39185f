# ZSTD ring size sweep in the experiments build (pages per CU by LDS: 8 KiB ring 6, 4 KiB 7).   bash tools/probes/r06_zstd_ring.sh <tag>
TAG=${1:-zring}; R=$PWD; mkdir -p gpurun_out
export DBHIP_LIBRARY=$R/databend_amd/libdbhip_exp.so
for r in 8192 4096 2048 16384; do
  DBHIP_PQ_ZSTD_RING=$r python tools/pq_scan_probe.py --codec zstd --reps 5 > gpurun_out/${TAG}_$r.json 2>&1; echo ring $r $(tail -1 gpurun_out/${TAG}_$r.json | cut -c230-300)
done

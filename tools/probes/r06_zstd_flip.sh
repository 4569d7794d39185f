TAG=${1:-zflip}; R=$PWD; mkdir -p gpurun_out
run() { n=$1; shift; env "$@" python tools/pq_scan_probe.py --codec zstd --reps 5 $EXTRA > gpurun_out/${TAG}_$n.json 2>&1; echo $n $(tail -1 gpurun_out/${TAG}_$n.json | cut -c230-300); }
export DBHIP_LIBRARY=$R/databend_amd/libdbhip_exp.so
run x0 DBHIP_PQ_ZSTD_X=0
run x2 DBHIP_PQ_ZSTD_X=2
run x4 DBHIP_PQ_ZSTD_X=4
run x8 DBHIP_PQ_ZSTD_X=8
run x16 DBHIP_PQ_ZSTD_X=16
EXTRA=--no-check
run x1 DBHIP_PQ_ZSTD_X=1
run x9 DBHIP_PQ_ZSTD_X=9

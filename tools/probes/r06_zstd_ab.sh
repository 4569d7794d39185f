# ZSTD two-wave kernel: the shipped build, the experiments build with the consumer switched off (DBHIP_PQ_ZSTD_X=1: producer alone,
# wrong output on purpose), and (with "pmc") one SQ counter pass + kernel stats of the shipped build.   bash tools/probes/r06_zstd_ab.sh <tag> [pmc]
TAG=${1:-zab}; R=$PWD; mkdir -p gpurun_out
run() { n=$1; shift; env "$@" python tools/pq_scan_probe.py --codec zstd --reps 5 $EXTRA > gpurun_out/${TAG}_$n.json 2>&1; tail -1 gpurun_out/${TAG}_$n.json | cut -c230-330; }
run ship A=1
EXTRA=--no-check run producer_alone DBHIP_LIBRARY=$R/databend_amd/libdbhip_exp.so DBHIP_PQ_ZSTD_X=1
if [ "$2" = pmc ]; then
bash tools/gpu_run.sh $TAG "pypmc:SQ_WAVE_CYCLES,SQ_WAIT_ANY,SQ_INSTS_SALU,SQ_WAIT_INST_ANY,SQ_INSTS_VALU,SQ_INSTS_LDS:tools/pq_scan_probe.py --codec zstd --reps 3" | grep zstd2
bash tools/gpu_run.sh $TAG "pyprof:tools/pq_scan_probe.py --codec zstd --reps 3" | grep zstd2
fi

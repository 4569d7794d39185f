#!/bin/bash
# HBM traffic counters for the Q1 kernel: separate --pmc passes (FETCH_SIZE, WRITE_SIZE), kernel-trace only.
TAG=${1:-run}
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$PWD; cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${TAG}_$C -o q1 -- python $R/bench.py --steps 3 --warmup 1 --no-cpu > $R/gpurun_out/pmc_${TAG}_$C.json 2> $R/gpurun_out/pmc_${TAG}_$C.err
  tail -1 $R/gpurun_out/pmc_${TAG}_$C.err
done
cd $R; ls gpurun_out/pmc_${TAG}_FETCH_SIZE/; grep -h q1_fused gpurun_out/pmc_${TAG}_*/q1_counter_collection.csv | head -8

#!/bin/bash
# round 2, run Z: bench line (Q1 SF100 + operator plans + Q3 SF100 + ANN + CPU baseline), rocprofv3 kernel stats of Q1 and of Q3 SF100
TAG=${1:-r02z}
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$PWD
timeout 1200 python bench.py > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err; tail -3 gpurun_out/bench_$TAG.err; cut -c1-3000 gpurun_out/bench_$TAG.json
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_q3_$TAG -o q3 -- python $R/tools/bench_q3.py --sf 100 --reps 3 --out $R/gpurun_out/q3_$TAG.json > $R/gpurun_out/prof_q3_$TAG.log 2>&1
cd $R
f=$(find gpurun_out/prof_q3_$TAG -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" gpurun_out/${TAG}_q3_sf100_kernel_stats.csv && head -12 "$f" | cut -c1-150
rm -rf gpurun_out/prof_q3_$TAG
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_op_$TAG -o op -- python $R/bench.py --steps 3 --warmup 1 --no-cpu --no-ann --no-q3 > $R/gpurun_out/prof_op_$TAG.json 2> $R/gpurun_out/prof_op_$TAG.err
cd $R
f=$(find gpurun_out/prof_op_$TAG -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" gpurun_out/${TAG}_q1_sf100_operator_plans_kernel_stats.csv && head -14 "$f" | cut -c1-150
rm -rf gpurun_out/prof_op_$TAG

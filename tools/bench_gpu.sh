#!/bin/bash
# usage (on the GPU box, repo root): bash tools_bench_gpu.sh <tag>  -> tests + bench + rocprof csv into gpurun_out/
TAG=${1:-run}
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -8 > gpurun_out/pytest_$TAG.log; cat gpurun_out/pytest_$TAG.log
timeout 600 python bench.py --no-cpu > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err; tail -2 gpurun_out/bench_$TAG.err; cat gpurun_out/bench_$TAG.json
R=$PWD; cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -o q1 -- python $R/bench.py --steps 10 --warmup 2 --no-cpu > $R/gpurun_out/prof_bench_$TAG.json 2> $R/gpurun_out/prof_bench_$TAG.err
cd $R; find gpurun_out/prof_$TAG -name "*kernel_stats*" | head -1 | xargs -r head -12

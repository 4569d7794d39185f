#!/bin/bash
# round 2, run B: whole GPU suite, SF100 bench line, rocprofv3 kernel stats + PMC traffic passes of the same command
TAG=${1:-r02b}
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | tail -8 > gpurun_out/pytest_$TAG.log; cat gpurun_out/pytest_$TAG.log
timeout 900 python bench.py > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err; tail -3 gpurun_out/bench_$TAG.err; cut -c1-2600 gpurun_out/bench_$TAG.json
R=$PWD; cd /tmp
Q="--steps 10 --warmup 2 --no-cpu --no-ann --no-q3 --no-opplan"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -o q1 -- python $R/bench.py $Q > $R/gpurun_out/prof_bench_$TAG.json 2> $R/gpurun_out/prof_bench_$TAG.err
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${TAG}_$C -o q1 -- python $R/bench.py --steps 3 --warmup 1 --no-cpu --no-ann --no-q3 --no-opplan > $R/gpurun_out/pmc_${TAG}_$C.json 2> $R/gpurun_out/pmc_${TAG}_$C.err
done
cd $R; find gpurun_out/prof_$TAG -name "*kernel_stats*" | head -1 | xargs -r head -8
grep -h q1_fused gpurun_out/pmc_${TAG}_*/*/*counter_collection.csv gpurun_out/pmc_${TAG}_*/*counter_collection.csv 2>/dev/null | head -8
for X in alltoall allgather; do
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --backend gloo --share-gpu --exchange $X --sf 10 --steps 5 --warmup 1 --no-ann > gpurun_out/bench_${TAG}_2rank_$X.json 2> gpurun_out/bench_${TAG}_2rank_$X.err; tail -2 gpurun_out/bench_${TAG}_2rank_$X.err; cut -c1-500 gpurun_out/bench_${TAG}_2rank_$X.json
done

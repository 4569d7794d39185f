#!/bin/bash
# round 2, run A: new group-by tests + the group-by / Q1 parity tests, the SF100 bench line, the 2-rank functional check
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_groupby2.py -q -x 2>&1 | tail -15 > gpurun_out/pytest_r02a_new.log; cat gpurun_out/pytest_r02a_new.log
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "groupby or q1 or state_block" 2>&1 | tail -8 > gpurun_out/pytest_r02a_gb.log; cat gpurun_out/pytest_r02a_gb.log
timeout 900 python bench.py > gpurun_out/bench_r02a.json 2> gpurun_out/bench_r02a.err; tail -5 gpurun_out/bench_r02a.err; cut -c1-3000 gpurun_out/bench_r02a.json
for X in alltoall allgather; do
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --backend gloo --share-gpu --exchange $X --sf 10 --steps 5 --warmup 1 --no-ann > gpurun_out/bench_r02a_2rank_$X.json 2> gpurun_out/bench_r02a_2rank_$X.err; tail -3 gpurun_out/bench_r02a_2rank_$X.err; cut -c1-600 gpurun_out/bench_r02a_2rank_$X.json
done

#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
DBHIP_TRACE=1 timeout 900 python -m pytest tests/test_gpu_fused.py -q -x 2>&1 | grep -v "^\[dbhip\] groupby" | tail -8
echo "== sync"; DBHIP_FAGG_JIT=sync timeout 900 python -m pytest tests/test_gpu_fused.py tests/test_gpu_groupby2.py -q -x 2>&1 | tail -3
echo "== fagg auto"; DBHIP_FAGG_AUTO=1 DBHIP_FAGG_JIT=sync timeout 200 python tools/microbench.py --only groupby --gb-card 4,8 2>&1 | grep name | cut -c1-140
timeout 600 python bench.py --steps 5 --warmup 1 --no-cpu --no-q3 --no-ann > gpurun_out/bench_r02j3.json 2> gpurun_out/bench_r02j3.err; tail -2 gpurun_out/bench_r02j3.err; python -c "
import json; d=json.load(open('gpurun_out/bench_r02j3.json')); print(json.dumps(d['q1_operator_plan'])[:1800])"

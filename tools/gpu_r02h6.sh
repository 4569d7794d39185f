#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python tools/bench_hnsw.py --rows 1000000 --queries 10000 --cpu 300 --out gpurun_out/r02h_hnsw_1m_iid.json 2>&1 | tail -1 | cut -c1-1200
timeout 900 python bench.py --steps 5 --warmup 1 --no-cpu --no-q3 --no-opplan > gpurun_out/bench_r02h6.json 2> gpurun_out/bench_r02h6.err; tail -2 gpurun_out/bench_r02h6.err; python -c "
import json; d=json.load(open('gpurun_out/bench_r02h6.json')); print(json.dumps(d['ann'])[:1500])"

#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python tools/bench_hnsw.py --rows 100000 --queries 10000 --out gpurun_out/r02h_hnsw_100k.json 2>&1 | tail -2 | cut -c1-700
timeout 900 python tools/bench_hnsw.py --rows 1000000 --queries 10000 --out gpurun_out/r02h_hnsw_1m.json 2>&1 | tail -2 | cut -c1-700

#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python tools/bench_hnsw.py --rows 1000000 --dim 128 --queries 10000 --clusters 1000 --metric l2 2>&1 | tail -1 | cut -c1-900
timeout 600 python tools/bench_hnsw.py --rows 1000000 --dim 128 --queries 10000 --clusters 1000 --normalize 2>&1 | tail -1 | cut -c1-900
timeout 600 python tools/bench_hnsw.py --rows 1000000 --dim 768 --queries 10000 --normalize 2>&1 | tail -1 | cut -c1-900

#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python tools/bench_hnsw.py --rows 1000000 --queries 10000 --out gpurun_out/r02h_hnsw_1m_iid.json 2>&1 | tail -1 | cut -c1-900
timeout 600 python tools/bench_hnsw.py --rows 1000000 --queries 10000 --clusters 1000 --out gpurun_out/r02h_hnsw_1m_clustered.json 2>&1 | tail -1 | cut -c1-900
timeout 600 python tools/bench_hnsw.py --rows 1000000 --dim 128 --queries 10000 --clusters 1000 --out gpurun_out/r02h_hnsw_1m_128d_clustered.json 2>&1 | tail -1 | cut -c1-900

#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python tools/cmp_hnsw_build.py 100000 64 l2 2>&1 | tail -1 | tee gpurun_out/r02h_cmp_build_l2.json
timeout 900 python tools/cmp_hnsw_build.py 100000 64 cosine 2>&1 | tail -1 | tee gpurun_out/r02h_cmp_build_cosine.json

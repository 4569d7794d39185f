#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 60 python -m pytest tests/test_gpu_hnsw.py tests/test_host_cpp.py -q -x 2>&1 | tail -3

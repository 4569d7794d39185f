#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 60 python -m pytest tests/test_host_cpp.py tests/test_gpu_hnsw.py -q -x 2>&1 | tail -6

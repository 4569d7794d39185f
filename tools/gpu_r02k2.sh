#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 150 python -m pytest tests/test_gpu_join_sort_vector.py -q -x -k "join" 2>&1 | tail -10

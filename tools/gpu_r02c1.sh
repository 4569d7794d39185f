#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 120 python -m pytest tests/test_gpu_cast.py -q -x 2>&1 | tail -12

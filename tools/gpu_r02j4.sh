#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
DBHIP_TRACE=1 timeout 120 python -m pytest tests/test_gpu_fused.py -q -x 2>&1 | grep -v "^\[dbhip\] groupby" | tail -6

#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
DBHIP_TRACE=1 timeout 100 python -m pytest tests/test_gpu_fused.py -q -x 2>&1 | grep -v "^\[dbhip\] groupby" | tail -8
DBHIP_FAGG_JIT=sync timeout 60 python tools/prof_fagg.py 59986052 3 fused_program 2>&1 | tail -1 | cut -c1-260

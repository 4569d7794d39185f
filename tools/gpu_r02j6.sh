#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 120 python -m pytest tests/test_gpu_fused.py -q -x 2>&1 | tail -2
DBHIP_FAGG_JIT=sync DBHIP_FAGG_JIT_DUMP=$PWD/gpurun_out/fagg_jit2 timeout 100 python tools/prof_fagg.py 59986052 3 fused_program 2>&1 | tail -1 | cut -c1-260
DBHIP_FAGG_JIT=0 timeout 100 python tools/prof_fagg.py 59986052 3 fused_program 2>&1 | tail -1 | cut -c1-260

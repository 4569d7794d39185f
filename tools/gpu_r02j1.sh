#!/bin/bash
# run-time specialised fused aggregation: correctness (the fused tests run through it by default), then time vs interpreter
mkdir -p gpurun_out; export TMPDIR=/tmp
DBHIP_TRACE=1 timeout 900 python -m pytest tests/test_gpu_fused.py -q -x 2>&1 | grep -v "^\[dbhip\] groupby" | tail -12
echo "== jit"; DBHIP_TRACE=1 DBHIP_FAGG_JIT_DUMP=$PWD/gpurun_out/fagg_jit timeout 300 python tools/prof_fagg.py 59986052 3 fused_program,q1_fused 2>&1 | grep -v "groupby" | tail -6 | cut -c1-400
echo "== interpreter"; DBHIP_FAGG_JIT=0 timeout 300 python tools/prof_fagg.py 59986052 3 fused_program 2>&1 | tail -2 | cut -c1-400
ls -la gpurun_out/ | grep fagg_jit

#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
for i in 1 2 3; do timeout 120 python tools/dbg_fagg.py keep 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-100; done
timeout 300 python tools/prof_fagg.py 59986052 3 fused_program,q1_fused,pushdown,plain4 > gpurun_out/fagg_r02k.json 2> gpurun_out/fagg_r02k.err; echo "rc=$?"; tail -2 gpurun_out/fagg_r02k.err; cat gpurun_out/fagg_r02k.json
timeout 600 python -m pytest tests/test_gpu_fused.py -q -x 2>&1 | tail -15

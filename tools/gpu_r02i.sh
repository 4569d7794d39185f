#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
for M in keep keep drop; do timeout 120 python tools/dbg_fagg.py $M 2>&1 | grep -v amdgpu.ids | tail -1; done
DBHIP_FAGG_NOCHAIN=1 timeout 120 python tools/dbg_fagg.py keep 2>&1 | grep -v amdgpu.ids | tail -1
timeout 300 python tools/prof_fagg.py 59986052 3 fused_program,q1_fused,pushdown,plain4 > gpurun_out/fagg_r02i.json 2> gpurun_out/fagg_r02i.err; echo "rc=$?"; tail -2 gpurun_out/fagg_r02i.err; cat gpurun_out/fagg_r02i.json
timeout 600 python -m pytest tests/test_gpu_fused.py -q -x 2>&1 | tail -15

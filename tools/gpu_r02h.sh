#!/bin/bash
TAG=${1:-r02h}
mkdir -p gpurun_out; export TMPDIR=/tmp
for M in drop keep; do timeout 120 python tools/dbg_fagg.py $M 2>&1 | grep -v amdgpu.ids | tail -1; done
timeout 600 python -m pytest tests/test_gpu_fused.py tests/test_gpu_groupby2.py -q -x 2>&1 | tail -15 > gpurun_out/pytest_${TAG}_fused.log; cat gpurun_out/pytest_${TAG}_fused.log
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_host_cpp.py tests/test_q3.py -q -x 2>&1 | tail -6 > gpurun_out/pytest_${TAG}_rest.log; cat gpurun_out/pytest_${TAG}_rest.log
timeout 300 python tools/prof_fagg.py 59986052 3 fused_program,q1_fused,pushdown,plain4 > gpurun_out/fagg_${TAG}.json 2> gpurun_out/fagg_${TAG}.err; echo "rc=$?"; tail -2 gpurun_out/fagg_${TAG}.err; cat gpurun_out/fagg_${TAG}.json

#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_golden.py tests/test_gpu_fused.py -q 2>&1 | tail -25 > gpurun_out/pytest_r02l_a.log; cat gpurun_out/pytest_r02l_a.log
timeout 900 python -m pytest tests/test_gpu_groupby2.py -q -x -k "string_keys or state_block or partition" 2>&1 | tail -25 > gpurun_out/pytest_r02l_b.log; cat gpurun_out/pytest_r02l_b.log
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "groupby or cmp or decimal" 2>&1 | tail -8 > gpurun_out/pytest_r02l_c.log; cat gpurun_out/pytest_r02l_c.log

#!/bin/bash
TAG=${1:-r02c}
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_fused.py -q -x 2>&1 | tail -25 > gpurun_out/pytest_${TAG}_fused.log; cat gpurun_out/pytest_${TAG}_fused.log
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_groupby2.py tests/test_host_cpp.py -q -x 2>&1 | tail -12 > gpurun_out/pytest_${TAG}_rest.log; cat gpurun_out/pytest_${TAG}_rest.log
timeout 600 python bench.py --no-ann --no-q3 --no-cpu > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err; tail -3 gpurun_out/bench_$TAG.err; python - <<PY
import json
j=json.load(open("gpurun_out/bench_$TAG.json"))
print(j["value"], j["roofline"]["kernel_ms"])
print(json.dumps(j["q1_operator_plan"], indent=1))
PY

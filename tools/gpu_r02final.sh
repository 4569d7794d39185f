#!/bin/bash
# round 2 final validation: the whole GPU suite, smoke(), the default bench line
TAG=r02final
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q -x 2>&1 | tail -6 > gpurun_out/pytest_$TAG.log; cat gpurun_out/pytest_$TAG.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 600 python bench.py > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err; tail -2 gpurun_out/bench_$TAG.err; cut -c1-1500 gpurun_out/bench_$TAG.json

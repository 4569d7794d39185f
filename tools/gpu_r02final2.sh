#!/bin/bash
# round 2 final validation (after the run-time compiler helper, casts, join kinds, KeysU256): the whole GPU suite + smoke()
TAG=r02final2
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 400 python -m pytest tests -m gpu -q -x 2>&1 | tail -6 > gpurun_out/pytest_$TAG.log; cat gpurun_out/pytest_$TAG.log
timeout 60 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1

#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 100 python tools/microbench.py --only groupby --gb-card 1000000,10000000 2>&1 | grep name | cut -c1-150
DBHIP_GB_NODIRECT=1 timeout 100 python tools/microbench.py --only groupby --gb-card 10000000 2>&1 | grep name | cut -c1-150
timeout 600 python -m pytest tests/test_gpu_groupby2.py -q -x 2>&1 | tail -3

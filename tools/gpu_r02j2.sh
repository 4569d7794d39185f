#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== plain add_block, LDS path"; timeout 200 python tools/microbench.py --only groupby --gb-card 4,8 2>&1 | grep name | cut -c1-140
echo "== plain add_block, fagg auto + jit"; DBHIP_FAGG_AUTO=1 timeout 200 python tools/microbench.py --only groupby --gb-card 4,8 2>&1 | grep name | cut -c1-140
for g in 256 1024 2048; do echo "== grid $g"; DBHIP_FAGG_GRID=$g timeout 300 python tools/prof_fagg.py 59986052 3 fused_program 2>&1 | tail -1 | cut -c1-300; done

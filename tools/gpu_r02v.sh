#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 200 python tools/microbench.py --only groupby 2>&1 | grep name | cut -c1-150 | tee gpurun_out/r02x_gb.log
DBHIP_TRACE=1 timeout 100 python tools/microbench.py --only groupby --gb-card 10000000 2>&1 | grep dbhip | head -12
for card in 1000 10000000; do
  d=gpurun_out/prof_gb_$card; rm -rf $d
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$d -o gb -- python $GRAFT_REPO_ROOT/tools/microbench.py --only groupby --gb-card $card > $GRAFT_REPO_ROOT/gpurun_out/gb_$card.log 2>&1)
  f=$(find $d -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && cp "$f" gpurun_out/r02x_groupby_${card}_kernel_stats.csv
  rm -rf $d
done
timeout 600 python -m pytest tests/test_gpu_groupby2.py -q -x 2>&1 | tail -3

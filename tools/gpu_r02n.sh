#!/bin/bash
# round 2 run N: partition-exclusive merge + cursor-free scatter: correctness, then the cardinality sweep
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_groupby2.py tests/test_gpu_parity.py tests/test_gpu_fused.py -q -x -k "groupby or group_by or hashagg or agg" 2>&1 | tail -8 > gpurun_out/pytest_r02n.log; cat gpurun_out/pytest_r02n.log
timeout 300 python tools/microbench.py --only groupby 2>&1 | cut -c1-150 | tee gpurun_out/r02n_gb.log
DBHIP_LDS_R=4 timeout 300 python tools/microbench.py --only groupby --gb-card 4,200 2>&1 | cut -c1-150 | tee gpurun_out/r02n_gb_r4.log
DBHIP_TRACE=1 timeout 300 python tools/microbench.py --only groupby --gb-card 10000000 2>&1 | grep dbhip | head -20
for card in 1000 10000000; do
  d=gpurun_out/prof_gb_$card; rm -rf $d
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$d -o gb -- python $GRAFT_REPO_ROOT/tools/microbench.py --only groupby --gb-card $card > $GRAFT_REPO_ROOT/gpurun_out/gb_$card.log 2>&1)
  f=$(find $d -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && cp "$f" gpurun_out/r02n_groupby_${card}_kernel_stats.csv
  rm -rf $d
done

#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python tools/microbench.py --only groupby --out gpurun_out/r02y_microbench_groupby.json 2>&1 | grep name | cut -c1-150
for card in 1000 100000 10000000; do
  d=gpurun_out/prof_gb_$card; rm -rf $d
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$d -o gb -- python $GRAFT_REPO_ROOT/tools/microbench.py --only groupby --gb-card $card > $GRAFT_REPO_ROOT/gpurun_out/gb_$card.log 2>&1)
  f=$(find $d -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && cp "$f" gpurun_out/r02y_groupby_${card}_kernel_stats.csv
  rm -rf $d
done
timeout 900 python -m pytest tests/test_gpu_groupby2.py tests/test_gpu_fused.py tests/test_gpu_golden.py tests/test_gpu_threads.py -q -x 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "groupby or hash" 2>&1 | tail -3

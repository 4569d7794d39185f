#!/bin/bash
# round 2 run M: per-kernel breakdown of the generic group-by at each cardinality regime + the tests not yet run on the GPU
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python tools/microbench.py --only groupby --gb-card 4,200,1000 2>&1 | cut -c1-150
for card in 4 1000 100000 10000000; do
  d=gpurun_out/prof_gb_$card; rm -rf $d
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$d -o gb -- python $GRAFT_REPO_ROOT/tools/microbench.py --only groupby --gb-card $card > $GRAFT_REPO_ROOT/gpurun_out/gb_$card.log 2>&1)
  f=$(find $d -name '*kernel_stats.csv' | head -1)
  echo "== $card"; tail -2 gpurun_out/gb_$card.log; [ -n "$f" ] && head -12 "$f" | cut -c1-160
  [ -n "$f" ] && cp "$f" gpurun_out/r02m_groupby_${card}_kernel_stats.csv
  rm -rf $d
done

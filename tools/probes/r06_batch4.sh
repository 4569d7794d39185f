R=$PWD; cd /tmp; export TMPDIR=/tmp
for b in 32 64; do
  DBHIP_FAGG_PIPE_BATCH=$b rocprofv3 --kernel-trace --stats -d /tmp/bt$b -o bt -- $R/databend_amd/host/block_sweep --only-q1 --quick --out /tmp/bt$b.json > /dev/null 2>&1
  echo "== batch $b"; f=$(find /tmp/bt$b -name "*kernel_stats.csv" | head -1); head -12 $f | cut -c1-160
done

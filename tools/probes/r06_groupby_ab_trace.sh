# kernel traces of the 2e4 and 1e6 group cases: round-4 library vs HEAD, same box
R=$PWD; mkdir -p gpurun_out; cd /tmp; export TMPDIR=/tmp
for v in ab_r04 head; do
  d=$R; [ $v = ab_r04 ] && d=$R/ab_r04
  (cd /tmp && DBHIP_JIT_CACHE_DIR=/tmp/jit_$v timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/gbt_$v -o t -- python $d/tools/microbench.py --only groupby --gb-card ${GBCARD:-20000} > /tmp/gbt_$v.log 2>&1); tail -2 /tmp/gbt_$v.log | cut -c1-200
  f=$(find /tmp/gbt_$v -name "*kernel_stats.csv" | head -1)
  echo "== $v"; python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:16]:
    print('%-90s calls %5s  avg %10.1f us  total %9.3f ms' % (r['Name'][:90], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6))
PY
  cp "$f" $R/gpurun_out/r06_gbab_${v}_kernel_stats.csv
done

#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for SET in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_ATOMIC_sum TCC_REQ_sum" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_BUSY_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"; do
  P=$(echo $SET | tr ' ' '_')
  d=gpurun_out/pmc_r02s_$P; rm -rf $d
  (cd /tmp && timeout 200 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $R/$d -o gb -- python $R/tools/microbench.py --only groupby --gb-card 10000000 > $R/gpurun_out/pmc_r02s_$P.log 2>&1)
  f=$(find $d -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python - "$f" <<'PY'
import csv,sys,collections
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k=r['Kernel_Name'].replace('(anonymous namespace)::','').split('(')[0][:40]
    agg[k][r['Counter_Name']]+=float(r['Counter_Value']); cnt[(k,r['Counter_Name'])]+=1
for k,v in agg.items():
    if 'gb_part' in k or 'probe' in k:
        print(k, {c: (x, cnt[(k,c)]) for c,x in v.items()})
PY
  rm -rf $d
done

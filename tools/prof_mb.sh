#!/bin/bash
# kernel-trace profile of one microbench family: bash tools/prof_mb.sh <tag> <only> [extra args]
TAG=$1; ONLY=$2; shift 2
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$PWD; cd /tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_$TAG -o t -- python $R/tools/microbench.py --only $ONLY "$@" > $R/gpurun_out/prof_$TAG.log 2>&1
cd $R
python - <<PY
import csv,collections,glob
f=glob.glob("gpurun_out/prof_$TAG/**/*kernel_trace.csv", recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
agg=collections.OrderedDict()
for r in rows:
    k=r['Kernel_Name'][:80]
    d=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
    a=agg.setdefault(k,[0,0.0,0.0]); a[0]+=1; a[1]+=d; a[2]=max(a[2],d)
for k,(n,t,m) in agg.items(): print(f"{k:80s} n={n:5d} avg={t/n:9.1f}us max={m:9.1f} tot={t/1e3:9.2f}ms")
PY
grep '^{' gpurun_out/prof_$TAG.log | cut -c1-160

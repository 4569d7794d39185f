#!/bin/bash
# bash tools/pmc_any.sh <tag> <kernel-substring> "<counters pass1>" "<counters pass2>" -- <microbench args>
TAG=$1; KSUB=$2; shift 2
SETS=()
while [ "$1" != "--" ]; do SETS+=("$1"); shift; done; shift
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$PWD; cd /tmp
P=0
for SET in "${SETS[@]}"; do
  P=$((P+1))
  timeout 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${TAG}_$P -o v -- python $R/tools/microbench.py "$@" > $R/gpurun_out/pmc_${TAG}_$P.log 2>&1
done
cd $R
python - <<PY
import csv,glob,collections
for f in sorted(glob.glob("gpurun_out/pmc_${TAG}_*/**/*counter_collection.csv", recursive=True)):
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"][:50]
        agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); n[(k,r["Counter_Name"])]+=1
    for k,v in agg.items():
        if "$KSUB" in k: print(k, {c:(x, n[(k,c)]) for c,x in v.items()})
PY

#!/bin/bash
TAG=${1:-r02d}
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python tools/prof_fagg.py > gpurun_out/fagg_$TAG.json 2> gpurun_out/fagg_$TAG.err; tail -3 gpurun_out/fagg_$TAG.err; cat gpurun_out/fagg_$TAG.json
R=$PWD; cd /tmp
rocprofv3 --list-avail 2>/dev/null | grep -i -E "ICACHE|IFETCH|SQC_" | head -30 > $R/gpurun_out/counters_$TAG.txt; cat $R/gpurun_out/counters_$TAG.txt | head -30
P=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS" "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH"; do
  P=$((P+1))
  timeout 200 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${TAG}_$P -o f -- python $R/tools/prof_fagg.py 59986052 1 fused_program,q1_fused > $R/gpurun_out/pmc_${TAG}_$P.log 2>&1
done
cd $R
python - <<PY
import csv,glob,collections
for f in sorted(glob.glob("gpurun_out/pmc_${TAG}_*/**/*counter_collection.csv", recursive=True)):
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"][:60]
        if "fagg" in k or "q1_fused" in k:
            agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); n[(k,r["Counter_Name"])]+=1
    for k,v in agg.items(): print(k, {c:(round(x/n[(k,c)]), n[(k,c)]) for c,x in v.items()})
PY

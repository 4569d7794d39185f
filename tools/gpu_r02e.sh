#!/bin/bash
TAG=${1:-r02e}
mkdir -p gpurun_out; export TMPDIR=/tmp
for W in fused_program,q1_fused pushdown plain4; do
  timeout 300 python tools/prof_fagg.py 59986052 3 $W > gpurun_out/fagg_${TAG}_$W.json 2> gpurun_out/fagg_${TAG}_$W.err; echo "rc=$? $W"; tail -2 gpurun_out/fagg_${TAG}_$W.err; cat gpurun_out/fagg_${TAG}_$W.json
done
timeout 900 python -m pytest tests/test_gpu_fused.py -q -x 2>&1 | tail -25 > gpurun_out/pytest_${TAG}_fused.log; cat gpurun_out/pytest_${TAG}_fused.log
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "fused_expression or decimal" 2>&1 | tail -12 > gpurun_out/pytest_${TAG}_rest.log; cat gpurun_out/pytest_${TAG}_rest.log
R=$PWD; cd /tmp
P=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM" "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES"; do
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

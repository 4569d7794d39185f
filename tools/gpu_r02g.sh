#!/bin/bash
TAG=${1:-r02g}
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 120 python tools/dbg_fagg.py drop 2>&1 | grep -v amdgpu.ids | tail -2
timeout 300 python tools/prof_fagg.py 59986052 3 fused_program,q1_fused,plain4 > gpurun_out/fagg_${TAG}.json 2> gpurun_out/fagg_${TAG}.err; echo "rc=$?"; tail -2 gpurun_out/fagg_${TAG}.err; cat gpurun_out/fagg_${TAG}.json
timeout 600 python -m pytest tests/test_gpu_fused.py -q 2>&1 | tail -25 > gpurun_out/pytest_${TAG}_fused.log; cat gpurun_out/pytest_${TAG}_fused.log
R=$PWD; cd /tmp
P=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM" "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES"; do
  P=$((P+1))
  timeout 100 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${TAG}_$P -o f -- python $R/tools/prof_fagg.py 59986052 1 fused_program > $R/gpurun_out/pmc_${TAG}_$P.log 2>&1
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

#!/bin/bash
# MFMA utilisation / stall counters for the distance GEMM (separate --pmc passes, kernel-trace only).
TAG=${1:-run}; NQ=${2:-512}
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$PWD; cd /tmp
rocprofv3 -L 2>/dev/null | grep -ioE "SQ_[A-Z_0-9]*MFMA[A-Z_0-9]*|SQ_BUSY_CYCLES|SQ_BUSY_CU_CYCLES|GRBM_GUI_ACTIVE|SQ_WAVE_CYCLES|SQ_WAIT_INST_ANY|SQ_WAIT_ANY|SQ_ACTIVE_INST_ANY|SQ_ACTIVE_INST_VALU|SQ_INSTS_VALU_MFMA[A-Z_0-9]*|SQ_LDS_BANK_CONFLICT|SQ_WAIT_INST_LDS|TCC_HIT_sum|TCC_MISS_sum" | sort -u > $R/gpurun_out/pmc_avail_$TAG.txt
P=0
for SET in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE"; do
  P=$((P+1))
  timeout 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $R/gpurun_out/pmcvec_${TAG}_$P -o v -- python $R/tools/microbench.py --only vector --vec-nq $NQ > $R/gpurun_out/pmcvec_${TAG}_$P.log 2>&1
  tail -2 $R/gpurun_out/pmcvec_${TAG}_$P.log | cut -c1-200
done
cd $R
python - <<PY
import csv,glob,collections
for f in sorted(glob.glob("gpurun_out/pmcvec_${TAG}_*/**/*counter_collection.csv", recursive=True)):
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"][:60]
        agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); 
    for k,v in agg.items():
        if "dot_tile" in k or "select" in k or "diff_valu" in k: print(f, k, dict(v))
PY

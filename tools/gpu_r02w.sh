#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for SET in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU" "SQ_LDS_ATOMIC_RETURN SQ_LDS_MEM_VIOLATIONS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  P=$(echo $SET | tr ' ' '_' | cut -c1-40)
  d=gpurun_out/pmc_r02w_$P; rm -rf $d
  (cd /tmp && timeout 200 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $R/$d -o gb -- python $R/tools/microbench.py --only groupby --gb-card 1000 > $R/gpurun_out/pmc_r02w_$P.log 2>&1)
  f=$(find $d -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python - "$f" <<'PY'
import csv,sys,collections
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k=r['Kernel_Name'].replace('(anonymous namespace)::','').split('(')[0][:40]
    agg[k][r['Counter_Name']]+=float(r['Counter_Value']); cnt[k]+=1
for k,v in agg.items():
    if 'gb_part_agg' in k or 'scatter' in k or 'preagg' in k:
        print(k, cnt[k], {c: x for c,x in v.items()})
PY
  [ -z "$f" ] && tail -3 $R/gpurun_out/pmc_r02w_$P.log
  rm -rf $d
done

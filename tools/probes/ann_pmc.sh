# SQ counters of the bf16 filter kernel at 1.25 M x 768, 2048 queries (tools/probes/ann_dim_probe.py, dim 768 only)
export TMPDIR=/tmp; R=$PWD; mkdir -p gpurun_out
for C in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16"; do
  d=$R/gpurun_out/pmc_ann; rm -rf $d
  (cd /tmp && ANN_DIMS=768 timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $d -o p -- python $R/tools/probes/ann_dim_probe.py 2048 > /dev/null 2> $d.err)
  f=$(find $d -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python tools/pmc_sum.py "$f" | grep -i "kernel,\|filter256" | cut -c1-200
  rm -rf $d
done

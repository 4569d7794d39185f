# same-process-family A/B of the bf16 filter's schedules (experiments build): kernel time from rocprofv3 --stats
export TMPDIR=/tmp; R=$PWD; mkdir -p gpurun_out
for V in 1 0 1 0; do
  d=$R/gpurun_out/prof_ab; rm -rf $d
  (cd /tmp && DBHIP_BF16_V=$V ANN_DIMS=768 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $d -o p -- python $R/tools/probes/ann_dim_probe.py 2048 > $d.out 2> $d.err)
  echo "V=$V $(cat $d.out | cut -c1-140)"; grep "filter256" $(find $d -name '*kernel_stats.csv' | head -1) | cut -d, -f1-4 | cut -c1-160
  rm -rf $d
done
timeout 900 python -m pytest tests/test_gpu_join_sort_vector.py -m gpu -q -x --timeout 300 -k "vec or vector" 2>&1 | grep -E "passed|failed|error" | tail -3

# experiments build: blocks per multi-block launch of the pipelined fused aggregation (DBHIP_FAGG_PIPE_BATCH), Q1 at 65,536-row blocks
for b in 32 64 128; do echo "== batch $b"; DBHIP_FAGG_PIPE_BATCH=$b ./databend_amd/host/block_sweep --only-q1 --quick --out gpurun_out/r06_batch_$b.json > /dev/null 2>&1; python - "$b" <<'PY'
import json,sys
b=json.load(open('gpurun_out/r06_batch_%s.json'%sys.argv[1]))
for l in b['lines']:
    if l['op']=='q1_pipelined': print(l['block_rows'], l['threads'], l['g_rows_per_s'], l['us_per_call_per_thread'])
PY
done

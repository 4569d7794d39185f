timeout 600 python -m pytest tests/test_gpu_pipelined.py -m gpu -q -x --timeout 300 2>&1 | grep -E "passed|failed|FAILED|Error|assert" | head -8
timeout 300 ./databend_amd/host/block_sweep --only-q1 --out gpurun_out/r06_window512_sweep.json > /dev/null 2>&1
python - <<'PY'
import json
b=json.load(open('gpurun_out/r06_window512_sweep.json'))
for l in b['lines']:
    if 'pipelined' in l['op']: print(l['op'], l['block_rows'], l['threads'], l['g_rows_per_s'], l['us_per_call_per_thread'], l.get('host_us_inside_call'), l['equals_whole_table'])
PY

timeout 600 python -m pytest tests/test_gpu_pipelined.py tests/test_gpu_fused.py tests/test_host_cpp.py -m gpu -q -x --timeout 300 2>&1 | grep -E "passed|failed|FAILED|Error" | head -5
timeout 300 ./databend_amd/host/block_sweep --out gpurun_out/r06_block_size_sweep.json > gpurun_out/r06_block_size_sweep.err 2>&1
python - <<'PY'
import json
b=json.load(open('gpurun_out/r06_block_size_sweep.json'))
for l in b['lines']:
    if 'pipelined' in l['op'] or 'pipe' in l['op']: print(l['op'], l['block_rows'], l['threads'], l['g_rows_per_s'], l['us_per_call_per_thread'], l.get('host_us_inside_call'), l['equals_whole_table'])
PY

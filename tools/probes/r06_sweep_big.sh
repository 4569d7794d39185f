timeout 900 ./databend_amd/host/block_sweep --only-q1 --rows $((512<<20)) --out gpurun_out/r06_sweep_512Mi.json > gpurun_out/r06_sweep_512Mi.err 2>&1; tail -3 gpurun_out/r06_sweep_512Mi.err
python - <<'PY'
import json
b=json.load(open('gpurun_out/r06_sweep_512Mi.json'))
for l in b['lines']:
    if l['op'].startswith('q1_'): print(l['op'], l['block_rows'], l['threads'], l['g_rows_per_s'], l['us_per_call_per_thread'], l.get('host_us_inside_call'), l['equals_whole_table'])
PY

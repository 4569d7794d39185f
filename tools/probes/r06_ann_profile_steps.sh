# ANN headline configuration (10 M x 768, 10 000 queries per step) on HEAD: kernel stats, MFMA-busy / GUI-active pass, FETCH / WRITE passes
A="--steps 2 --warmup 1 --no-cpu --no-q3 --no-opplan --no-readiness --no-blocks --no-hnsw --ann-steps 2"
bash tools/gpu_run.sh r06_vector_10m "prof:$A" "pmc:SQ_VALU_MFMA_BUSY_CYCLES,GRBM_GUI_ACTIVE:$A" "pmc:FETCH_SIZE:$A" "pmc:WRITE_SIZE:$A"
python - <<'PY'
import json
j = json.loads(open("gpurun_out/prof_bench_r06_vector_10m.json").read().strip().splitlines()[-1])
json.dump(j.get("ann"), open("gpurun_out/r06_vector_10m_bench_ann.json", "w"), indent=1)
print({k: j["ann"][k] for k in ("value", "ms_per_step", "recall_at_10") if k in j["ann"]})
PY
timeout 900 python -m pytest tests/test_gpu_parquet_device.py tests/test_gpu_comm.py tests/test_host_cpp.py tests/test_gpu_pipelined.py -m gpu -q -x --timeout 300 2>&1 | grep -E "passed|failed|error" | tail -3

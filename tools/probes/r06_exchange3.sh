# round 6: the single-workgroup merge + queued exchange: parity of everything that merges, then the host timeline of the world-of-one step
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_groupby2.py tests/test_gpu_pipelined.py tests/test_gpu_comm.py tests/test_gpu_groupby_compact.py tests/test_gpu_fused.py tests/test_host_cpp.py -m gpu -q -x --timeout 300 2>&1 | tail -4
python tools/probes/exchange_timeline.py
python tools/probes/exchange_timeline.py 4000000
python bench.py --no-cpu --no-ann --no-q3 --no-opplan --no-blocks 2>/dev/null | python -c "import json,sys; b=json.loads(sys.stdin.read()); print('readiness', b['multi_gpu_readiness']); print('value', b['value'], b['ms_per_step'])"
./databend_amd/host/block_sweep --only-q1 --quick --out gpurun_out/r06h_sweep.json 2>&1 | grep -E "q1_sync|q1_pipelined" | head -8

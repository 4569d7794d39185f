timeout 1200 python -m pytest tests/test_gpu_comm.py tests/test_gpu_groupby2.py tests/test_gpu_pipelined.py tests/test_gpu_parity.py -m gpu -q -x --timeout 300 2>&1 | tail -4
python tools/probes/exchange_timeline.py
python tools/probes/exchange_timeline.py 4000000
./databend_amd/host/block_sweep --only-q1 --quick --out gpurun_out/r06h_sweep.json 2>&1 | grep -E "q1_sync|q1_pipelined" | head -4

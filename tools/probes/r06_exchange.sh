# round 6: the queued block exchange (gbk_api.h blocks_queued) — parity tests, then the world-of-one exchange overhead with and without the
# block-size sweep running before it in the same process
timeout 900 python -m pytest tests/test_gpu_groupby2.py tests/test_gpu_comm.py tests/test_gpu_parity.py -m gpu -q -x --timeout 300 -k "block or exchange or partition or rank or world or allgather" 2>&1 | tail -5
for i in 1 2; do python bench.py --no-cpu --no-ann --no-q3 --no-opplan --no-blocks 2>/dev/null | python -c "import json,sys; b=json.loads(sys.stdin.read()); print('no-blocks', b['multi_gpu_readiness'])"; done
python bench.py --no-cpu --no-ann --no-q3 --no-opplan 2>/dev/null | python -c "import json,sys; b=json.loads(sys.stdin.read()); print('with-blocks', b['multi_gpu_readiness'])"

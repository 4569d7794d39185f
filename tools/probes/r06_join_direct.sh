timeout 600 python -m pytest tests/test_gpu_join_sort_vector.py -m gpu -q -x --timeout 300 -k "join" 2>&1 | grep -E "passed|failed|FAILED|Error|assert" | head -8
timeout 300 python tools/bench_q3.py --main-only 2>&1 | tail -1 | cut -c1-300
timeout 300 python tools/microbench.py --only join 2>&1 | grep -i "join" | cut -c1-200

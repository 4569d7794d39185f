timeout 300 python -m pytest tests/test_gpu_groupby2.py -m gpu -q -x --timeout 300 -k "queued" 2>&1 | tail -3
python tools/probes/exchange_timeline.py
python tools/probes/exchange_timeline.py 4000000
R=$PWD; mkdir -p gpurun_out; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/xt -o xt -- python $R/tools/probes/exchange_timeline.py 4000000 > /dev/null 2>&1
cd $R; python tools/trace_summary.py /tmp/xt 2>&1 | tail -40 > gpurun_out/r06_exchange_trace.txt; tail -40 gpurun_out/r06_exchange_trace.txt

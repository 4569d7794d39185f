mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 700 python -m pytest tests -m gpu -q -x 2>&1 | tail -4 > gpurun_out/pytest_final.log; cat gpurun_out/pytest_final.log
timeout 500 python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err; tail -2 gpurun_out/bench_final.err; cut -c1-600 gpurun_out/bench_final.json
R=$PWD; cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_final -o q1 -- python $R/bench.py --steps 10 --warmup 2 --no-cpu --no-ann > $R/gpurun_out/prof_bench_final.json 2> $R/gpurun_out/prof_bench_final.err
cd $R; find gpurun_out/prof_final -name "*kernel_stats*" | head -1 | xargs -r head -8

#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 80 python bench.py --steps 5 --warmup 1 --no-cpu --no-q3 --no-ann > gpurun_out/bench_r02final3.json 2> gpurun_out/bench_r02final3.err; tail -2 gpurun_out/bench_r02final3.err | cut -c1-200; python -c "
import json; d=json.load(open('gpurun_out/bench_r02final3.json')); print(round(d['value']/1e9,1), {k:(round(v['ms'],2), v.get('prepare_ms')) for k,v in d['q1_operator_plan'].items()})"

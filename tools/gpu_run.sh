#!/bin/bash
# The ONE GPU-box runner (replaces the per-run scripts of round 2). From the repo root on the box:
#   bash tools/gpu_run.sh <tag> <step> [<step> ...]
# steps (each writes under gpurun_out/, named by <tag>):
#   tests[:<pytest args>]     pytest -m gpu (default: the whole suite)
#   bench[:<bench args>]      python bench.py <args>                  -> bench_<tag>.json
#   prof[:<bench args>]       rocprofv3 --kernel-trace --stats of bench.py -> <tag>_kernel_stats.csv
#   pmc:<C1,C2..>[:<bench args>]  one --pmc pass (kernel-trace only) of bench.py -> <tag>_pmc_<C1>.csv
#   mb:<microbench args>      python tools/microbench.py <args>       -> mb_<tag>.json
#   mbprof:<microbench args>  rocprofv3 --kernel-trace --stats of tools/microbench.py
#   py:<script and args>      python <script …>                       -> py_<tag>.log
#   mbpmc:<C1,C2..>:<microbench args>  one --pmc pass of tools/microbench.py -> <tag>_pmc_<C1>.csv
#   pypmc:<C1,C2..>:<script and args>  one --pmc pass (kernel-trace only) of python <script …> -> <tag>_pmc_<C1>.csv
#   pyprof:<script and args>  rocprofv3 --kernel-trace --stats of python <script …> -> <tag>_kernel_stats.csv
TAG=$1; shift
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$PWD
stats() {  # $1 = profile dir, $2 = destination csv
  f=$(find "$1" -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" "$2" && head -12 "$2" | cut -c1-200
}
for STEP in "$@"; do
  KIND=${STEP%%:*}; ARG=""; [ "$STEP" != "$KIND" ] && ARG=${STEP#*:}
  case $KIND in
    tests)
      timeout 1500 python -m pytest ${ARG:-tests} -m gpu -q -x --timeout 300 > gpurun_out/pytest_full_$TAG.log 2>&1; tail -60 gpurun_out/pytest_full_$TAG.log | cut -c1-400 > gpurun_out/pytest_$TAG.log; tail -8 gpurun_out/pytest_$TAG.log ;;
    bench)
      timeout 900 python bench.py $ARG > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err
      tail -2 gpurun_out/bench_$TAG.err | cut -c1-300; cut -c1-1200 gpurun_out/bench_$TAG.json ;;
    prof)
      d=$R/gpurun_out/prof_$TAG; rm -rf $d
      (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $d -o p -- python $R/bench.py ${ARG:---steps 10 --warmup 2 --no-cpu --no-ann --no-q3 --no-opplan --no-readiness --no-blocks --no-scan} > $R/gpurun_out/prof_bench_$TAG.json 2> $R/gpurun_out/prof_bench_$TAG.err)
      stats $d gpurun_out/${TAG}_kernel_stats.csv; rm -rf $d ;;
    pmc)
      C=${ARG%%:*}; BA=""; [ "$ARG" != "$C" ] && BA=${ARG#*:}
      d=$R/gpurun_out/pmc_${TAG}_${C%%,*}; rm -rf $d
      (cd /tmp && timeout 600 rocprofv3 --pmc ${C//,/ } --kernel-trace --output-format csv -d $d -o p -- python $R/bench.py ${BA:---steps 3 --warmup 1 --no-cpu --no-ann --no-q3 --no-opplan --no-readiness --no-blocks --no-scan} > $d.json 2> $d.err)
      f=$(find $d -name '*counter_collection.csv' | head -1)
      [ -n "$f" ] && python tools/pmc_sum.py "$f" > gpurun_out/${TAG}_pmc_${C%%,*}.csv && head -20 gpurun_out/${TAG}_pmc_${C%%,*}.csv
      rm -rf $d ;;
    mb)
      timeout 900 python tools/microbench.py $ARG --out gpurun_out/mb_$TAG.json 2>&1 | grep -v '^$' | cut -c1-220 | tail -60 ;;
    mbprof)
      d=$R/gpurun_out/prof_$TAG; rm -rf $d
      (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $d -o p -- python $R/tools/microbench.py $ARG > $R/gpurun_out/mbprof_$TAG.log 2>&1)
      stats $d gpurun_out/${TAG}_kernel_stats.csv; rm -rf $d ;;
    py)
      timeout 1200 python $ARG > gpurun_out/py_$TAG.log 2>&1; tail -40 gpurun_out/py_$TAG.log | cut -c1-300 ;;
    pyprof)
      d=$R/gpurun_out/prof_$TAG; rm -rf $d
      (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $d -o p -- python $R/$ARG > $R/gpurun_out/pyprof_$TAG.log 2>&1)
      stats $d gpurun_out/${TAG}_kernel_stats.csv; rm -rf $d ;;
    mbpmc)
      C=${ARG%%:*}; SA=${ARG#*:}
      d=$R/gpurun_out/pmc_${TAG}_${C%%,*}; rm -rf $d
      (cd /tmp && timeout 900 rocprofv3 --pmc ${C//,/ } --kernel-trace --output-format csv -d $d -o p -- python $R/tools/microbench.py $SA > $d.json 2> $d.err)
      f=$(find $d -name '*counter_collection.csv' | head -1)
      [ -n "$f" ] && python tools/pmc_sum.py "$f" > gpurun_out/${TAG}_pmc_${C%%,*}.csv && head -20 gpurun_out/${TAG}_pmc_${C%%,*}.csv
      rm -rf $d ;;
    pypmc)
      C=${ARG%%:*}; SA=${ARG#*:}
      d=$R/gpurun_out/pmc_${TAG}_${C%%,*}; rm -rf $d
      (cd /tmp && timeout 900 rocprofv3 --pmc ${C//,/ } --kernel-trace --output-format csv -d $d -o p -- python $R/$SA > $d.json 2> $d.err)
      f=$(find $d -name '*counter_collection.csv' | head -1)
      [ -n "$f" ] && python tools/pmc_sum.py "$f" > gpurun_out/${TAG}_pmc_${C%%,*}.csv && head -20 gpurun_out/${TAG}_pmc_${C%%,*}.csv
      rm -rf $d ;;
    *) echo "unknown step $STEP" ;;
  esac
done

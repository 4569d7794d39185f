mkdir -p gpurun_out; export TMPDIR=/tmp; R=$PWD
for D in "" "-DFA_X_SKIP_EPI" "-DFA_X_SKIP_RESOLVE" "-DFA_X_SKIP_LOOP"; do
  export DBHIP_FAGG_JIT_DEFS="$D"
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_x -o p -- $R/databend_amd/host/block_sweep --quick --only-q1 --rows 8388608 > /dev/null 2> $R/gpurun_out/x.err)
  echo "== defs '$D'"; grep "q1_pipelined" gpurun_out/x.err | tail -4 | cut -c1-120
  python tools/trace_summary.py $(find gpurun_out/prof_x -name "*kernel_trace.csv" | head -1) 30 | grep fagg_jit
  rm -rf gpurun_out/prof_x
done

#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
export DBHIP_FAGG_JIT=sync
for d in "" "-DFA_JIT_GLOBAL" "-DFA_JIT_ROWS=3 -DFA_JIT_GLOBAL" "-DFA_JIT_ROWS=4 -DFA_JIT_GLOBAL"; do
  echo "== defs: $d"
  DBHIP_FAGG_JIT_DEFS="$d" timeout 100 python tools/prof_fagg.py 59986052 3 fused_program 2>&1 | tail -1 | cut -c1-260
done

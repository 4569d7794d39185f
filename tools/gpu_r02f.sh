#!/bin/bash
mkdir -p gpurun_out
for M in check keep drop same; do
  timeout 120 python tools/dbg_fagg.py $M 2>&1 | grep -v amdgpu.ids | tail -4
done
AMD_SERIALIZE_KERNEL=3 timeout 120 python tools/dbg_fagg.py drop 2>&1 | grep -v amdgpu.ids | tail -4
timeout 120 python tools/dbg_fagg.py drop 6000000 2>&1 | grep -v amdgpu.ids | tail -3

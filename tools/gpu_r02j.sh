#!/bin/bash
run() { for i in 1 2 3 4; do timeout 120 python tools/dbg_fagg.py keep 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-100; done; }
echo "== baseline"; run
echo "== NOCHAIN"; DBHIP_FAGG_NOCHAIN=1 run
echo "== debug 4 (no merge)"; DBHIP_FAGG_DEBUG=4 run
echo "== debug 5 (no emit, no merge)"; DBHIP_FAGG_DEBUG=5 run
echo "== debug 7 (loads+interp only)"; DBHIP_FAGG_DEBUG=7 run

"""round 6 probe: where the world-of-one exchange step's time goes — host wall time of every C-ABI call of bench.py's
multi_gpu_readiness step (reset, q1_fused, last_kernel_ms, exchange_alltoall split into its three calls), 300 steps.
usage: python tools/probes/exchange_timeline.py [rows]"""
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from databend_amd import device as D, tpch
from databend_amd._lib import check, lib

D.init(0)
L = lib()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 75_004_736
li = tpch.LineitemTorch(n, seed=2, torch=torch, row0=0)
g = D.GroupBy.q1()
W = g.row_bytes() // 8
blocks = D.DeviceBuffer(2 * 257 * W * 8)
recv = D.DeviceBuffer(2 * 257 * W * 8)
names = ["reset", "q1_fused", "last_kernel_ms", "partition_blocks", "copy", "replace_with_blocks"]
acc = {k: 0.0 for k in names}
kms = []


def step(record):
    t = [time.perf_counter()]
    g.reset(); t.append(time.perf_counter())
    D.q1_fused(g, li.qty, li.price, li.disc, li.tax, li.rf, li.ls, li.ship, tpch.Q1_CUTOFF); t.append(time.perf_counter())
    ms = C.c_float(); check(L.dbhip_last_kernel_ms(C.byref(ms))); t.append(time.perf_counter())
    g.partition_blocks(blocks.ptr, 1, 256); t.append(time.perf_counter())
    check(L.dbhip_memcpy_d2d(C.c_void_p(recv.ptr), C.c_void_p(blocks.ptr), C.c_size_t(257 * W * 8), None)); t.append(time.perf_counter())
    g.replace_with_blocks(recv.ptr, 1, 256); t.append(time.perf_counter())
    if record:
        for i, k in enumerate(names):
            acc[k] += t[i + 1] - t[i]
        kms.append(ms.value)


for _ in range(20):
    step(False)
check(L.dbhip_stream_sync(None))
steps = 300
t0 = time.perf_counter()
for _ in range(steps):
    step(True)
check(L.dbhip_stream_sync(None))
wall = (time.perf_counter() - t0) * 1e3 / steps
out = {"rows": n, "step_ms": wall, "kernel_ms": sum(kms) / len(kms), "overhead_ms": wall - sum(kms) / len(kms),
       "host_us_per_call": {k: round(v * 1e6 / steps, 2) for k, v in acc.items()}}
print(json.dumps(out))

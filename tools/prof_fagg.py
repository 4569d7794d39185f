#!/usr/bin/env python3
"""Times the generic fused filter->map->aggregate kernel (tpch.q1_fused_program) next to the query-specific fused kernel on
an SF10-sized lineitem (profiling driver: tools/gpu_run.sh <tag> py:tools/prof_fagg.py)."""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from databend_amd import device as D, tpch  # noqa: E402
from databend_amd._lib import check, lib  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 59_986_052
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
which = sys.argv[3].split(",") if len(sys.argv) > 3 else ["fused_program", "q1_fused", "pushdown", "plain4"]
D.init(0)
L = lib()
li = tpch.LineitemTorch(n, seed=2, torch=torch)
out = {}


def timed(fn):
    fn()
    ts = []
    for _ in range(reps):
        check(L.dbhip_stream_sync(None))
        t0 = time.perf_counter()
        fn()
        check(L.dbhip_stream_sync(None))
        ms = C.c_float()
        check(L.dbhip_last_kernel_ms(C.byref(ms)))
        ts.append(((time.perf_counter() - t0) * 1e3, ms.value))
    return ts


ref = tpch.q1_rows(tpch.q1_fused(li))
if "fused_program" in which:
    assert tpch.q1_rows(tpch.q1_fused_program(li)) == ref
    out["fused_program"] = timed(lambda: tpch.q1_fused_program(li))
if "q1_fused" in which:
    out["q1_fused"] = timed(lambda: tpch.q1_fused(li))
if "pushdown" in which:
    out["pushdown"] = timed(lambda: tpch.q1_operator_pushdown(li))
if "plain4" in which:
    # plain add_block, 4 groups, i64 key + sum + count (the microbench row "generic group-by at 4 groups")
    from databend_amd import _lib as T
    k = torch.randint(0, 4, (n,), device="cuda", dtype=torch.int64)
    a = torch.randint(-10**9, 10**9, (n,), device="cuda", dtype=torch.int64)
    ck = D.Column(T.T_I64, n, D.BorrowedBuffer.of_tensor(k))
    ca = D.Column(T.T_I64, n, D.BorrowedBuffer.of_tensor(a))

    def plain():
        g = D.GroupBy([T.T_I64], [(T.AGG_SUM, T.T_I64, 0, 0, 0), (T.AGG_COUNT, 0, 0, 0, 0)])
        g.add_block([ck], [ca, None], n)
        return g
    got = sorted(plain().result())
    assert [r[2] for r in got] == torch.bincount(k, minlength=4).tolist()
    out["plain4"] = timed(plain)
print(json.dumps({"rows": n, "ms (wall, last kernel)": out}))

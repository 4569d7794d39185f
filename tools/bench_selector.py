#!/usr/bin/env python3
"""The Selector against the Bitmap path on a conjunction whose first predicate is selective (600 M rows):
   l_shipdate <= cutoff_low (5 %)  AND  l_discount >= 5  AND  l_quantity < 2400
Bitmap path: three dbhip_cmp over all rows + two dbhip_bitmap_binary + dbhip_filter_select.  Selector: dbhip_select_cmp x 3, the second and
third only on the rows still alive (tools/gpu_run.sh <tag> py:tools/bench_selector.py)."""
import json
import os
import sys
import time

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from databend_amd import _lib as L  # noqa: E402
from databend_amd import device as D, tpch  # noqa: E402
from databend_amd._lib import check, lib  # noqa: E402

D.init(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 600_037_902
li = tpch.LineitemTorch(n, seed=2, torch=torch)
sync = lambda: check(lib().dbhip_stream_sync(None))  # noqa: E731
out = {}
for name, cut in (("first predicate keeps 5 %", 8036 + 126), ("first predicate keeps 98 %", tpch.Q1_CUTOFF)):
    c_cut, c5, c24 = D.Column.scalar(cut, L.T_DATE), D.Column.scalar(5, L.T_DEC64, 15, 2), D.Column.scalar(2400, L.T_DEC64, 15, 2)

    def bitmap_path():
        p1 = D.cmp(L.CMP_LTE, li.ship, c_cut, n)
        p2 = D.cmp(L.CMP_GTE, li.disc, c5, n)
        p3 = D.cmp(L.CMP_LT, li.qty, c24, n)
        both = D.DeviceBuffer(((n + 63) // 64) * 8 + 8)
        check(lib().dbhip_bitmap_binary(0, D.C.c_void_p(p1.data.ptr), D.C.c_void_p(p2.data.ptr), D.C.c_int64(n), D.C.c_void_p(both.ptr), None))
        check(lib().dbhip_bitmap_binary(0, D.C.c_void_p(both.ptr), D.C.c_void_p(p3.data.ptr), D.C.c_int64(n), D.C.c_void_p(both.ptr), None))
        return D.filter_select(D.Column(L.T_BOOL, n, both))

    def selector_path():
        return D.select_tree(("and", [("cmp", L.CMP_LTE, li.ship, c_cut), ("cmp", L.CMP_GTE, li.disc, c5), ("cmp", L.CMP_LT, li.qty, c24)]), n)

    res = {}
    for pname, fn in (("bitmap", bitmap_path), ("selector", selector_path)):
        sel, k = fn()
        sync()
        ts = []
        for _ in range(3):
            sync(); t0 = time.perf_counter(); sel, k = fn(); sync(); ts.append((time.perf_counter() - t0) * 1e3)
        res[pname] = {"ms": min(ts), "rows_selected": int(k), "first_rows": sel.to_numpy(np.uint32, min(k, 4)).tolist()}
    assert res["bitmap"]["rows_selected"] == res["selector"]["rows_selected"] and res["bitmap"]["first_rows"] == res["selector"]["first_rows"]
    out[name] = res
print(json.dumps({"rows": n, "predicate": "l_shipdate <= c AND l_discount >= 0.05 AND l_quantity < 24", "result": out}))

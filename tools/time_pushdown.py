#!/usr/bin/env python3
"""Where the wall time of the Q1 pushdown plan goes (SF100): every operator timed with a drain of the library stream behind it,
next to the plan's own wall time (tools/gpu_run.sh <tag> py:tools/time_pushdown.py)."""
import os
import sys
import time

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch  # noqa: E402

from databend_amd import _lib as L  # noqa: E402
from databend_amd import device as D, tpch  # noqa: E402
from databend_amd._lib import check, lib  # noqa: E402

D.init(0)
li = tpch.LineitemTorch(int(sys.argv[1]) if len(sys.argv) > 1 else 600_037_902, seed=2, torch=torch)
n = li.n
sync = lambda: check(lib().dbhip_stream_sync(None))  # noqa: E731
for _ in range(2):
    tpch.q1_operator_pushdown(li)
sync()
ts = []
for _ in range(3):
    sync(); t0 = time.perf_counter(); tpch.q1_operator_pushdown(li); sync(); ts.append((time.perf_counter() - t0) * 1e3)
print("plan wall ms", ts)


def timed(name, fn):
    sync(); t0 = time.perf_counter(); r = fn(); sync()
    print(f"  {name:34s} {(time.perf_counter() - t0) * 1e3:8.3f} ms")
    return r


for rep in range(2):
    print("rep", rep)
    g = timed("GroupBy.q1()", D.GroupBy.q1)
    pred = timed("cmp", lambda: D.cmp(L.CMP_LTE, li.ship, D.Column.scalar(tpch.Q1_CUTOFF, L.T_DATE), n))
    one = D.Column.scalar(1, L.T_U8)
    one_minus = timed("1 - disc", lambda: D.decimal_arith(L.OP_MINUS, one, li.disc, n))
    disc_price = timed("price * (1 - disc)", lambda: D.decimal_arith(L.OP_MULTIPLY, li.price, one_minus, n))
    one_plus = timed("1 + tax", lambda: D.decimal_arith(L.OP_PLUS, one, li.tax, n))
    charge = timed("disc_price * (1 + tax)", lambda: D.decimal_arith(L.OP_MULTIPLY, disc_price, one_plus, n))
    timed("add_block(filter)", lambda: g.add_block([li.rf, li.ls], [li.qty, li.price, disc_price, charge, li.disc, None], n, filter=pred))

    def drop():
        global one_minus, disc_price, one_plus, charge, pred, g
        del one_minus, disc_price, one_plus, charge, pred, g
    timed("free the temporaries", drop)

print("literal plan, operator by operator")
for rep in range(2):
    print("rep", rep)
    g = D.GroupBy.q1()
    pred = timed("cmp", lambda: D.cmp(L.CMP_LTE, li.ship, D.Column.scalar(tpch.Q1_CUTOFF, L.T_DATE)))
    sel, k = timed("filter_select", lambda: D.filter_select(pred))
    qty = timed("take qty (8 B)", lambda: D.take(li.qty, sel, k))
    price = timed("take price", lambda: D.take(li.price, sel, k))
    disc = timed("take disc", lambda: D.take(li.disc, sel, k))
    tax = timed("take tax", lambda: D.take(li.tax, sel, k))
    rf = timed("take returnflag (16 B views)", lambda: D.take(li.rf, sel, k))
    ls = timed("take linestatus", lambda: D.take(li.ls, sel, k))
    one = D.Column.scalar(1, L.T_U8)
    one_minus = timed("1 - disc", lambda: D.decimal_arith(L.OP_MINUS, one, disc, k))
    disc_price = timed("price * (1 - disc)", lambda: D.decimal_arith(L.OP_MULTIPLY, price, one_minus, k))
    one_plus = timed("1 + tax", lambda: D.decimal_arith(L.OP_PLUS, one, tax, k))
    charge = timed("disc_price * (1 + tax)", lambda: D.decimal_arith(L.OP_MULTIPLY, disc_price, one_plus, k))
    timed("add_block", lambda: g.add_block([rf, ls], [qty, price, disc_price, charge, disc, None], k))
ts = []
for _ in range(3):
    sync(); t0 = time.perf_counter(); tpch.q1_operator_at_a_time(li); sync(); ts.append((time.perf_counter() - t0) * 1e3)
print("literal plan wall ms", ts)

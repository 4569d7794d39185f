"""Synthetic TPC-H lineitem and the two GPU Q1 plans.

Generator: numpy Generator(PCG64(seed)); column definitions follow
benchmark/tpch/create.sql:12-29 and dbgen's value rules:
  l_quantity      Decimal(15,2) as i64  = U{1..50} * 100
  l_extendedprice Decimal(15,2) as i64  = U[90000, 10494950] (cents)
  l_discount      Decimal(15,2) as i64  = U{0..10}
  l_tax           Decimal(15,2) as i64  = U{0..8}
  l_shipdate      Date32                = U[1992-01-02, 1998-12-01]
  l_returnflag / l_linestatus  16-byte inline views, correlated with the dates
                  like dbgen (receipt = ship + U[1,30]; currentdate 1995-06-17):
                  receipt <= current -> flag in {A,R} else N; ship > current -> O else F
                  (so the rare (N,F) group exists).
Q1 predicate: l_shipdate <= 1998-12-01 - 90 days = 1998-09-02
(benchmark/tpch/queries/01.sql:1-16).
"""
import datetime as _dt

import numpy as np

from . import _lib as L
from . import device as D

EPOCH = _dt.date(1970, 1, 1)


def days(y, m, d):
    return (_dt.date(y, m, d) - EPOCH).days


SHIP_LO, SHIP_HI = days(1992, 1, 2), days(1998, 12, 1)
CURRENT = days(1995, 6, 17)
Q1_CUTOFF = days(1998, 12, 1) - 90  # 1998-09-02
SF1_ROWS = 6_001_215


def rows_for_sf(sf):
    return {1: 6_001_215, 10: 59_986_052, 100: 600_037_902}.get(sf, int(SF1_ROWS * sf))


def _views_from_chars(chars):
    v = np.zeros((len(chars), 16), dtype=np.uint8)
    v[:, 0] = 1  # len = 1 (little endian u32)
    v[:, 4] = chars
    return v


def gen_lineitem(n, seed=2):
    rng = np.random.Generator(np.random.PCG64(seed))
    qty = rng.integers(1, 51, n, dtype=np.int64) * 100
    price = rng.integers(90000, 10494951, n, dtype=np.int64)
    disc = rng.integers(0, 11, n, dtype=np.int64)
    tax = rng.integers(0, 9, n, dtype=np.int64)
    ship = rng.integers(SHIP_LO, SHIP_HI + 1, n, dtype=np.int32)
    receipt = ship + rng.integers(1, 31, n, dtype=np.int32)
    ar = np.where(rng.integers(0, 2, n, dtype=np.uint8) == 0, ord("A"), ord("R")).astype(np.uint8)
    rf = np.where(receipt <= CURRENT, ar, ord("N")).astype(np.uint8)
    ls = np.where(ship > CURRENT, ord("O"), ord("F")).astype(np.uint8)
    return {
        "l_quantity": qty, "l_extendedprice": price, "l_discount": disc, "l_tax": tax,
        "l_returnflag": _views_from_chars(rf), "l_linestatus": _views_from_chars(ls), "l_shipdate": ship,
    }


class LineitemDevice:
    """lineitem columns resident in HBM (68 B/row)."""

    def __init__(self, host):
        self.n = len(host["l_quantity"])
        dec = dict(precision=15, scale=2)
        self.qty = D.Column.from_numpy(host["l_quantity"], L.T_DEC64, **dec)
        self.price = D.Column.from_numpy(host["l_extendedprice"], L.T_DEC64, **dec)
        self.disc = D.Column.from_numpy(host["l_discount"], L.T_DEC64, **dec)
        self.tax = D.Column.from_numpy(host["l_tax"], L.T_DEC64, **dec)
        self.rf = D.Column.from_views(host["l_returnflag"])
        self.ls = D.Column.from_views(host["l_linestatus"])
        self.ship = D.Column.from_numpy(host["l_shipdate"], L.T_DATE)


def q1_fused(li, g=None, cutoff=Q1_CUTOFF):
    """One fused kernel + merge (dbhip_q1_fused)."""
    g = g or D.GroupBy.q1()
    D.q1_fused(g, li.qty, li.price, li.disc, li.tax, li.rf, li.ls, li.ship, cutoff)
    return g


def q1_operator_at_a_time(li, g=None, cutoff=Q1_CUTOFF):
    """The reference's plan shape, one kernel per operator / call node:
    TransformFilter -> take -> decimal maps -> TransformPartialAggregate."""
    g = g or D.GroupBy.q1()
    pred = D.cmp(L.CMP_LTE, li.ship, D.Column.scalar(cutoff, L.T_DATE))
    sel, k = D.filter_select(pred)
    qty, price, disc, tax = (D.take(c, sel, k) for c in (li.qty, li.price, li.disc, li.tax))
    rf, ls = D.take(li.rf, sel, k), D.take(li.ls, sel, k)
    one = D.Column.scalar(1, L.T_U8)
    one_minus = D.decimal_arith(L.OP_MINUS, one, disc, k)            # Decimal(16,2)
    disc_price = D.decimal_arith(L.OP_MULTIPLY, price, one_minus, k)  # Decimal(31,4)
    one_plus = D.decimal_arith(L.OP_PLUS, one, tax, k)                # Decimal(16,2)
    charge = D.decimal_arith(L.OP_MULTIPLY, disc_price, one_plus, k)  # Decimal(38,6)
    g.add_block([rf, ls], [qty, price, disc_price, charge, disc, None], k)
    return g


def q1_rows(g):
    """-> {(returnflag, linestatus): dict} from a Q1 group-by table."""
    out = {}
    for rf, ls, sq, sp, sdp, sch, sd, cnt in g.result():
        out[(rf, ls)] = dict(sum_qty=sq, sum_base_price=sp, sum_disc_price=sdp, sum_charge=sch, sum_disc=sd, count=cnt)
    return out


def q1_finalize(rows):
    """Post-aggregate projection of Q1 on the (<= a handful of) group rows, on the GPU:
    avg(x) was rewritten by the planner to sum(x) / if(count(x)=0, 1, count(x))
    (aggregate_rewriter.rs:62-66,176-177): Decimal(18,2) / UInt64 -> Decimal(24,8)
    with round-half-away (decimal/arithmetic.rs:212-243), then ORDER BY the two keys."""
    keys = sorted(rows)
    if not keys:
        return []
    cnt = np.array([rows[k]["count"] for k in keys], dtype=np.uint64)
    div = D.Column.from_numpy(np.where(cnt == 0, 1, cnt).astype(np.uint64), L.T_U64)
    outs = {}
    for name in ("sum_qty", "sum_base_price", "sum_disc"):
        s = D.Column.from_numpy(np.array([rows[k][name] for k in keys], dtype=np.int64), L.T_DEC64, precision=18, scale=2)
        outs[name] = D.decimal_arith(L.OP_DIVIDE, s, div, len(keys)).to_numpy()
    res = []
    for i, k in enumerate(keys):
        r = rows[k]
        res.append((k[0], k[1], r["sum_qty"], r["sum_base_price"], r["sum_disc_price"], r["sum_charge"],
                    outs["sum_qty"][i], outs["sum_base_price"][i], outs["sum_disc"][i], r["count"]))
    return res

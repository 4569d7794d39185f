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
import ctypes as C
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


class LineitemTorch:
    """lineitem generated ON THE DEVICE (torch is plumbing: an SF100 shard is 600 M rows / 40.8 GB — numpy generation plus
    a pageable upload would take minutes). Same column definitions and the same dbgen correlation as gen_lineitem, drawn
    from torch's Philox generator (seeded per 2^26-row chunk: the data of a row range does not depend on how many rows
    the shard has); columns are handed to the library as raw device pointers. `host(lo, hi)` copies a row range back in
    gen_lineitem's dict layout (CPU baseline sample / full-size check in chunks)."""

    CHUNK = 1 << 26

    def __init__(self, n, seed=2, torch=None, device=None, row0=0):
        import torch as _torch
        torch = torch or _torch
        dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.n, self.torch = int(n), torch
        i64, i32 = torch.int64, torch.int32
        self.t_qty, self.t_price = torch.empty(n, dtype=i64, device=dev), torch.empty(n, dtype=i64, device=dev)
        self.t_disc, self.t_tax = torch.empty(n, dtype=i64, device=dev), torch.empty(n, dtype=i64, device=dev)
        self.t_ship = torch.empty(n, dtype=i32, device=dev)
        self.t_rf, self.t_ls = torch.zeros((n, 2), dtype=i64, device=dev), torch.zeros((n, 2), dtype=i64, device=dev)
        g = torch.Generator(device=dev)
        first = row0 // self.CHUNK
        for c in range(first, (row0 + n + self.CHUNK - 1) // self.CHUNK):
            # chunk c covers GLOBAL rows [c * CHUNK, (c + 1) * CHUNK); a shard takes the part that falls into its range
            g.manual_seed(seed * 1_000_003 + c)
            m = self.CHUNK
            qty = torch.randint(1, 51, (m,), generator=g, device=dev, dtype=i64) * 100
            price = torch.randint(90000, 10494951, (m,), generator=g, device=dev, dtype=i64)
            disc = torch.randint(0, 11, (m,), generator=g, device=dev, dtype=i64)
            tax = torch.randint(0, 9, (m,), generator=g, device=dev, dtype=i64)
            ship = torch.randint(SHIP_LO, SHIP_HI + 1, (m,), generator=g, device=dev, dtype=i32)
            receipt = ship + torch.randint(1, 31, (m,), generator=g, device=dev, dtype=i32)
            ar = torch.where(torch.randint(0, 2, (m,), generator=g, device=dev, dtype=i32) == 0, ord("A"), ord("R"))
            rf = torch.where(receipt <= CURRENT, ar, ord("N")).to(i64)
            ls = torch.where(ship > CURRENT, ord("O"), ord("F")).to(i64)
            g0 = c * self.CHUNK
            lo, hi = max(g0, row0), min(g0 + m, row0 + n)
            a, b, d0, d1 = lo - g0, hi - g0, lo - row0, hi - row0
            self.t_qty[d0:d1], self.t_price[d0:d1], self.t_disc[d0:d1], self.t_tax[d0:d1] = qty[a:b], price[a:b], disc[a:b], tax[a:b]
            self.t_ship[d0:d1] = ship[a:b]
            # 16-byte inline view {len = 1, 'X', 0...} as two little-endian u64 words
            self.t_rf[d0:d1, 0] = (rf[a:b] << 32) | 1
            self.t_ls[d0:d1, 0] = (ls[a:b] << 32) | 1
            del qty, price, disc, tax, ship, receipt, ar, rf, ls
        if dev.type == "cuda":
            torch.cuda.synchronize()
        dec = dict(precision=15, scale=2)
        B = D.BorrowedBuffer.of_tensor
        self.qty, self.price = D.Column(L.T_DEC64, n, B(self.t_qty), **dec), D.Column(L.T_DEC64, n, B(self.t_price), **dec)
        self.disc, self.tax = D.Column(L.T_DEC64, n, B(self.t_disc), **dec), D.Column(L.T_DEC64, n, B(self.t_tax), **dec)
        self.rf, self.ls = D.Column(L.T_STRING, n, B(self.t_rf)), D.Column(L.T_STRING, n, B(self.t_ls))
        self.ship = D.Column(L.T_DATE, n, B(self.t_ship))

    def host(self, lo=0, hi=None):
        hi = self.n if hi is None else hi
        f = lambda t: t[lo:hi].cpu().numpy()  # noqa: E731
        return {"l_quantity": f(self.t_qty), "l_extendedprice": f(self.t_price), "l_discount": f(self.t_disc), "l_tax": f(self.t_tax),
                "l_returnflag": f(self.t_rf).view(np.uint8).reshape(-1, 16), "l_linestatus": f(self.t_ls).view(np.uint8).reshape(-1, 16),
                "l_shipdate": f(self.t_ship)}

    def slice(self, lo, hi):
        """zero-copy row range [lo, hi) as a lineitem with the LineitemDevice surface (lo must keep 16-byte alignment)"""
        assert lo % 4 == 0
        class _S:
            pass
        s = _S()
        s.n = hi - lo
        for name in ("qty", "price", "disc", "tax", "rf", "ls", "ship"):
            c = getattr(self, name)
            es = D.ELEM_SIZE[c.dtype]
            setattr(s, name, D.Column(c.dtype, hi - lo, D.BorrowedBuffer(c.data.ptr + lo * es, (hi - lo) * es, keep=self), precision=c.precision, scale=c.scale))
        return s


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


def q1_operator_pushdown(li, g=None, cutoff=Q1_CUTOFF):
    """The same operators with the filter pushed down into the aggregate: the predicate stays a Bitmap, the decimal
    maps and the partial aggregation read the UNFILTERED columns (dbhip_groupby_add_block_filtered) — no selection
    vector, no take of six columns. Same result: rows that fail the predicate never reach a state, and the maps raise
    no row errors on Q1's value ranges (a binding that cannot prove that keeps the literal plan)."""
    g = g or D.GroupBy.q1()
    n = li.n
    pred = D.cmp(L.CMP_LTE, li.ship, D.Column.scalar(cutoff, L.T_DATE), n)
    one = D.Column.scalar(1, L.T_U8)
    one_minus = D.decimal_arith(L.OP_MINUS, one, li.disc, n)            # Decimal(16,2)
    disc_price = D.decimal_arith(L.OP_MULTIPLY, li.price, one_minus, n)  # Decimal(31,4)
    one_plus = D.decimal_arith(L.OP_PLUS, one, li.tax, n)                # Decimal(16,2)
    charge = D.decimal_arith(L.OP_MULTIPLY, disc_price, one_plus, n)     # Decimal(38,6)
    g.add_block([li.rf, li.ls], [li.qty, li.price, disc_price, charge, li.disc, None], n, filter=pred)
    return g


def q1_program(li, cutoff=Q1_CUTOFF):
    """Q1's filter predicate and decimal maps flattened into ONE register program (what Evaluator::run walks node by node,
    block_operator.rs:42-85) -> (program, key columns, aggregate argument registers, filter register): the arguments of
    dbhip_groupby_add_block_program / dbhip_groupby_prepare_program. Built once per pipeline, launched per block."""
    p = D.ExprProgram([li.ship, li.qty, li.price, li.disc, li.tax])
    ship, cut = p.load(0), p.const(cutoff, L.T_DATE)
    f = p.cmp(L.EX_LTE, ship, cut)                                   # l_shipdate <= cutoff          (filter root, first)
    qty, price, disc, tax = p.load(1), p.load(2), p.load(3), p.load(4)
    one = p.const(1, L.T_U8)
    one_minus = p.arith(L.EX_MINUS, one, disc, keep=(one, disc))       # 1 - l_discount               Decimal(16,2)
    disc_price = p.arith(L.EX_MULTIPLY, price, one_minus, keep=(price,))  # l_extendedprice * (..)     Decimal(31,4)
    one_plus = p.arith(L.EX_PLUS, one, tax)                           # 1 + l_tax                    Decimal(16,2)
    charge = p.arith(L.EX_MULTIPLY, disc_price, one_plus, keep=(disc_price,))  # (..) * (..)          Decimal(38,6)
    return p, [li.rf, li.ls], [qty, price, disc_price, charge, disc, None], f


def q1_fused_program(li, g=None, cutoff=Q1_CUTOFF, prepare=False, plan=None, stream=None):
    """The whole Q1 pipeline as ONE generic launch (dbhip_groupby_add_block_program): the binding flattens the filter
    predicate and the three decimal maps into a register program and the fused filter -> map -> partial-aggregate kernel runs
    it — interpreted, or through the kernel that the library specialises for this program at run time (hiprtc; `prepare=True` =
    dbhip_groupby_prepare_program, the PREPARE of the pipeline: waits for that kernel instead of launching). No query-specific
    code on the device. `plan` = a q1_program() result to launch again (a pipeline builds its program once)."""
    g = g or D.GroupBy.q1()
    p, keys, regs, f = plan or q1_program(li, cutoff)
    if prepare:
        g.prepare_program(keys, p, regs, filter_reg=f)
        return g
    g.add_block_program(keys, p, regs, li.n, filter_reg=f, stream=stream)
    return g


def q1_rescale_program(li, cutoff=Q1_CUTOFF):
    """A Q1-like pipeline whose scales FORCE a rescale (VERDICT r03 missing #5): l_extendedprice and l_discount taken as Decimal(15,8)
    — their product is a ROUNDING multiply (scale 16 -> 12: a division by 10^4 per row, decimal/src/arithmetic.rs:212-243) — and
    sum(l_quantity / l_extendedprice), a decimal divide. -> (program, keys, argument registers, filter register, aggregate list)."""
    price8 = D.Column(L.T_DEC64, li.n, li.price.data, None, 15, 8)
    disc8 = D.Column(L.T_DEC64, li.n, li.disc.data, None, 15, 8)
    p = D.ExprProgram([li.ship, li.qty, price8, disc8])
    f = p.cmp(L.EX_LTE, p.load(0), p.const(cutoff, L.T_DATE))
    qty, price, disc = p.load(1), p.load(2), p.load(3)
    prod = p.arith(L.EX_MULTIPLY, price, disc, keep=(price,))
    quo = p.arith(L.EX_DIVIDE, qty, price, keep=(qty,))
    aggs = [(L.AGG_SUM, L.T_DEC64, 15, 2, 0), (L.AGG_SUM, p.types[prod], p.size[prod][0], p.size[prod][1], 0),
            (L.AGG_SUM, p.types[quo], p.size[quo][0], p.size[quo][1], 0), (L.AGG_COUNT, 0, 0, 0, 0)]
    return p, [li.rf, li.ls], [qty, prod, quo, None], f, aggs


def q1_rescale_fused(li, g=None, prepare=False, plan=None, stream=None):
    p, keys, regs, f, aggs = plan or q1_rescale_program(li)
    g = g or D.GroupBy([L.T_STRING, L.T_STRING], aggs, [0, 0])
    if prepare:
        g.prepare_program(keys, p, regs, filter_reg=f)
        return g
    g.add_block_program(keys, p, regs, li.n, filter_reg=f, stream=stream)
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


# ---------------------------------------------------------------------------------------------
# TPC-H Q3 (BASELINE.json configs[2]; benchmark/tpch/queries/03.sql:1-18)
# ---------------------------------------------------------------------------------------------
SEGMENTS = [b"AUTOMOBILE", b"BUILDING", b"FURNITURE", b"MACHINERY", b"HOUSEHOLD"]
Q3_SEGMENT = "BUILDING"
Q3_DATE = days(1995, 3, 15)
ORDER_LO, ORDER_HI = days(1992, 1, 1), days(1998, 8, 2)


def _views_from_short_strings(table, codes):
    """16-byte inline views for strings of <= 12 bytes picked by `codes` from `table`."""
    tv = np.zeros((len(table), 16), dtype=np.uint8)
    for i, s in enumerate(table):
        assert len(s) <= 12
        tv[i, 0] = len(s)
        tv[i, 4:4 + len(s)] = np.frombuffer(s, dtype=np.uint8)
    return tv[codes]


def gen_q3(sf, seed=3):
    """Synthetic customer / orders / lineitem for Q3, numpy PCG64(seed).

    customer: c_custkey dense 1..Nc, c_mktsegment uniform over 5 segments (16-B inline views)
    orders:   o_orderkey sparse like dbgen (8 of every 32 keys), o_custkey uniform over customers
              with key % 3 != 0, o_orderdate uniform 1992-01-01..1998-08-02, o_shippriority = 0
    lineitem: 1..7 lines per order (in order-key order), price/discount as in Q1,
              l_shipdate = o_orderdate + U[1,121]
    """
    rng = np.random.Generator(np.random.PCG64(seed))
    nc, no = max(int(150_000 * sf), 5), max(int(1_500_000 * sf), 8)
    c_custkey = np.arange(1, nc + 1, dtype=np.int64)
    c_seg = _views_from_short_strings(SEGMENTS, rng.integers(0, 5, nc))
    i = np.arange(no, dtype=np.int64)
    o_orderkey = (i // 8) * 32 + (i % 8) + 1
    ck = rng.integers(1, nc + 1, no, dtype=np.int64)
    bad = ck % 3 == 0  # dbgen never assigns orders to every third customer
    ck[bad] = np.where(ck[bad] + 1 > nc, 1, ck[bad] + 1)
    o_orderdate = rng.integers(ORDER_LO, ORDER_HI + 1, no, dtype=np.int32)
    o_shipprio = np.zeros(no, dtype=np.int32)
    lines = rng.integers(1, 8, no)
    l_orderkey = np.repeat(o_orderkey, lines)
    nl = len(l_orderkey)
    l_price = rng.integers(90000, 10494951, nl, dtype=np.int64)
    l_disc = rng.integers(0, 11, nl, dtype=np.int64)
    l_ship = (np.repeat(o_orderdate, lines) + rng.integers(1, 122, nl, dtype=np.int32)).astype(np.int32)
    return {
        "customer": {"c_custkey": c_custkey, "c_mktsegment": c_seg},
        "orders": {"o_orderkey": o_orderkey, "o_custkey": ck, "o_orderdate": o_orderdate, "o_shippriority": o_shipprio},
        "lineitem": {"l_orderkey": l_orderkey, "l_extendedprice": l_price, "l_discount": l_disc, "l_shipdate": l_ship},
    }


class Q3Device:
    """The three Q3 tables resident in HBM (customer 24 B/row, orders 24 B/row, lineitem 28 B/row)."""

    def __init__(self, host):
        c, o, li = host["customer"], host["orders"], host["lineitem"]
        dec = dict(precision=15, scale=2)
        self.c_custkey = D.Column.from_numpy(c["c_custkey"], L.T_I64)
        self.c_mktsegment = D.Column.from_views(c["c_mktsegment"])
        self.o_orderkey = D.Column.from_numpy(o["o_orderkey"], L.T_I64)
        self.o_custkey = D.Column.from_numpy(o["o_custkey"], L.T_I64)
        self.o_orderdate = D.Column.from_numpy(o["o_orderdate"], L.T_DATE)
        self.o_shippriority = D.Column.from_numpy(o["o_shippriority"], L.T_I32)
        self.l_orderkey = D.Column.from_numpy(li["l_orderkey"], L.T_I64)
        self.l_extendedprice = D.Column.from_numpy(li["l_extendedprice"], L.T_DEC64, **dec)
        self.l_discount = D.Column.from_numpy(li["l_discount"], L.T_DEC64, **dec)
        self.l_shipdate = D.Column.from_numpy(li["l_shipdate"], L.T_DATE)
        self.n_customer, self.n_orders, self.n_lineitem = self.c_custkey.n, self.o_orderkey.n, self.l_orderkey.n


Q3_AGGS = [(L.AGG_SUM, L.T_DEC128, 31, 4, 0)]
Q3_KEYS = [L.T_I64, L.T_DATE, L.T_I32]


def q3_operator_at_a_time(t, segment=Q3_SEGMENT, date=Q3_DATE, limit=10, stats=None, bitmap_probe=True, block_take=True):
    """Q3 in the reference's plan shape, one C-ABI call per operator / expression node:

      customer -> TransformFilter(c_mktsegment = seg) -> Join#1 build (c_custkey)
      orders   -> filter predicate(o_orderdate < date) -> Join#1 probe (o_custkey, predicate as validity) -> Join#2 build (o_orderkey)
      lineitem -> filter predicate(l_shipdate > date)  -> Join#2 probe (l_orderkey, predicate as validity)
               -> take probe/build columns (inner_join.rs:248-268)
               -> 1 - l_discount ; l_extendedprice * (..)   (decimal/arithmetic.rs:190-316)
               -> TransformPartialAggregate / Final on (l_orderkey, o_orderdate, o_shippriority), sum -> Decimal128(38,4)
               -> sort revenue DESC, o_orderdate ASC LIMIT 10 (kernels/sort.rs:91-113)
    Returns [(l_orderkey, revenue, o_orderdate, o_shippriority)] in output order.
    """
    # customer
    seg = D.Column.from_views(_views_from_short_strings([segment.encode()], np.zeros(1, dtype=np.int64)))
    seg.is_scalar = True
    csel, kc = D.filter_select(D.cmp(L.CMP_EQ, t.c_mktsegment, seg, t.n_customer))
    j1 = D.HashJoin(max(kc, 16))
    j1.add_block(D.take(t.c_custkey, csel, kc))
    j1.final_build()
    # orders. Two plans for filter -> probe (measured side by side by tools/bench_q3.py):
    #   bitmap_probe (default, 21.1 ms at SF100): the filter's Bitmap rides into the probe as the key column's validity
    #     (rows that fail the predicate are not probed), nothing is materialised and the pairs carry ORIGINAL row ids;
    #   materialise (23.6 ms): TransformFilter's selection, take the surviving columns, probe the compacted keys.
    opred = D.cmp(L.CMP_LT, t.o_orderdate, D.Column.scalar(date, L.T_DATE), t.n_orders)
    if bitmap_probe:
        ko = D.bitmap_count(opred, t.n_orders) if stats is not None else 0
        pp, _pb, kj = j1.probe_block_device(D.Column(t.o_custkey.dtype, t.n_orders, t.o_custkey.data, validity=opred.data))
        # (block_take: dbhip_take_block, one launch per selection instead of one dbhip_take per column. r03, SF100, same box: 9.66 ms
        # against 9.81 ms — the first version of the kernel, at 238 VGPRs, had lost 0.2 ms instead)
        take3 = D.take_block if block_take else (lambda cols, sel, k: [D.take(c, sel, k) for c in cols])
        b_ok, b_od, b_sp = take3([t.o_orderkey, t.o_orderdate, t.o_shippriority], pp, kj)
    else:
        osel, ko = D.filter_select(opred)
        f_ok, f_ck, f_od, f_sp = (D.take(c, osel, ko) for c in (t.o_orderkey, t.o_custkey, t.o_orderdate, t.o_shippriority))
        pp, _pb, kj = j1.probe_block_device(f_ck)
        b_ok, b_od, b_sp = (D.take(c, pp, kj) for c in (f_ok, f_od, f_sp))
    j2 = D.HashJoin(max(kj, 16))
    j2.add_block(b_ok)
    j2.final_build()
    # lineitem
    lpred = D.cmp(L.CMP_GT, t.l_shipdate, D.Column.scalar(date, L.T_DATE), t.n_lineitem)
    if bitmap_probe:
        kl = D.bitmap_count(lpred, t.n_lineitem) if stats is not None else 0
        lp, lb, kp = j2.probe_block_device(D.Column(t.l_orderkey.dtype, t.n_lineitem, t.l_orderkey.data, validity=lpred.data))
        j_ok, j_price, j_disc = take3([t.l_orderkey, t.l_extendedprice, t.l_discount], lp, kp)
    else:
        lsel, kl = D.filter_select(lpred)
        f_lok, f_price, f_disc = (D.take(c, lsel, kl) for c in (t.l_orderkey, t.l_extendedprice, t.l_discount))
        lp, lb, kp = j2.probe_block_device(f_lok)
        j_ok, j_price, j_disc = (D.take(c, lp, kp) for c in (f_lok, f_price, f_disc))
    j_od, j_sp = (D.take_block if block_take else (lambda cols, sel, k: [D.take(c, sel, k) for c in cols]))([b_od, b_sp], lb, kp)
    one_minus = D.decimal_arith(L.OP_MINUS, D.Column.scalar(1, L.T_U8), j_disc, kp)   # Decimal(16,2)
    revenue = D.decimal_arith(L.OP_MULTIPLY, j_price, one_minus, kp)                   # Decimal(31,4)
    g = D.GroupBy(Q3_KEYS, Q3_AGGS, capacity=max(1024, 2 * kp))
    g.add_block([j_ok, j_od, j_sp], [revenue], kp)
    if stats is not None:
        stats.update(customers_kept=kc, orders_kept=ko, orders_joined=kj, lineitem_kept=kl, lineitem_pairs=kp, groups=g.num_groups())
    return q3_top_rows(g, limit)


def q3_top_rows(g, limit):
    """ORDER BY revenue DESC, o_orderdate LIMIT over the table's groups -> [(l_orderkey, revenue, o_orderdate, o_shippriority)]"""
    k_ok, k_od, k_sp, rev = g.result_columns()
    ng = k_ok.n
    perm = D.sort_perm([rev, k_od], desc=[1, 0], limit=limit) if ng else np.zeros(0, np.uint32)
    m = len(perm)
    psel = D.DeviceBuffer.from_numpy(perm.astype(np.uint32))
    ok, od, sp = (D.take(c, psel, m).to_numpy() for c in (k_ok, k_od, k_sp))
    rv = D.take(rev, psel, m).to_numpy()
    return [(int(ok[i]), int(rv[i]), int(od[i]), int(sp[i])) for i in range(m)]


class _Borrowed:
    """a torch tensor's storage as a column buffer (the tensor stays alive with the column)"""

    def __init__(self, t):
        self.t, self.ptr, self.nbytes = t, t.data_ptr(), t.numel() * t.element_size()


class Q3DeviceOps:
    """The single-node operators of the distributed broadcast-join plan (databend_amd.dist.q3_broadcast_join) over the C-ABI:
    this rank's shard is a Q3Device-like object (columns resident in HBM); the columns that cross ranks are CUDA tensors
    (torch.distributed moves them over RCCL), everything else stays inside the library."""

    def __init__(self, torch, segment=Q3_SEGMENT, date=Q3_DATE):
        self.torch, self.segment, self.date = torch, segment, date
        self.keep = []

    def _tensor(self, col, dtype):
        t = self.torch.empty(col.n, dtype=dtype, device="cuda")
        if col.n:
            L.check(L.lib().dbhip_memcpy_d2d(C.c_void_p(t.data_ptr()), C.c_void_p(col.data.ptr), C.c_size_t(t.numel() * t.element_size()), None))
        L.check(L.lib().dbhip_stream_sync(None))
        return t

    def _column(self, t, dtype):
        self.torch.cuda.current_stream().synchronize()   # the collective that produced it has finished
        return D.Column(dtype, int(t.shape[0]), _Borrowed(t.contiguous()))

    def filter_customers(self, t):
        seg = D.Column.from_views(_views_from_short_strings([self.segment.encode()], np.zeros(1, dtype=np.int64)))
        seg.is_scalar = True
        csel, kc = D.filter_select(D.cmp(L.CMP_EQ, t.c_mktsegment, seg, t.n_customer))
        return [self._tensor(D.take(t.c_custkey, csel, kc), self.torch.int64)]

    def join_orders(self, all_custkeys, t):
        j1 = D.HashJoin(max(int(all_custkeys.shape[0]), 16))
        j1.add_block(self._column(all_custkeys, L.T_I64))
        j1.final_build()
        opred = D.cmp(L.CMP_LT, t.o_orderdate, D.Column.scalar(self.date, L.T_DATE), t.n_orders)
        pp, _pb, kj = j1.probe_block_device(D.Column(t.o_custkey.dtype, t.n_orders, t.o_custkey.data, validity=opred.data))
        return [self._tensor(D.take(c, pp, kj), dt) for c, dt in ((t.o_orderkey, self.torch.int64), (t.o_orderdate, self.torch.int32),
                                                                  (t.o_shippriority, self.torch.int32))]

    def aggregate_lineitem(self, okey, odate, oprio, t):
        b_ok, b_od, b_sp = self._column(okey, L.T_I64), self._column(odate, L.T_DATE), self._column(oprio, L.T_I32)
        j2 = D.HashJoin(max(b_ok.n, 16))
        j2.add_block(b_ok)
        j2.final_build()
        lpred = D.cmp(L.CMP_GT, t.l_shipdate, D.Column.scalar(self.date, L.T_DATE), t.n_lineitem)
        lp, lb, kp = j2.probe_block_device(D.Column(t.l_orderkey.dtype, t.n_lineitem, t.l_orderkey.data, validity=lpred.data))
        j_ok, j_price, j_disc = (D.take(c, lp, kp) for c in (t.l_orderkey, t.l_extendedprice, t.l_discount))
        j_od, j_sp = D.take(b_od, lb, kp), D.take(b_sp, lb, kp)
        one_minus = D.decimal_arith(L.OP_MINUS, D.Column.scalar(1, L.T_U8), j_disc, kp)
        revenue = D.decimal_arith(L.OP_MULTIPLY, j_price, one_minus, kp)
        g = D.GroupBy(Q3_KEYS, Q3_AGGS, capacity=max(1024, 2 * kp))
        if kp:
            g.add_block([j_ok, j_od, j_sp], [revenue], kp)
        return g

    def exchange(self, g, dist, device):
        from . import dist as DX
        return DX.exchange_partials_alltoall_variable(g, dist, self.torch, device, lib_sync=lambda: L.check(L.lib().dbhip_stream_sync(None)))

    def top_rows(self, g, limit):
        return q3_top_rows(g, limit)

#!/usr/bin/env python3
"""TPC-H Q3 (BASELINE.json configs[2]) on one MI355X: the operator-at-a-time plan of
databend_amd.tpch.q3_operator_at_a_time over HBM-resident synthetic tables.

    python tools/bench_q3.py [--sf 100] [--reps 3] [--out gpurun_out/q3.json]

Tables are generated ON THE DEVICE with torch (plumbing only, same distributions as
tpch.gen_q3 / SURVEY.md §8d C3 — numpy generation of 600 M rows would take minutes) and handed
to the library as raw device pointers. Unit of work = probe-side (lineitem) rows; streamed
algorithmic bytes = customer 24 B + orders 24 B + lineitem 28 B per row.
Full-size check (the oracle cannot run SF100 in seconds): the query result is compared with
an independent torch statement of Q3 (sorted-key membership instead of hash tables).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch

from databend_amd import _lib as L
from databend_amd import device as D
from databend_amd import tpch
from databend_amd._lib import check, lib


class Borrowed:
    def __init__(self, t):
        self.t = t
        self.ptr = t.data_ptr()
        self.nbytes = t.numel() * t.element_size()


def col(t, dtype, **kw):
    return D.Column(dtype, t.shape[0], Borrowed(t), **kw)


class Q3Torch:
    def __init__(self, sf, seed=3):
        g = torch.Generator(device="cuda")
        g.manual_seed(seed)
        dev = "cuda"
        nc, no = max(int(150_000 * sf), 5), max(int(1_500_000 * sf), 8)
        self.c_custkey = torch.arange(1, nc + 1, dtype=torch.int64, device=dev)
        seg_code = torch.randint(0, 5, (nc,), generator=g, device=dev)
        table = torch.from_numpy(tpch._views_from_short_strings(tpch.SEGMENTS, np.arange(5))).to(dev)
        self.c_seg = table[seg_code].contiguous()
        self.seg_code = seg_code
        i = torch.arange(no, dtype=torch.int64, device=dev)
        self.o_orderkey = (i // 8) * 32 + (i % 8) + 1
        ck = torch.randint(1, nc + 1, (no,), generator=g, device=dev, dtype=torch.int64)
        bad = ck % 3 == 0
        ck[bad] = torch.where(ck[bad] + 1 > nc, torch.ones_like(ck[bad]), ck[bad] + 1)
        self.o_custkey = ck
        self.o_orderdate = torch.randint(tpch.ORDER_LO, tpch.ORDER_HI + 1, (no,), generator=g, device=dev, dtype=torch.int32)
        self.o_shipprio = torch.zeros(no, dtype=torch.int32, device=dev)
        lines = torch.randint(1, 8, (no,), generator=g, device=dev)
        self.l_orderkey = torch.repeat_interleave(self.o_orderkey, lines)
        nl = self.l_orderkey.shape[0]
        self.l_price = torch.randint(90000, 10494951, (nl,), generator=g, device=dev, dtype=torch.int64)
        self.l_disc = torch.randint(0, 11, (nl,), generator=g, device=dev, dtype=torch.int64)
        self.l_ship = (torch.repeat_interleave(self.o_orderdate, lines) + torch.randint(1, 122, (nl,), generator=g, device=dev, dtype=torch.int32)).to(torch.int32)
        del lines, i
        self.nc, self.no, self.nl = nc, no, nl

    def table(self):
        class T:
            pass
        t = T()
        dec = dict(precision=15, scale=2)
        t.c_custkey, t.c_mktsegment = col(self.c_custkey, L.T_I64), col(self.c_seg, L.T_STRING)
        t.o_orderkey, t.o_custkey = col(self.o_orderkey, L.T_I64), col(self.o_custkey, L.T_I64)
        t.o_orderdate, t.o_shippriority = col(self.o_orderdate, L.T_DATE), col(self.o_shipprio, L.T_I32)
        t.l_orderkey, t.l_shipdate = col(self.l_orderkey, L.T_I64), col(self.l_ship, L.T_DATE)
        t.l_extendedprice, t.l_discount = col(self.l_price, L.T_DEC64, **dec), col(self.l_disc, L.T_DEC64, **dec)
        t.n_customer, t.n_orders, t.n_lineitem = self.nc, self.no, self.nl
        return t

    def torch_q3(self, date, limit):
        """independent statement (sorted membership + index_add in exact i64: revenue sums of <= 7 lines fit)"""
        seg = tpch.SEGMENTS.index(tpch.Q3_SEGMENT.encode())
        cust_ok = torch.zeros(self.nc + 2, dtype=torch.bool, device="cuda")
        cust_ok[self.c_custkey[self.seg_code == seg]] = True
        okeep = (self.o_orderdate < date) & cust_ok[self.o_custkey]
        oidx = torch.nonzero(okeep).squeeze(1)                 # orders are sorted by key -> position lookup
        lkeep = self.l_ship > date
        lk = self.l_orderkey[lkeep]
        pos = (lk - 1) // 32 * 8 + (lk - 1) % 32              # inverse of the o_orderkey formula
        hit = okeep[pos]
        pos = pos[hit]
        rev = (self.l_price[lkeep][hit] * (100 - self.l_disc[lkeep][hit]))
        acc = torch.zeros(self.no, dtype=torch.int64, device="cuda")
        acc.index_add_(0, pos, rev)
        has = torch.zeros(self.no, dtype=torch.bool, device="cuda")
        has[pos] = True
        gidx = torch.nonzero(has).squeeze(1)
        r, d, k = acc[gidx], self.o_orderdate[gidx], self.o_orderkey[gidx]
        # ORDER BY revenue DESC, o_orderdate ASC: composite key (revenue < 2^40, date < 2^15)
        comp = (-r) * 65536 + d.to(torch.int64)
        order = torch.argsort(comp)[: (limit or len(comp))]
        return [(int(k[j]), int(r[j]), int(d[j]), 0) for j in order.tolist()], int(gidx.numel()), int(oidx.numel())


def gather_bytes(stats):
    """bytes the takes of the plan move (values read + written, and the 4-byte selection / pair index each take reads):
    c_custkey by the filter's selection; o_orderkey, o_orderdate, o_shippriority by probe #1's pairs; l_orderkey, l_extendedprice,
    l_discount by probe #2's probe rows and the taken o_orderdate, o_shippriority by its build rows. Reported beside the streamed
    bytes (they are not part of them)."""
    kj, kp, kc = stats["orders_joined"], stats["lineitem_pairs"], stats["customers_kept"]
    return kc * (8 + 8 + 4) + kj * ((8 + 4 + 4) * 2 + 3 * 4) + kp * ((8 + 8 + 8) * 2 + 3 * 4 + (4 + 4) * 2 + 2 * 4)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sf", type=float, default=100.0)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--out", default="")
    ap.add_argument("--oracle", action="store_true", help="also run the CPU oracle's Q3 (oracle/liboracle.so, all host threads) on the SAME tables "
                    "copied back to the host and compare stage counts and the top rows (a check, ~1 min at SF100)")
    ap.add_argument("--main-only", action="store_true", help="skip the alternative (materialise) plan: for profiles of the default plan")
    args = ap.parse_args()
    D.init(0)
    t0 = time.perf_counter()
    src = Q3Torch(args.sf)
    torch.cuda.synchronize()
    gen_s = time.perf_counter() - t0
    t = src.table()
    stats = {}
    got = tpch.q3_operator_at_a_time(t, stats=stats)  # warm-up (allocations land in the block cache)
    def timed(**kw):
        out, ts = None, []
        for _ in range(args.reps):
            check(lib().dbhip_stream_sync(None))
            torch.cuda.synchronize()
            c0 = time.perf_counter()
            out = tpch.q3_operator_at_a_time(t, **kw)
            check(lib().dbhip_stream_sync(None))
            ts.append(time.perf_counter() - c0)
        return out, ts

    got, times = timed()
    # the alternative filter->probe plan (filter_select + take, probe the compacted keys), timed in the same process
    if args.main_only:
        got_b, times_b = got, []
    else:
        tpch.q3_operator_at_a_time(t, bitmap_probe=False)
        got_b, times_b = timed(bitmap_probe=False)
    assert [(r[1], r[2]) for r in got_b] == [(r[1], r[2]) for r in got] and sorted(got_b) == sorted(got), "the two plans disagree"
    exp, ngroups, njoined = src.torch_q3(tpch.Q3_DATE, 10)
    ok = [(r[1], r[2]) for r in got] == [(r[1], r[2]) for r in exp] and sorted(got) == sorted(exp) and ngroups == stats["groups"] \
        and njoined == stats["orders_joined"]
    oracle = None
    if args.oracle:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib as O
        h = lambda x: x.cpu().numpy()  # noqa: E731
        host = {"customer": {"c_custkey": h(src.c_custkey), "c_mktsegment": h(src.c_seg)},
                "orders": {"o_orderkey": h(src.o_orderkey), "o_custkey": h(src.o_custkey), "o_orderdate": h(src.o_orderdate),
                           "o_shippriority": h(src.o_shipprio)},
                "lineitem": {"l_orderkey": h(src.l_orderkey), "l_extendedprice": h(src.l_price), "l_discount": h(src.l_disc),
                             "l_shipdate": h(src.l_ship)}}
        so = {}
        c0 = time.perf_counter()
        threads = os.cpu_count() or 8
        exp_o = O.q3_run(host, tpch.Q3_SEGMENT, tpch.Q3_DATE, limit=10, threads=threads, stages=so)
        osec = time.perf_counter() - c0
        same = [(r[1], r[2]) for r in got] == [(r[1], r[2]) for r in exp_o] and sorted(got) == sorted(exp_o)
        stages_same = all(stats[k] == so[k] for k in ("customers_kept", "orders_kept", "orders_joined", "groups"))
        oracle = {"matches": bool(same and stages_same), "seconds": osec, "threads": threads, "stages": so}
        del host
        if not oracle["matches"]:
            print("ORACLE MISMATCH", got, exp_o, stats, so, file=sys.stderr)
            sys.exit(1)
    best = min(times)
    streamed = 24 * src.nc + 24 * src.no + 28 * src.nl
    gather = gather_bytes(stats)
    out = {"workload": f"TPC-H Q3 SF{args.sf:g} operator-at-a-time (customer {src.nc}, orders {src.no}, lineitem {src.nl} rows)",
           "seconds": best, "all_seconds": times, "lineitem_rows_per_s": src.nl / best, "streamed_bytes": streamed,
           "streamed_GBps": streamed / best / 1e9, "frac_of_hbm_peak": streamed / best / 8e12, "gather_bytes": gather,
           "matches_cpu_oracle_on_the_same_tables": oracle, "stages": stats,
           "plan": "predicate Bitmaps as probe-key validity (nothing materialised before the joins)",
           "materialise_plan_seconds": times_b,
           "matches_independent_torch_statement": bool(ok), "top10": got, "generate_seconds": gen_s}
    print(json.dumps(out))
    if args.out:
        os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
        with open(args.out, "w") as f:
            json.dump(out, f, indent=1)
    if not ok:
        print("MISMATCH", got, exp, ngroups, stats, file=sys.stderr)
        sys.exit(1)


if __name__ == "__main__":
    main()

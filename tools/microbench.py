#!/usr/bin/env python3
"""Per-kernel-family microbenchmark on one MI355X (not the driver's bench.py contract).

For every C-ABI family it times the call with HIP events on the library stream
(dbhip_event_*), at working sets well past the 256 MiB Infinity Cache, and reports
ALGORITHMIC bytes (or flops) / time against the HBM (8 TB/s) or FP32-MFMA (157.3 TF)
peak. Inputs are generated on the device with torch (plumbing only) and handed to the
library as raw device pointers.

    python tools/microbench.py [--only name,name] [--out gpurun_out/microbench.json]
"""
import argparse
import ctypes as C
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
from databend_amd._lib import check, lib

HBM, MFMA32 = 8000.0, 157.3


class Borrowed:
    """non-owning view of a torch tensor's storage with the DeviceBuffer surface"""

    def __init__(self, t):
        self.t = t
        self.ptr = t.data_ptr()
        self.nbytes = t.numel() * t.element_size()


def col(t, dtype, **kw):
    return D.Column(dtype, t.shape[0], Borrowed(t), **kw)


def timed(fn, reps=5, warm=2):
    Lb = lib()
    torch.cuda.synchronize()
    for _ in range(warm):
        fn()
    e0, e1 = C.c_void_p(), C.c_void_p()
    check(Lb.dbhip_event_create(C.byref(e0)))
    check(Lb.dbhip_event_create(C.byref(e1)))
    best = 1e30
    tot = 0.0
    for _ in range(reps):
        check(Lb.dbhip_event_record(e0, None))
        fn()
        check(Lb.dbhip_event_record(e1, None))
        check(Lb.dbhip_stream_sync(None))
        ms = C.c_float()
        check(Lb.dbhip_event_elapsed_ms(e0, e1, C.byref(ms)))
        best = min(best, ms.value)
        tot += ms.value
    return tot / reps, best


def report(out, name, unit_count, unit, alg_bytes=None, flops=None, ms=None, note=""):
    avg, best = ms
    r = {"name": name, "ms_avg": round(avg, 4), "ms_best": round(best, 4), unit + "/s": unit_count / (avg * 1e-3), "note": note}
    if alg_bytes is not None:
        gbs = alg_bytes / (avg * 1e-3) / 1e9
        r.update(bound="hbm", alg_bytes=alg_bytes, GBps=round(gbs, 1), frac=round(gbs / HBM, 4))
    if flops is not None:
        tf = flops / (avg * 1e-3) / 1e12
        r.update(bound="mfma", flops=flops, TFLOPs=round(tf, 2), frac=round(tf / MFMA32, 4))
    out.append(r)
    print(json.dumps(r), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    ap.add_argument("--out", default="")
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--vec-nq", default="1,32,64,128,512,2048")
    ap.add_argument("--gb-card", default="4,200,1000,5000,20000,100000,1000000,10000000")
    ap.add_argument("--gbl-card", default="200,20000,1000000")
    args = ap.parse_args()
    only = set(x for x in args.only.split(",") if x)
    want = lambda k: not only or k in only
    torch.cuda.set_device(0)
    D.init(0)
    Lb = lib()
    dev = "cuda"
    g = torch.Generator(device=dev)
    g.manual_seed(7)
    out = []
    N = int(128_000_000 * args.scale)

    def ri(lo, hi, n, dt=torch.int64):
        return torch.randint(lo, hi, (n,), device=dev, dtype=dt, generator=g)

    if want("arith"):
        a, b = ri(-2**31, 2**31, N), ri(-2**31, 2**31, N)
        ca, cb = col(a, L.T_I64), col(b, L.T_I64)
        o = torch.empty(N + 8, dtype=torch.int64, device=dev)
        cca, ccb = ca.c(), cb.c()
        f = lambda: check(Lb.dbhip_arith(L.OP_PLUS, C.byref(cca), C.byref(ccb), C.c_int64(N), L.T_I64, C.c_void_p(o.data_ptr()), None, None, None))
        report(out, "arith plus i64,i64->i64", N, "rows", alg_bytes=24 * N, ms=timed(f))
        a32, b32 = ri(-2**31, 2**31, N, torch.int32), ri(-2**31, 2**31, N, torch.int32)
        c1, c2 = col(a32, L.T_I32).c(), col(b32, L.T_I32).c()
        f = lambda: check(Lb.dbhip_arith(L.OP_MULTIPLY, C.byref(c1), C.byref(c2), C.c_int64(N), L.T_I64, C.c_void_p(o.data_ptr()), None, None, None))
        report(out, "arith multiply i32,i32->i64", N, "rows", alg_bytes=16 * N, ms=timed(f))
        eb = torch.empty(N // 8 + 64, dtype=torch.uint8, device=dev)
        ec = torch.zeros(1, dtype=torch.int64, device=dev)
        bnz = b.clamp(min=1)
        c3 = col(bnz, L.T_I64).c()
        of = torch.empty(N + 8, dtype=torch.float64, device=dev)
        f = lambda: check(Lb.dbhip_arith(L.OP_DIVIDE, C.byref(cca), C.byref(c3), C.c_int64(N), L.T_F64, C.c_void_p(of.data_ptr()), C.c_void_p(eb.data_ptr()), C.c_void_p(ec.data_ptr()), None))
        report(out, "arith divide i64,i64->f64 (+err bitmap)", N, "rows", alg_bytes=24 * N + N // 8, ms=timed(f))
        del a32, b32, of, bnz, eb

    if want("sum3"):
        n1 = 10_000_000
        a, b, c = ri(-2**31, 2**31, N), ri(-2**31, 2**31, N), ri(-2**31, 2**31, N)
        s = torch.zeros(1, dtype=torch.int64, device=dev)
        for n, tag in ((n1, "C1 10M rows (L3-resident)"), (N, f"{N} rows")):
            f = lambda: check(Lb.dbhip_sum_a_plus_b_mul_c_i64(C.c_void_p(a.data_ptr()), C.c_void_p(b.data_ptr()), C.c_void_p(c.data_ptr()), C.c_int64(n), C.c_void_p(s.data_ptr()), None))
            report(out, "sum(a+b*c) i64 fused, " + tag, n, "rows", alg_bytes=24 * n, ms=timed(f))
        # operator-at-a-time: multiply, plus, sum (reference plan shape, 3 kernels, 2 intermediates)
        cb_, cc_, ca_ = col(b, L.T_I64).c(), col(c, L.T_I64).c(), col(a, L.T_I64).c()
        t1 = torch.empty(N + 8, dtype=torch.int64, device=dev)
        t2 = torch.empty(N + 8, dtype=torch.int64, device=dev)
        ct1 = col(t1[:N], L.T_I64).c()
        ct2 = col(t2[:N], L.T_I64).c()

        def plan():
            check(Lb.dbhip_arith(L.OP_MULTIPLY, C.byref(cb_), C.byref(cc_), C.c_int64(N), L.T_I64, C.c_void_p(t1.data_ptr()), None, None, None))
            check(Lb.dbhip_arith(L.OP_PLUS, C.byref(ca_), C.byref(ct1), C.c_int64(N), L.T_I64, C.c_void_p(t2.data_ptr()), None, None, None))
            check(Lb.dbhip_sum(C.byref(ct2), C.c_int64(N), C.c_void_p(s.data_ptr()), None))
        report(out, "sum(a+b*c) operator-at-a-time (3 kernels)", N, "rows", alg_bytes=24 * N, ms=timed(plan), note="moves 56 B/row")
        # the same tree through the fused expression interpreter (dbhip_expr_eval): one launch, 24 B/row
        prog = (L.ExprIns * 5)()
        for k, (op, dst, x, y) in enumerate([(L.EX_LOAD, 0, 0, 0), (L.EX_LOAD, 1, 1, 0), (L.EX_LOAD, 2, 2, 0), (L.EX_MULTIPLY, 1, 1, 2), (L.EX_PLUS, 0, 0, 1)]):
            prog[k].op, prog[k].dst, prog[k].a, prog[k].b, prog[k].type = op, dst, x, y, L.T_I64
        cols3 = (L.Col * 3)(ca_, cb_, cc_)
        fx = lambda: check(Lb.dbhip_expr_eval(prog, 5, cols3, 3, C.c_int64(N), 0, None, None, None, None, C.c_void_p(s.data_ptr()), None))
        report(out, "sum(a+b*c) fused expression program (dbhip_expr_eval, sum only)", N, "rows", alg_bytes=24 * N, ms=timed(fx))
        t3 = torch.empty(N + 8, dtype=torch.int64, device=dev)
        fy = lambda: check(Lb.dbhip_expr_eval(prog, 5, cols3, 3, C.c_int64(N), 0, C.c_void_p(t3.data_ptr()), None, None, None, None, None))
        report(out, "a+b*c fused expression program -> column", N, "rows", alg_bytes=32 * N, ms=timed(fy))
        del t3
        del t1, t2

    if want("decimal"):
        p = ri(90000, 10494951, N)
        d = ri(0, 11, N)
        cp = col(p, L.T_DEC64, precision=15, scale=2).c()
        cd = col(d, L.T_DEC64, precision=16, scale=2).c()
        o = torch.empty(2 * N + 8, dtype=torch.int64, device=dev)
        f = lambda: check(Lb.dbhip_decimal_arith(L.OP_MULTIPLY, C.byref(cp), C.byref(cd), C.c_int64(N), L.T_DEC128, 31, 4, C.c_void_p(o.data_ptr()), None, None, None))
        report(out, "decimal multiply dec64*dec64->dec128(31,4)", N, "rows", alg_bytes=32 * N, ms=timed(f))
        del o

    if want("cmp"):
        N4 = 4 * N
        sd = ri(8000, 10600, N4, torch.int32)
        csd = col(sd, L.T_DATE).c()
        cut = D.Column.scalar(10471, L.T_DATE)
        ccut = cut.c()
        bm = torch.zeros(N4 // 8 + 64, dtype=torch.uint8, device=dev)
        f = lambda: check(Lb.dbhip_cmp(L.CMP_LTE, C.byref(csd), C.byref(ccut), C.c_int64(N4), C.c_void_p(bm.data_ptr()), None))
        report(out, "cmp date<=scalar -> bitmap", N4, "rows", alg_bytes=N4 * 4 + N4 // 8, ms=timed(f))
        f()
        sel = torch.empty(N4 + 64, dtype=torch.int32, device=dev)
        cnt = torch.zeros(1, dtype=torch.int64, device=dev)
        f2 = lambda: check(Lb.dbhip_filter_select(C.c_void_p(bm.data_ptr()), C.c_int64(0), C.c_int64(N4), C.c_void_p(sel.data_ptr()), C.c_void_p(cnt.data_ptr()), None))
        ms = timed(f2)
        k = int(cnt.item())
        report(out, f"filter_select bitmap -> u32 selection ({k / N4:.1%} kept)", N4, "rows", alg_bytes=N4 // 8 + 4 * k, ms=ms)
        src = ri(-2**31, 2**31, N, torch.int64)
        sel2 = torch.sort(torch.randint(0, N, (N,), device=dev, dtype=torch.int32, generator=g))[0]
        o = torch.empty(N + 8, dtype=torch.int64, device=dev)
        f3 = lambda: check(Lb.dbhip_take(C.c_void_p(src.data_ptr()), 8, C.c_void_p(sel2.data_ptr()), C.c_int64(N), C.c_void_p(o.data_ptr()), None))
        report(out, "take 8-byte, ascending selection (filter output shape)", N, "rows", alg_bytes=N * 20, ms=timed(f3))
        selr = torch.randint(0, N, (N,), device=dev, dtype=torch.int32, generator=g)
        f4 = lambda: check(Lb.dbhip_take(C.c_void_p(src.data_ptr()), 8, C.c_void_p(selr.data_ptr()), C.c_int64(N), C.c_void_p(o.data_ptr()), None))
        report(out, "take 8-byte, random selection (join output shape)", N, "rows", alg_bytes=N * 20, ms=timed(f4), note="random 8-B gathers: 64-B sector per row")
        del sd, bm, sel, src, sel2, selr, o

    if want("hash"):
        a = ri(-2**62, 2**62, N)
        ca = col(a, L.T_I64).c()
        o = torch.empty(N, dtype=torch.int64, device=dev)
        arr = (L.Col * 1)(ca)
        f = lambda: check(Lb.dbhip_group_hash(arr, 1, C.c_int64(N), C.c_void_p(o.data_ptr()), None))
        report(out, "group_hash 1 x i64", N, "rows", alg_bytes=16 * N, ms=timed(f))

    if want("groupby"):
        n = int(60_000_000 * args.scale)
        for card in [int(x) for x in args.gb_card.split(',')]:
            keys = ri(0, card, n)
            vals = ri(0, 1000, n)
            gb = D.GroupBy([L.T_I64], [(L.AGG_SUM, L.T_I64, 0, 0, 0), (L.AGG_COUNT, 0, 0, 0, 0)], capacity=max(1024, card * 2))
            kc, vc = col(keys, L.T_I64), col(vals, L.T_I64)
            def f():
                gb.reset()
                gb.add_block([kc], [vc, None], n)
            note = ""
            if card <= 8:
                # a handful of groups = the run-time specialised few-groups kernel, which a cold code cache compiles in the BACKGROUND
                # (the block that asked takes the compact-row kernel): warm up until add_block really goes through it, bounded
                st = (C.c_uint64 * 3)()
                t0 = time.perf_counter()
                used = False
                while not used and time.perf_counter() - t0 < 30.0:
                    Lb.dbhip_fagg_stats(st)
                    before = st[0]
                    f()
                    Lb.dbhip_fagg_stats(st)
                    used = st[0] > before
                    if not used:
                        time.sleep(0.25)
                note = "fagg_jit (run-time specialised few-groups kernel)" if used else "compact-row kernel (no specialised kernel within 30 s)"
            ms = timed(f, reps=3, warm=1)
            report(out, f"groupby add_block i64 key, sum+count, {card} groups", n, "rows", alg_bytes=16 * n, ms=ms, note=note)
            gb.destroy()
            del keys, vals

    if want("gblayouts"):
        # the layouts plans actually produce, beside the one-i64-key sum + count of the sweep above: nullable keys / arguments, three keys
        # (TPC-H Q3's group-by), a 16-byte key, eight aggregates, and TPC-H Q1's group-by WITHOUT its fused program (two String keys, six
        # aggregates, two of them Decimal128 sums). alg_bytes = the bytes of the key and argument columns.
        n = int(60_000_000 * args.scale)
        vbits = torch.randint(0, 256, (n // 8 + 64,), device=dev, dtype=torch.uint8, generator=g) | 0x0f   # ~3 % NULL
        v1 = ri(0, 1000, n); v2 = ri(-500, 500, n)
        d128 = torch.stack([ri(0, 10**9, n), torch.zeros(n, dtype=torch.int64, device=dev)], dim=1).contiguous()
        SUM, CNT, MIN, MAX = L.AGG_SUM, L.AGG_COUNT, L.AGG_MIN, L.AGG_MAX
        for card in [int(x) for x in args.gbl_card.split(',')]:
            k1 = ri(0, card, n)
            c3 = max(2, round(card ** (1 / 3)))
            ka, kb, kc3 = ri(0, c3, n), ri(0, c3, n, torch.int32), ri(0, max(card // (c3 * c3), 1), n, torch.int32)
            k128 = torch.stack([k1, torch.zeros_like(k1)], dim=1).contiguous()
            cases = [
                ("i64 key; sum, count", [L.T_I64], None, [(SUM, L.T_I64, 0, 0, 0), (CNT, 0, 0, 0, 0)], [col(k1, L.T_I64)], [col(v1, L.T_I64), None], 16),
                ("nullable i64 key; sum(nullable), count", [L.T_I64], [1], [(SUM, L.T_I64, 0, 0, 1), (CNT, 0, 0, 0, 0)],
                 [col(k1, L.T_I64, validity=Borrowed(vbits))], [col(v1, L.T_I64, validity=Borrowed(vbits)), None], 16.25),
                ("3 keys (i64, date, i32); sum(Decimal128)", [L.T_I64, L.T_DATE, L.T_I32], None, [(SUM, L.T_DEC128, 31, 4, 0)],
                 [col(ka, L.T_I64), col(kb, L.T_DATE), col(kc3, L.T_I32)], [col(d128, L.T_DEC128)], 32),
                ("Decimal128 key; sum, count", [L.T_DEC128], None, [(SUM, L.T_I64, 0, 0, 0), (CNT, 0, 0, 0, 0)], [col(k128, L.T_DEC128)], [col(v1, L.T_I64), None], 24),
                ("i64 key; 8 aggregates", [L.T_I64], None,
                 [(SUM, L.T_I64, 0, 0, 0), (SUM, L.T_I64, 0, 0, 0), (MIN, L.T_I64, 0, 0, 0), (MAX, L.T_I64, 0, 0, 0), (SUM, L.T_I64, 0, 0, 0), (MIN, L.T_I64, 0, 0, 0),
                  (MAX, L.T_I64, 0, 0, 0), (CNT, 0, 0, 0, 0)], [col(k1, L.T_I64)],
                 [col(v1, L.T_I64), col(v2, L.T_I64), col(v1, L.T_I64), col(v1, L.T_I64), col(v2, L.T_I64), col(v2, L.T_I64), col(v2, L.T_I64), None], 24),
            ]
            fbits = D.Column(L.T_BOOL, n, Borrowed(torch.randint(0, 256, (n // 8 + 64,), device=dev, dtype=torch.uint8, generator=g)))   # ~50 % pass
            cases.append(("i64 key; sum, count; pushed-down filter (50 % pass)", [L.T_I64], None, [(SUM, L.T_I64, 0, 0, 0), (CNT, 0, 0, 0, 0)], [col(k1, L.T_I64)],
                          [col(v1, L.T_I64), None], 16.125))
            for name, kt, kn, aggs, kcols, acols, bpr in cases:
                gb = D.GroupBy(kt, aggs, key_nullable=kn, capacity=max(1024, card * 2))
                flt = fbits if "filter" in name else None
                def f():
                    gb.reset()
                    gb.add_block(kcols, acols, n, filter=flt)
                ms = timed(f, reps=3, warm=2)
                ng = gb.num_groups()
                report(out, f"groupby layouts: {name}, {card} groups", n, "rows", alg_bytes=bpr * n, ms=ms, note=f"{ng} groups met")
                gb.destroy()
            del k1, ka, kb, kc3, k128
        # TPC-H Q1's group-by without the fused program: the aggregate arguments arrive as columns
        rf = torch.zeros((n, 4), dtype=torch.int32, device=dev); rf[:, 0] = 1; rf[:, 1] = ri(65, 68, n, torch.int32)
        ls = torch.zeros((n, 4), dtype=torch.int32, device=dev); ls[:, 0] = 1; ls[:, 1] = ri(70, 72, n, torch.int32)
        gb = D.GroupBy([L.T_STRING, L.T_STRING], [(SUM, L.T_DEC64, 15, 2, 0), (SUM, L.T_DEC64, 15, 2, 0), (SUM, L.T_DEC128, 31, 4, 0), (SUM, L.T_DEC128, 38, 6, 0),
                                                 (SUM, L.T_DEC64, 15, 2, 0), (CNT, 0, 0, 0, 0)], capacity=1024)
        kcols = [col(rf, L.T_STRING), col(ls, L.T_STRING)]
        acols = [col(v1, L.T_DEC64), col(v2, L.T_DEC64), col(d128, L.T_DEC128), col(d128, L.T_DEC128), col(v1, L.T_DEC64), None]
        def f():
            gb.reset()
            gb.add_block(kcols, acols, n)
        ms = timed(f, reps=3, warm=2)
        report(out, "groupby layouts: TPC-H Q1's table from columns (2 String keys; 3 Decimal64 + 2 Decimal128 sums, count), 6 groups", n, "rows",
               alg_bytes=(32 + 24 + 32) * n, ms=ms, note=f"{gb.num_groups()} groups met")
        gb.destroy()
        del rf, ls, d128, v1, v2, vbits

    if want("join"):
        nb, npr = int(15_000_000 * args.scale), int(120_000_000 * args.scale)
        bk = torch.randperm(nb * 4, device=dev, generator=g)[:nb].to(torch.int64)
        pk = ri(0, nb * 4, npr)
        j = D.HashJoin(nb)
        bc = col(bk, L.T_U64)
        pc = col(pk, L.T_U64)
        def build():
            jj = D.HashJoin(nb)
            jj.add_block(bc)
            jj.final_build()
            jj.destroy()
        report(out, f"join build {nb} u64 keys (create+insert+finalize)", nb, "rows", alg_bytes=8 * nb, ms=timed(build, reps=3, warm=1))
        j.add_block(bc)
        j.final_build()
        total = C.c_uint64()
        fcount = lambda: check(Lb.dbhip_join_probe_count(j.h, C.c_void_p(pk.data_ptr()), None, C.c_int64(npr), C.byref(total), None))
        report(out, f"join probe_count {npr} probes vs {nb} build (25% match)", npr, "rows", alg_bytes=8 * npr, ms=timed(fcount, reps=3, warm=1))
        m = total.value
        op = torch.empty(m + 8, dtype=torch.int32, device=dev)
        ob = torch.empty(m + 8, dtype=torch.int32, device=dev)
        got = C.c_uint64()
        fprobe = lambda: check(Lb.dbhip_join_probe(j.h, C.c_void_p(pk.data_ptr()), None, C.c_int64(npr), C.c_void_p(op.data_ptr()), C.c_void_p(ob.data_ptr()), C.c_int64(m), C.byref(got), None))
        report(out, f"join probe (emit {m} ordered pairs)", npr, "rows", alg_bytes=8 * npr + 8 * m, ms=timed(fprobe, reps=3, warm=1))
        j.destroy()
        del bk, pk, op, ob

    if want("sort"):
        n = int(64_000_000 * args.scale)
        k64 = ri(-2**62, 2**62, n)
        kc = col(k64, L.T_I64)
        arr = (L.Col * 1)(kc.c())
        d0 = (C.c_uint8 * 1)(0)
        perm = torch.empty(n, dtype=torch.int32, device=dev)
        f = lambda: check(Lb.dbhip_sort_perm(arr, d0, d0, 1, C.c_int64(n), C.c_int64(0), C.c_void_p(perm.data_ptr()), None))
        report(out, f"sort_perm {n} x i64 (8 LSD passes)", n, "rows", alg_bytes=n * (8 + 4) * 2 * 8, ms=timed(f, reps=3, warm=1), note="alg = 8 passes x (read+write) x 12 B")
        k32 = ri(0, 2**31 - 1, n, torch.int32)
        arr32 = (L.Col * 1)(col(k32, L.T_I32).c())
        f = lambda: check(Lb.dbhip_sort_perm(arr32, d0, d0, 1, C.c_int64(n), C.c_int64(0), C.c_void_p(perm.data_ptr()), None))
        report(out, f"sort_perm {n} x i32", n, "rows", alg_bytes=n * (8 + 4) * 2 * 4, ms=timed(f, reps=3, warm=1), note="alg = 4 passes x 2 x 12 B")
        f = lambda: check(Lb.dbhip_sort_perm(arr, d0, d0, 1, C.c_int64(n), C.c_int64(10), C.c_void_p(perm.data_ptr()), None))
        report(out, f"sort_perm {n} x i64 LIMIT 10", n, "rows", alg_bytes=n * 8, ms=timed(f, reps=3, warm=1))
        del k64, k32, perm

    if want("vector"):
        n, dim = int(1_250_000 * args.scale), 768
        base = torch.randn((n, dim), device=dev, dtype=torch.float32, generator=g)
        for nq in [int(x) for x in args.vec_nq.split(',')]:
            q = torch.randn((nq, dim), device=dev, dtype=torch.float32, generator=g)
            oi = torch.empty(nq * 10, dtype=torch.int32, device=dev)
            od = torch.empty(nq * 10, dtype=torch.float32, device=dev)
            f = lambda: check(Lb.dbhip_vec_topk(L.VEC_COSINE, C.c_void_p(base.data_ptr()), C.c_int64(n), dim, C.c_void_p(q.data_ptr()), nq, 10, C.c_void_p(oi.data_ptr()), C.c_void_p(od.data_ptr()), None))
            ms = timed(f, reps=3, warm=1)
            if nq < 40:
                report(out, f"vec_topk cosine {n}x{dim}, nq={nq}, k=10", nq, "queries", alg_bytes=n * dim * 4, ms=ms)
            else:
                report(out, f"vec_topk cosine {n}x{dim}, nq={nq}, k=10", nq, "queries", flops=2.0 * n * dim * nq, ms=ms)
        for imetric, iname in ((L.VEC_COSINE, 'cosine'), (L.VEC_L2, 'l2')):
            ix = C.c_void_p()
            check(Lb.dbhip_vec_index_build(imetric, C.c_void_p(base.data_ptr()), C.c_int64(n), dim, C.byref(ix), None))
            for nq in [int(x) for x in args.vec_nq.split(',')]:
                if imetric == L.VEC_L2 and nq > 1 and nq < 2048:
                    continue
                q = torch.randn((nq, dim), device=dev, dtype=torch.float32, generator=g)
                oi = torch.empty(nq * 10, dtype=torch.int32, device=dev)
                od = torch.empty(nq * 10, dtype=torch.float32, device=dev)
                oi2 = torch.empty(nq * 10, dtype=torch.int32, device=dev)
                od2 = torch.empty(nq * 10, dtype=torch.float32, device=dev)
                f = lambda: check(Lb.dbhip_vec_index_search(ix, C.c_void_p(q.data_ptr()), nq, 10, C.c_void_p(oi.data_ptr()), C.c_void_p(od.data_ptr()), None))
                ms = timed(f, reps=3, warm=1)
                check(Lb.dbhip_vec_topk(imetric, C.c_void_p(base.data_ptr()), C.c_int64(n), dim, C.c_void_p(q.data_ptr()), nq, 10, C.c_void_p(oi2.data_ptr()), C.c_void_p(od2.data_ptr()), None))
                check(Lb.dbhip_stream_sync(None))
                same = float((oi == oi2).float().mean().item())
                report(out, f"vec_index_search {iname} (bf16 prefilter + exact rescore) {n}x{dim}, nq={nq}, k=10", nq, "queries", flops=2.0 * n * dim * nq, ms=ms,
                       note=f"frac is vs the FP32-MFMA peak the exact scan is bound by (bf16 dense peak 2500 TF); ids identical to the exact scan: {same:.4f}")
            check(Lb.dbhip_vec_index_destroy(ix))
        nq = 64
        q = torch.randn((nq, dim), device=dev, dtype=torch.float32, generator=g)
        o = torch.empty(nq * n, dtype=torch.float32, device=dev)
        for metric, nm in ((L.VEC_L2, "l2"), (L.VEC_DOT, "dot")):
            f = lambda: check(Lb.dbhip_vec_distance(metric, C.c_void_p(base.data_ptr()), C.c_int64(n), dim, C.c_void_p(q.data_ptr()), nq, C.c_void_p(o.data_ptr()), None))
            report(out, f"vec_distance {nm} {n}x{dim}, nq={nq}", nq, "queries", flops=(3.0 if nm == "l2" else 2.0) * n * dim * nq, ms=timed(f, reps=3, warm=1))
        del base, o

    if want("parquet"):
        # scan-side decode: lineitem-shaped column chunks written by pyarrow with the reference writer's
        # settings, chunk bytes already resident in HBM when the timed region starts. Algorithmic bytes = chunk bytes read +
        # column (+ validity) bytes written. The host-side plan (dbhip_pq_chunk_open) and the PCIe upload are reported in the note.
        import time as _t
        import numpy as np
        import pyarrow as pa
        sys.path.insert(0, ROOT)
        from tests import parquet_util as PU
        n = int(60_000_000 * args.scale)
        rng = np.random.default_rng(3)
        price = rng.integers(90000, 10494951, n)
        cols = [
            ("l_extendedprice Decimal(15,2) INT64 PLAIN v1", pa.array(price, pa.int64()), False, L.T_DEC64),
            ("l_extendedprice INT64 PLAIN v1, 3% NULL", pa.array(price, pa.int64(), mask=rng.random(n) < 0.03), False, L.T_DEC64),
            ("l_discount INT64 RLE_DICTIONARY v2 (11 values)", pa.array(rng.integers(0, 11, n), pa.int64()), True, L.T_DEC64),
            ("l_shipdate Date INT32 RLE_DICTIONARY v2 (2526 values)", pa.array(rng.integers(8036, 10562, n).astype(np.int32), pa.int32()).cast(pa.date32()), True, L.T_DATE),
            ("l_returnflag String RLE_DICTIONARY v2 (3 values) -> 16-B views", pa.array(np.array(["A", "R", "N"])[rng.integers(0, 3, n)], pa.string()), True, L.T_STRING),
            ("l_comment-like String PLAIN v1 (unique, 10-44 B) -> views into the chunk", None, False, L.T_STRING),
        ]
        for name, arr, dictionary, ot in cols:
            m = n
            if arr is None:
                m = n // 4
                lens = rng.integers(10, 45, m)
                blob = rng.integers(97, 123, int(lens.sum()), dtype=np.uint8)
                offs = np.zeros(m + 1, dtype=np.int32)
                np.cumsum(lens, out=offs[1:])
                arr = pa.Array.from_buffers(pa.string(), m, [None, pa.py_buffer(offs.tobytes()), pa.py_buffer(blob.tobytes())])
            chunks, _ = PU.column_chunks(PU.write_parquet(pa.table({"c": arr}), dictionary=dictionary))
            ch = chunks[0]
            t0 = _t.perf_counter()
            pc = D.ParquetChunk(ch["chunk"], ch["physical"], ot, ch["type_length"], ch["max_def"])
            open_ms = (_t.perf_counter() - t0) * 1e3
            t0 = _t.perf_counter()
            cd = pc.upload()
            check(Lb.dbhip_stream_sync(None))
            up_ms = (_t.perf_counter() - t0) * 1e3
            i = pc.info
            outb = D.DeviceBuffer(i.out_bytes + 16)
            valb = D.DeviceBuffer(i.validity_bytes + 8) if i.has_validity else None
            f = lambda: check(Lb.dbhip_pq_chunk_decode(pc.h, C.c_void_p(cd.ptr), C.c_void_p(outb.ptr), C.c_void_p(valb.ptr) if valb else None, None))
            ms = timed(f, reps=5, warm=2)
            alg = len(ch["chunk"]) + i.out_bytes + (i.validity_bytes if i.has_validity else 0)
            report(out, f"parquet decode {name}, {m} rows", m, "rows", alg_bytes=alg, ms=ms,
                   note=f"chunk {len(ch['chunk'])} B, {i.n_pages} pages, {i.num_nulls} nulls; host plan (open) {open_ms:.1f} ms; H2D upload of the chunk {up_ms:.1f} ms "
                        f"(pageable host memory); decode + upload + plan = {m / ((ms[0] + open_ms + up_ms) * 1e-3) / 1e9:.2f} G rows/s")
            pc.close()

    if want("u8"):
        n, dim = int(8_000_000 * args.scale), 768
        b = torch.randint(0, 128, (n, dim), device=dev, dtype=torch.uint8, generator=g)
        q = torch.randint(0, 128, (dim,), device=dev, dtype=torch.uint8, generator=g)
        o = torch.empty(n, dtype=torch.float32, device=dev)
        f = lambda: check(Lb.dbhip_score_u8(0, C.c_void_p(q.data_ptr()), C.c_void_p(b.data_ptr()), C.c_int64(n), dim, C.c_void_p(o.data_ptr()), None))
        report(out, f"score_u8 dot {n}x{dim}", n, "rows", alg_bytes=n * dim + 4 * n, ms=timed(f))

    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        json.dump(out, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()

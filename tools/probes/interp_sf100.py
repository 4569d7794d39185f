#!/usr/bin/env python3
"""The interpreter bridge at full size: TPC-H Q1 SF100 as ONE fused program WITHOUT the run-time specialised kernel
(DBHIP_FAGG_JIT=0: every launch goes through the ahead-of-time interpreter), checked against the hand-written kernel's result.

    DBHIP_FAGG_JIT=0 python tools/probes/interp_sf100.py [--sf 100] > gpurun_out/interp_sf100.json
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sf", type=float, default=100.0)
    ap.add_argument("--reps", type=int, default=3)
    args = ap.parse_args()
    assert os.environ.get("DBHIP_FAGG_JIT") == "0", "run with DBHIP_FAGG_JIT=0"
    import torch
    from databend_amd import device as D, tpch
    from databend_amd._lib import check, lib
    D.init(0)
    L = lib()
    n = tpch.rows_for_sf(args.sf)
    li = tpch.LineitemTorch(n, seed=2, torch=torch, row0=0)
    plan = tpch.q1_program(li)
    ref = tpch.q1_rows(tpch.q1_fused(li))
    st0 = (C.c_uint64 * 3)()
    L.dbhip_fagg_stats(st0)
    before = list(st0)
    g = tpch.q1_fused_program(li, plan=plan)       # warm-up
    same = tpch.q1_rows(g) == ref
    ts, kms = [], []
    for _ in range(args.reps):
        check(L.dbhip_stream_sync(None))
        t0 = time.perf_counter()
        tpch.q1_fused_program(li, plan=plan)
        check(L.dbhip_stream_sync(None))
        ts.append((time.perf_counter() - t0) * 1e3)
        k = C.c_float()
        if L.dbhip_last_kernel_ms(C.byref(k)) == 0:
            kms.append(k.value)
    L.dbhip_fagg_stats(st0)
    print(json.dumps({"what": "TPC-H Q1 as one fused program through the INTERPRETER (DBHIP_FAGG_JIT=0)", "rows": n, "ms": ts, "kernel_ms": kms,
                      "ms_per_60M_rows": min(ts) * 60e6 / n, "equals_hand_written_kernel": bool(same),
                      "launches_specialised_interpreted_pending": [int(a - b) for a, b in zip(st0, before)]}))
    assert same


if __name__ == "__main__":
    main()

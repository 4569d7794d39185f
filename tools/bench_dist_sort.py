"""One rank's share of the distributed sort and of the shuffle hash join's scatter (databend_amd.dist.range_partitioned_sort) on one MI355X: the stages a rank runs on
`--rows` rows of an i64 key + an i64 payload cut at `--ranges - 1` bounds — range partition (dbhip_sort_bound_partition), grouping
by range (one radix pass over the partition ids + dbhip_take_block), and the local sort of what a rank receives (rows / ranges
... here: the same rows, i.e. the balanced case) — each timed with HIP events through torch on the library's stream.
  python tools/bench_dist_sort.py --rows 60000000 --ranges 8 --out gpurun_out/dist_sort.json"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(torch, rows=60_000_000, ranges=8, reps=5):
    """-> dict of stage timings (ms, wall clock around synchronised library calls) for `rows` rows of (i64 key, i64 payload)"""
    from databend_amd import device as D, _lib as L
    from databend_amd.sort_bounds import balanced_cuts
    from databend_amd.sort_ops import ShuffleDeviceOps, SortDeviceOps
    ops, sops = SortDeviceOps(torch), ShuffleDeviceOps(torch)
    g = torch.Generator(device="cuda").manual_seed(3)
    key = torch.randint(-2**62, 2**62, (rows,), dtype=torch.int64, device="cuda", generator=g)
    pay = torch.arange(rows, dtype=torch.int64, device="cuda")
    flat = [key, pay]
    ids = (torch.arange(1024, dtype=torch.int64, device="cuda") * rows) // 1024
    bounds = balanced_cuts(ops.ordered_rows([key[ids]], [None], [0], [0]), ranges)

    def wall(fn):
        best = 1e9
        for _ in range(reps):
            torch.cuda.synchronize()
            L.check(L.lib().dbhip_stream_sync(None))
            t0 = time.perf_counter()
            r = fn()
            L.check(L.lib().dbhip_stream_sync(None))
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) * 1e3)
        return best, r

    kcol = [D.Column(L.T_I64, rows, type("B", (), {"ptr": key.data_ptr(), "nbytes": rows * 8})())]
    bcol = [D.Column.from_numpy(np.array([b[0] for b in bounds], dtype=np.int64), L.T_I64)]
    t_part, (part, counts) = wall(lambda: D.sort_bound_partition(kcol, bcol))
    t_all, (grouped, counts2) = wall(lambda: ops.partition(flat, [0], [None], bounds, [0], [0]))
    share = rows // ranges
    recv = [c[:share].contiguous() for c in grouped]
    t_sort, out = wall(lambda: ops.sort(recv, [0], [None], [0], [0]))
    t_sort_all, _ = wall(lambda: ops.sort(flat, [0], [None], [0], [0]))
    t_scat, (_sidx, scnt) = wall(lambda: D.scatter_indices(kcol, ranges, 0))
    t_scat_all, _ = wall(lambda: sops.scatter(flat, 0, None, ranges))
    # the ABI-owned block exchange (dbhip_exchange_begin / _finish) of this rank, in a world of ONE: everything a rank does around the
    # wire — scatter by destination, the counts read back and exchanged, the payload pieces "sent" (device copies), the concat of what
    # arrived — with nothing on the wire; begin (incl. its host round trip) and finish timed apart
    import ctypes as C
    comm = D.Comm.local()
    pcol = D.Column(L.T_I64, rows, type("B", (), {"ptr": pay.data_ptr(), "nbytes": rows * 8})())
    dest = torch.zeros(rows, dtype=torch.int32, device="cuda")
    dbuf = type("B", (), {"ptr": dest.data_ptr(), "nbytes": rows * 4})()
    xcols = [kcol[0], pcol]
    outs = [D.DeviceBuffer(rows * 8 + 64) for _ in xcols]
    t_begin, t_finish = 1e9, 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        x, got = C.c_void_p(), C.c_int64()
        t0 = time.perf_counter()
        L.check(L.lib().dbhip_exchange_begin(comm.h, D._cols(xcols), len(xcols), C.c_void_p(dbuf.ptr), C.c_int64(rows), C.byref(got), C.byref(x), None))
        L.check(L.lib().dbhip_stream_sync(None))
        t1 = time.perf_counter()
        dp = (C.c_void_p * 2)(*[b.ptr for b in outs])
        vp = (C.c_void_p * 2)(None, None)
        starts = (C.c_int64 * 2)()
        L.check(L.lib().dbhip_exchange_finish(x, dp, vp, starts, None))
        L.check(L.lib().dbhip_stream_sync(None))
        t2 = time.perf_counter()
        L.lib().dbhip_exchange_destroy(x)
        assert got.value == rows
        t_begin, t_finish = min(t_begin, (t1 - t0) * 1e3), min(t_finish, (t2 - t1) * 1e3)
    comm.destroy()
    assert int(scnt.sum()) == rows
    assert counts.tolist() == [int(c) for c in counts2] and int(counts.sum()) == rows
    assert bool((out[0][1:] >= out[0][:-1]).all())
    return {"rows": rows, "ranges": ranges, "rows_per_range": [int(c) for c in counts],
            "bound_partition_ms": round(t_part, 3), "bound_partition_GBps": round(rows * 12 / t_part / 1e6, 1),
            "partition_operator_ms": round(t_all, 3),
            "siphash_scatter_indices_ms": round(t_scat, 3), "siphash_scatter_indices_GBps": round(rows * 12 / t_scat / 1e6, 1),
            "scatter_operator_ms": round(t_scat_all, 3), "rows_per_destination": [int(c) for c in scnt],
            "local_sort_of_one_share_ms": round(t_sort, 3), "single_gpu_sort_of_all_rows_ms": round(t_sort_all, 3),
            "abi_exchange_world_of_one": {"begin_ms": round(t_begin, 3), "finish_ms": round(t_finish, 3), "GBps": round(rows * 16 * 2 / (t_begin + t_finish) / 1e6, 1),
                                          "what": "dbhip_exchange_begin (scatter_columns of two 8-byte columns by destination + the counts' device-to-host copy + "
                                                  "the count all-to-all) and dbhip_exchange_finish (the grouped send / recv of every column = device copies here, "
                                                  "+ concat_columns), the whole block to one destination; no wire"},
            "what": "one rank's stages of the distributed sort (sample -> Bounds -> dbhip_sort_bound_partition -> dbhip_scatter_block | all-to-all | "
                    "dbhip_sort_perm) and of the shuffle hash join's scatter (dbhip_scatter_indices = the reference's siphash64 % n -> "
                    "dbhip_scatter_block) on (i64 key, i64 payload) rows; wall clock around synchronised library calls"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=60_000_000)
    ap.add_argument("--ranges", type=int, default=8)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    import torch
    from databend_amd import device as D
    D.init(0)
    res = run(torch, a.rows, a.ranges, a.reps)
    print(json.dumps(res))
    if a.out:
        json.dump(res, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()

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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=60_000_000)
    ap.add_argument("--ranges", type=int, default=8)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    import torch
    from databend_amd import device as D, _lib as L
    from databend_amd.sort_bounds import balanced_cuts
    from databend_amd.sort_ops import SortDeviceOps
    D.init(0)
    ops = SortDeviceOps(torch)
    g = torch.Generator(device="cuda").manual_seed(3)
    key = torch.randint(-2**62, 2**62, (a.rows,), dtype=torch.int64, device="cuda", generator=g)
    pay = torch.arange(a.rows, dtype=torch.int64, device="cuda")
    flat = [key, pay]
    ids = (torch.arange(1024, dtype=torch.int64, device="cuda") * a.rows) // 1024
    bounds = balanced_cuts(ops.ordered_rows([key[ids]], [None], [0], [0]), a.ranges)

    def wall(fn):
        best = 1e9
        for _ in range(a.reps):
            torch.cuda.synchronize()
            L.check(L.lib().dbhip_stream_sync(None))
            t0 = time.perf_counter()
            r = fn()
            L.check(L.lib().dbhip_stream_sync(None))
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) * 1e3)
        return best, r

    kcol = [D.Column(L.T_I64, a.rows, type("B", (), {"ptr": key.data_ptr(), "nbytes": a.rows * 8})())]
    bcol = [D.Column.from_numpy(np.array([b[0] for b in bounds], dtype=np.int64), L.T_I64)]
    t_part, (part, counts) = wall(lambda: D.sort_bound_partition(kcol, bcol))
    t_group, (perm, m) = wall(lambda: D.sort_perm_device([D.Column(L.T_U32, a.rows, part)]))
    t_take, grouped = wall(lambda: ops._take(flat, perm, m))
    t_all, (grouped2, counts2) = wall(lambda: ops.partition(flat, [0], [None], bounds, [0], [0]))
    share = a.rows // a.ranges
    recv = [c[:share].contiguous() for c in grouped]
    t_sort, out = wall(lambda: ops.sort(recv, [0], [None], [0], [0]))
    t_sort_all, _ = wall(lambda: ops.sort(flat, [0], [None], [0], [0]))
    t_scat, (sidx, scnt) = wall(lambda: D.scatter_indices(kcol, a.ranges, 0))
    from databend_amd.sort_ops import ShuffleDeviceOps
    sops = ShuffleDeviceOps(torch)
    t_scat_all, _ = wall(lambda: sops.scatter(flat, 0, None, a.ranges))
    assert int(scnt.sum()) == a.rows
    assert counts.tolist() == [int(c) for c in counts2] and int(counts.sum()) == a.rows
    assert bool((out[0][1:] >= out[0][:-1]).all())
    res = {"rows": a.rows, "ranges": a.ranges, "rows_per_range": [int(c) for c in counts],
           "bound_partition_ms": round(t_part, 3), "bound_partition_GBps": round(a.rows * 12 / t_part / 1e6, 1),
           "group_by_range_perm_ms": round(t_group, 3), "take_block_2cols_ms": round(t_take, 3), "partition_operator_ms": round(t_all, 3),
           "siphash_scatter_indices_ms": round(t_scat, 3), "siphash_scatter_indices_GBps": round(a.rows * 12 / t_scat / 1e6, 1),
           "scatter_operator_ms": round(t_scat_all, 3), "rows_per_destination": [int(c) for c in scnt],
           "local_sort_of_one_share_ms": round(t_sort, 3), "single_gpu_sort_of_all_rows_ms": round(t_sort_all, 3),
           "note": "wall clock around synchronised library calls (includes the host-side sync of each call); one rank's stages of the distributed sort"}
    print(json.dumps(res))
    if a.out:
        json.dump(res, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()

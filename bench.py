#!/usr/bin/env python3
"""bench.py — TPC-H Q1 hash-aggregation throughput (rows/s) on synthetic lineitem, SF100 (BASELINE.json's metric).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--sf 100]

A "step" is one pass of the Q1 pipeline (filter -> decimal maps -> partial hash aggregation) over one rank's lineitem
shard that is already resident in HBM, followed by the partial-state exchange over RCCL when N > 1 and the final merge.
  N = 1: the whole SF100 lineitem (600,037,902 rows, 68 B/row = 40.8 GB) on one MI355X.
  N > 1: BASELINE configs[3] — the SAME SF100 table split into N row-range shards (rank r holds rows
         [r n/N, (r+1) n/N)); `scaling` is therefore "strong". `--exchange alltoall` (default for N > 1) routes every
         partial-state row to rank hash % N on the device (dbhip_groupby_partition_blocks) and exchanges them with one
         all_to_all_single; `--exchange allgather` all-gathers the blocks and merges everything on every rank.
The lineitem shard is generated on the device (databend_amd.tpch.LineitemTorch: torch is plumbing; identical data for
every N). Prints ONE JSON line (rank 0) with the contract fields plus
  roofline       q1_fused_kernel, HIP events on its launch stream, 68 B/row algorithmic
  cpu_baseline   the reference-shaped CPU Q1 (oracle/q1_typed.c) timed on the host cores over a stated sample,
                 and used as the checker of the device result (sample AND the whole table, chunk by chunk)
  q1_operator_plan  the SAME query through the generic operator kernels (what a drop-in dispatches to), ms + fraction
  q3_sf100       BASELINE configs[2] (tools/bench_q3.py's plan) in the same process
  ann            BASELINE configs[4]
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BYTES_PER_ROW = 68  # 4 x Decimal64 + 2 x 16-B view + Date32
HBM_PEAK_GBS = 8000.0


def launcher_command(n_gpus, argv, port=None, python=None):
    """The command line `python bench.py --gpus N` re-executes itself under when N > 1 and no launcher set WORLD_SIZE: one rank
    per GPU on THIS node through torch.distributed.run, rendezvous on 127.0.0.1 (the container hostname may not resolve).
    `argv` = this process's own arguments (sys.argv[1:]), passed through unchanged."""
    if port is None:
        import socket
        with socket.socket() as s:          # a free port of the loopback interface
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
    return [python or sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={int(n_gpus)}",
            "--master-addr", "127.0.0.1", "--master-port", str(int(port)), os.path.abspath(__file__), *argv]


def self_launch(args, argv):
    """`--gpus N` with N > 1 outside a launcher: start the N ranks ourselves and hand their exit code back. Never falls through
    to a one-rank run (a line with n_gpus = 1 for an N > 1 request would void the scaling measurement)."""
    import subprocess
    if not args.share_gpu:
        import torch
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < args.gpus:
            raise SystemExit(f"bench.py --gpus {args.gpus}: this node exposes {have} GPU(s); refusing to run fewer ranks than asked "
                             f"(--share-gpu --backend gloo runs a FUNCTIONAL check of the N-rank path on one GPU)")
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: what RCCL needs on this driver
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = launcher_command(args.gpus, argv)
    print("bench.py: launching " + " ".join(cmd), file=sys.stderr, flush=True)
    raise SystemExit(subprocess.call(cmd, env=env))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--sf", type=float, default=100.0)
    ap.add_argument("--rows", type=int, default=0, help="override the rows of the WHOLE job")
    ap.add_argument("--exchange", default="", choices=["", "allgather", "alltoall"], help="partial-state exchange for N > 1")
    ap.add_argument("--cpu-rows", type=int, default=59_986_052, help="rows of the CPU-baseline sample (default: an SF10-sized prefix)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline and the CPU checks")
    ap.add_argument("--no-verify-full", action="store_true", help="skip the full-size CPU check of the device result")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="wall time of the timed CPU-baseline passes")
    ap.add_argument("--no-opplan", action="store_true", help="skip the generic operator-plan measurement")
    ap.add_argument("--no-readiness", action="store_true", help="skip the SF100/8 exchange-overhead measurement (it launches q1_fused_kernel on 1/8 of the rows: "
                    "profiles of the headline kernel are taken without it)")
    ap.add_argument("--no-q3", action="store_true", help="skip TPC-H Q3 SF100 (BASELINE configs[2])")
    ap.add_argument("--no-blocks", action="store_true", help="skip the block-size sweep (Q1 fed as 65,536 ... all-row blocks through the C-ABI from 1 and 8 host threads)")
    ap.add_argument("--q3-sf", type=float, default=100.0)
    ap.add_argument("--no-hnsw", action="store_true", help="skip the HNSW reference-comparable mode inside the ANN measurement")
    ap.add_argument("--hnsw-rows", type=int, default=1_000_000, help="base vectors of the HNSW reference-comparable run (the build is timed too)")
    ap.add_argument("--no-ann", action="store_true", help="skip the secondary ANN measurement (BASELINE configs[4])")
    ap.add_argument("--no-scan", action="store_true", help="skip the scan-side measurement (Q1's seven columns as ZSTD Parquet pages -> HBM columns)")
    ap.add_argument("--ann-rows", type=int, default=10_000_000, help="base vectors of the WHOLE job (sharded by row range over the ranks)")
    ap.add_argument("--ann-dim", type=int, default=768)
    ap.add_argument("--ann-queries", type=int, default=10_000, help="queries per ANN step (replicated on every rank)")
    ap.add_argument("--ann-steps", type=int, default=3)
    ap.add_argument("--backend", default="nccl", help="process-group backend (nccl = RCCL; gloo only for a functional check)")
    ap.add_argument("--share-gpu", action="store_true", help="functional check of the N>1 path on a 1-GPU box: every rank uses "
                    "cuda:0 (with --backend gloo); the numbers of such a run are not a measurement")
    ap.add_argument("--headline", choices=["program", "hand"], default="program",
                    help="what `value` / `roofline` time: the generic fused program through dbhip_groupby_add_block_program (default), or the "
                         "query-specific hand-written kernel dbhip_q1_fused")
    ap.add_argument("--exchange-impl", choices=["torch", "abi"], default="torch",
                    help="N > 1: who owns the communicator of the partial-state exchange: torch.distributed (default), or the C-ABI's "
                         "own RCCL communicator (dbhip_comm_*, dbhip_groupby_exchange_*: what a Rust host would call)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args, sys.argv[1:])       # does not return
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the line would not describe the job that ran")

    import numpy as np
    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: libdbhip has no CPU fallback")
    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    # who really takes part: one all-reduce of ones over the process group's own backend (nccl = RCCL), and the distinct
    # (PCI bus id) devices behind the ranks — a line from ranks that share a GPU says so
    ranks_seen, devices_seen = 1, 1
    if world > 1:
        one = torch.ones(1, dtype=torch.int64, device="cuda")
        dist.all_reduce(one, op=dist.ReduceOp.SUM)
        ranks_seen = int(one.item())
        ids = [None] * world
        dist.all_gather_object(ids, str(getattr(torch.cuda.get_device_properties(local_rank), "pci_bus_id", local_rank))
                               + ":" + str(getattr(torch.cuda.get_device_properties(local_rank), "uuid", "")))
        devices_seen = len(set(ids))

    from databend_amd import device as D, tpch
    from databend_amd import dist as DX
    from databend_amd._lib import check, lib
    D.init(local_rank)
    L = lib()
    exchange = args.exchange or ("alltoall" if world > 1 else "")

    n_total = args.rows or tpch.rows_for_sf(args.sf)
    # row-range shards of ONE table; shard boundaries are multiples of 4 rows (16-byte alignment of every column)
    lo = (rank * n_total // world) & ~3
    hi = n_total if rank == world - 1 else ((rank + 1) * n_total // world) & ~3
    n = hi - lo
    t0 = time.perf_counter()
    li = tpch.LineitemTorch(n, seed=2, torch=torch, row0=lo)
    gen_s = time.perf_counter() - t0
    g = D.GroupBy.q1()

    # N = 1: the library's own stream (every dbhip call of a step is ordered on it). N > 1: one torch side stream is
    # handed to the library, so that the fused kernel, the block flush, the RCCL collective and the merge of the other
    # ranks' blocks are ordered on ONE stream with no host round trip between them (torch's default stream has handle
    # 0, which the C-ABI reads as "the library's stream", hence a side stream).
    ts = torch.cuda.Stream() if world > 1 else None
    stream = C.c_void_p(ts.cuda_stream) if ts is not None else None
    kms = []
    abi_comm = None
    # N > 1 over RCCL: BOTH exchange implementations are set up — the one `--exchange-impl` names carries the headline, the other is
    # timed right after it (same steps) and reported beside it (`exchange_impl_other`): torch.distributed's collectives on the
    # library's buffers, and the C-ABI's own communicator (dbhip_comm_*: what a Rust host would call). (Ranks that SHARE a GPU cannot
    # form an RCCL communicator — RCCL refuses duplicate devices — so `--share-gpu` runs the selected implementation only.)
    both_impls = world > 1 and args.backend == "nccl" and not args.share_gpu
    abi_handle = None
    if world > 1 and (args.exchange_impl == "abi" or both_impls):
        ids = [D.Comm.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(ids, src=0)     # the host's control plane ships the 128 bytes
        abi_handle = D.Comm(rank, world, ids[0])
    if args.exchange_impl == "abi":
        abi_comm = abi_handle

    # THE HEADLINE PATH is the generic one: the binding flattens Q1's predicate and decimal maps into one register
    # program and dbhip_groupby_add_block_program runs the fused filter -> map -> partial-aggregate kernel the library specialised
    # for it at PREPARE time (hiprtc) — what a physical plan can dispatch to. The query-specific hand-written kernel (k_q1.hip,
    # dbhip_q1_fused) is measured beside it as the ceiling (`hand_written_kernel`); `--headline hand` swaps the two.
    plan = tpch.q1_program(li)
    prepare_ms = None
    if args.headline == "program":
        t0p = time.perf_counter()
        tpch.q1_fused_program(li, g, prepare=True, plan=plan)     # PREPARE of the pipeline: compile / load the specialised kernel
        prepare_ms = (time.perf_counter() - t0p) * 1e3

    def launch(record, into):
        if args.headline == "program":
            tpch.q1_fused_program(li, g, plan=plan, stream=stream)
        else:
            D.q1_fused(g, li.qty, li.price, li.disc, li.tax, li.rf, li.ls, li.ship, tpch.Q1_CUTOFF, stream=stream)
        if record:
            ms = C.c_float()
            check(L.dbhip_last_kernel_ms(C.byref(ms)))  # HIP events around the fused kernel on its stream
            into.append(ms.value)

    def step(record=False, use_abi=None):
        g.reset(stream)
        launch(record, kms)
        comm = abi_comm if use_abi is None else (abi_handle if use_abi else None)
        if comm is not None:
            (comm.exchange_alltoall if exchange == "alltoall" else comm.exchange_allgather)(g, 256, stream)
        elif world > 1:
            with torch.cuda.stream(ts):
                if exchange == "alltoall":
                    DX.exchange_partials_alltoall_nccl(g, dist, torch, stream=stream)
                else:
                    DX.exchange_partials_nccl(g, dist, torch, stream=stream)
        return g

    for _ in range(args.warmup):
        step()
    check(L.dbhip_stream_sync(stream))
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(True)
    check(L.dbhip_stream_sync(stream))
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # the OTHER exchange implementation, same steps, same barriers (N > 1 over RCCL only)
    other_impl = None
    if both_impls:
        use_abi = args.exchange_impl != "abi"
        for _ in range(args.warmup):
            step(False, use_abi)
        check(L.dbhip_stream_sync(stream))
        torch.cuda.synchronize()
        dist.barrier()
        t1 = time.perf_counter()
        for i in range(args.steps):
            step(False, use_abi)
        check(L.dbhip_stream_sync(stream))
        torch.cuda.synchronize()
        dist.barrier()
        t = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        other_impl = {"impl": "abi" if use_abi else "torch", "ms_per_step": float(t.item()) / args.steps * 1e3,
                      "rows_per_s": n_total * args.steps / float(t.item()), "_rows": tpch.q1_rows(g)}
        step(False)   # (leave the table as the selected implementation's step leaves it: the result checks below read it)
        check(L.dbhip_stream_sync(stream))
    # average launch duration of the dominant kernel (the specialised fused-program kernel, or q1_fused_kernel), HIP events on its stream
    kernel_ms = float(np.mean(kms)) if kms else 0.0
    dominant = "fagg_jit (run-time specialised fused filter+map+aggregate program)" if args.headline == "program" else "q1_fused_kernel"
    result_headline = tpch.q1_rows(g)
    if other_impl is not None:
        other_impl["equals_headline_result"] = other_impl.pop("_rows") == result_headline
    # the OTHER kernel beside it, same rows, same steps: its per-launch duration and wall time
    other = None
    if world == 1:
        okms = []
        g2 = D.GroupBy.q1()

        def other_step(rec):
            g2.reset(stream)
            if args.headline == "program":
                D.q1_fused(g2, li.qty, li.price, li.disc, li.tax, li.rf, li.ls, li.ship, tpch.Q1_CUTOFF, stream=stream)
            else:
                tpch.q1_fused_program(li, g2, plan=plan, stream=stream)
            if rec:
                ms = C.c_float()
                check(L.dbhip_last_kernel_ms(C.byref(ms)))
                okms.append(ms.value)
        if args.headline == "hand":
            tpch.q1_fused_program(li, g2, prepare=True, plan=plan)
        for _ in range(max(args.warmup, 1)):
            other_step(False)
        check(L.dbhip_stream_sync(stream))
        t0o = time.perf_counter()
        for _ in range(args.steps):
            other_step(True)
        check(L.dbhip_stream_sync(stream))
        oms = (time.perf_counter() - t0o) * 1e3 / args.steps
        okm = float(np.mean(okms))
        same = tpch.q1_rows(g2) == result_headline
        other = {"name": "q1_fused_kernel (query-specific, k_q1.hip: the ceiling)" if args.headline == "program" else "fagg_jit (generic fused program)",
                 "ms_per_step": oms, "kernel_ms": okm, "kernel_ms_min_median_max": [float(np.min(okms)), float(np.median(okms)), float(np.max(okms))],
                 "rows_per_s": n_total / (oms * 1e-3),
                 "hbm_frac": n * BYTES_PER_ROW / (okm * 1e-3) / 1e9 / HBM_PEAK_GBS if okm else None,
                 "headline_kernel_slowdown": kernel_ms / okm if okm else None, "equals_headline_result": bool(same)}
        assert same, "the generic fused program and the hand-written kernel disagree"
        g2.destroy()

    value = n_total * args.steps / dt
    result = tpch.q1_rows(g)
    n_groups = len(result)
    if world > 1:
        # the exchanged result against the un-exchanged local states: count(*) summed over the ranks, and every aggregate
        # of every group summed over the ranks (allgather: every rank holds everything, so the sums are world x the local
        # sums; alltoall: rank r holds the groups with hash % world == r, once)
        local = tpch.q1_rows(tpch.q1_fused(li))
        keys = [(b"A", b"F"), (b"N", b"F"), (b"N", b"O"), (b"R", b"F")]
        fields = ("sum_qty", "sum_base_price", "sum_disc_price", "sum_charge", "sum_disc", "count")

        def vec(rows):
            # 3 x 42-bit limbs per value: exact in int64 under a SUM all-reduce over <= 8 ranks
            out = []
            for k in keys:
                for f in fields:
                    v = int(rows.get(k, {}).get(f, 0))
                    out += [v & ((1 << 42) - 1), (v >> 42) & ((1 << 42) - 1), v >> 84]
            return torch.tensor(out, dtype=torch.int64, device="cuda")
        exp, got = vec(local), vec(result)
        dist.all_reduce(exp, op=dist.ReduceOp.SUM)
        dist.all_reduce(got, op=dist.ReduceOp.SUM)
        mult = world if exchange == "allgather" else 1

        def ints(t):  # limb sums -> exact python ints (the limbs of a sum are not the sums of the limbs: recombine first)
            x = t.tolist()
            return [x[i] + (x[i + 1] << 42) + (x[i + 2] << 84) for i in range(0, len(x), 3)]
        assert [v * mult for v in ints(exp)] == ints(got), "exchange lost or duplicated partial states"
        ng = torch.tensor([len(result)], dtype=torch.int64, device="cuda")
        dist.all_reduce(ng, op=dist.ReduceOp.SUM if exchange == "alltoall" else dist.ReduceOp.MAX)
        n_groups = int(ng.item())

    readiness = None
    if world == 1 and rank == 0 and n_total >= 8 * (1 << 20) and not args.no_readiness:
        readiness = bench_exchange_overhead(li, n_total // 8 & ~3, tpch, D, L, check, args.steps)

    opplan = None
    if not args.no_opplan and world == 1:
        opplan = bench_operator_plan(li, tpch, D, L, check, result, kernel_ms)

    cpu = None
    if rank == 0 and not args.no_cpu and world == 1:
        cpu = cpu_baseline(args, li, n, tpch, result)

    q3 = None
    ann = None
    del li  # the lineitem shard is not needed any more: give its HBM back
    torch.cuda.empty_cache()
    if not args.no_q3 and world == 1:
        q3 = bench_q3(args, torch, tpch, D, L, check)
        torch.cuda.empty_cache()
    elif not args.no_q3:
        try:      # (a secondary measurement never takes the headline line down)
            q3 = bench_q3_dist(args, rank, world, torch, dist, tpch, D, DX, L, check)
        except Exception as e:  # noqa: BLE001
            q3 = {"error": repr(e)}
        torch.cuda.empty_cache()
    plans = None
    if world == 1 and not args.no_readiness:
        # one rank's stages of the distributed sort and of the shuffle join's scatter (SURVEY §8e rows "sort" / "hash join") at 60 M rows
        try:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import bench_dist_sort as BDS
            plans = BDS.run(torch, rows=60_000_000, ranges=8, reps=3)
        except Exception as e:      # a secondary measurement never takes the headline line down
            plans = {"error": repr(e)}
        torch.cuda.empty_cache()
    scan = None
    if not args.no_scan and world == 1:
        # the scan side in front of the hot path: Q1's seven lineitem columns as the reference's writer lays them out —
        # ZSTD level 1 (its default codec), DATA_PAGE_V2, 20 000-row pages — 8 blocks of 6 M rows decoded by ONE dbhip_pq_chunks_decode_device
        # call from the stored bytes resident in HBM; the decode is checked to be the identity on what was written
        try:
            torch.cuda.empty_cache()
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import pq_scan_probe as PSP
            scan = PSP.run(codec="zstd", reps=3)
        except Exception as e:      # a secondary measurement never takes the headline line down
            scan = {"error": repr(e)}
        torch.cuda.empty_cache()
    if not args.no_ann:
        ann = bench_ann(args, rank, world, torch, dist, D, DX, L, check)
        if world == 1 and not args.no_hnsw:
            # the reference's own indexed path beside the exact index: HNSW (m=10, ef_construct=40, ef=4k) over u8-quantised
            # vectors, same distribution (SURVEY §8d C5 "reference-comparable mode"), smaller base (the build is part of it)
            torch.cuda.empty_cache()
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import bench_hnsw as BH
            ann["hnsw_reference_mode"] = BH.run(rows=args.hnsw_rows, dim=args.ann_dim, queries=10_000, k=10)
            # embeddings are not i.i.d. Gaussians (where every point is equally far from every other and a graph index has
            # nothing to find): unit-length vectors around 1024 centres, the shape embedding models emit
            ann["hnsw_reference_mode_clustered"] = BH.run(rows=args.hnsw_rows, dim=args.ann_dim, queries=10_000, k=10, clusters=1024, normalize=True)

    blocks = None
    if rank == 0 and world == 1 and not args.no_blocks:
        blocks = bench_block_sizes(cpu)
    if rank == 0:
        achieved = (n * BYTES_PER_ROW) / (kernel_ms * 1e-3) / 1e9 if kernel_ms else 0.0
        traffic, traffic_src = None, None
        tf = os.path.join(ROOT, "profiles", "q1_traffic_program.json" if args.headline == "program" else "q1_traffic.json")
        if os.path.exists(tf):
            try:
                tj = json.load(open(tf))
                if tj.get("rows") == n:
                    traffic, traffic_src = tj.get("hbm_bytes_per_launch"), tj.get("source")
            except Exception:
                traffic = None
        sf_txt = f"SF{args.sf:g}" if not args.rows else f"{n_total} rows"
        out = {
            "metric": "rows/s TPC-H Q1 hash-agg", "value": value, "unit": "rows/s", "n_gpus": world, "steps": args.steps,
            "rccl_ranks_seen": ranks_seen if args.backend == "nccl" else 0, "ranks_seen": ranks_seen, "distinct_devices_seen": devices_seen,
            "backend": args.backend if world > 1 else None,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong",   # one SF100 table whatever N: the rows per GPU shrink as N grows
            "vs_baseline": None, "dtype": "i64/i128 decimal", "data": "synthetic" + (" (FUNCTIONAL CHECK: ranks share one GPU, not a measurement)" if args.share_gpu else ""),
            "config": {"workload": f"TPC-H Q1 hash-aggregation, {sf_txt} synthetic lineitem ({n_total} rows, 68 B/row, resident in HBM), "
                                   f"fused filter+decimal maps+group-by ({'generic register program through dbhip_groupby_add_block_program' if args.headline == 'program' else 'hand-written dbhip_q1_fused'}), "
                                   + (f"row-range sharded over {world} GPUs ({n} rows on rank 0), {exchange} of partial states over RCCL + final merge"
                                      if world > 1 else "1 MI355X"),
                       "rows_total": n_total, "rows_per_rank": n, "groups": n_groups, "exchange": exchange or None,
                       "generate_seconds": gen_s},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src, "kernel": dominant,
                         "kernel_ms": kernel_ms, "kernel_ms_min_median_max": [float(np.min(kms)), float(np.median(kms)), float(np.max(kms))] if kms else None,
                         "algorithmic_bytes_per_launch": n * BYTES_PER_ROW},
            "headline_path": {"what": "dbhip_groupby_add_block_program (generic fused program, PREPAREd)" if args.headline == "program" else "dbhip_q1_fused (hand-written)",
                              "prepare_ms_cold_or_cached": prepare_ms},
            "hand_written_kernel" if args.headline == "program" else "generic_program_kernel": other,
            "cpu_baseline": cpu,
            "exchange_impl": (args.exchange_impl if world > 1 else None),
            "exchange_impl_other": other_impl,
            "multi_gpu_readiness": readiness,
            "q1_operator_plan": opplan,
            "q1_block_size_sweep": blocks,
            "q3_sf100": q3,
            "distributed_plan_stages": plans,
            "scan_side_zstd": scan,
            "ann": ann,
        }
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def bench_exchange_overhead(li, n_shard, tpch, D, L, check, steps):
    """What one rank of the 8-GPU strong-scaling job does per step, on ONE GPU: the fused kernel over SF/8 rows, then the whole
    block protocol of the hash-partitioned exchange through the C-ABI's communicator (dbhip_groupby_exchange_alltoall on a local
    world of one: partition by hash % world -> transfer (here a copy) -> rebuild the table). exchange_overhead_ms = everything in
    a step that is not q1_fused_kernel; >= 6x at 8 GPUs leaves 0.27 ms for it. The xGMI transfer of the
    (~ KB-sized) blocks itself is not in this number."""
    import ctypes as _C
    shard = li.slice(0, n_shard)
    g = D.GroupBy.q1()
    comm = D.Comm.local()
    kms, wall = [], []

    def step():
        g.reset()
        D.q1_fused(g, shard.qty, shard.price, shard.disc, shard.tax, shard.rf, shard.ls, shard.ship, tpch.Q1_CUTOFF)
        ms = _C.c_float()
        check(L.dbhip_last_kernel_ms(_C.byref(ms)))
        comm.exchange_alltoall(g, 256)
        return ms.value
    for _ in range(3):
        step()
    check(L.dbhip_stream_sync(None))
    t0 = time.perf_counter()
    for _ in range(steps):
        kms.append(step())
    check(L.dbhip_stream_sync(None))
    step_ms = (time.perf_counter() - t0) * 1e3 / steps
    kernel_ms = sum(kms) / len(kms)
    comm.destroy()
    return {"rows_per_rank_at_8_gpus": n_shard, "step_ms": step_ms, "kernel_ms": kernel_ms, "exchange_overhead_ms": step_ms - kernel_ms,
            "budget_ms_for_6x_at_8_gpus": 0.27,
            "what": "reset + q1_fused + dbhip_groupby_exchange_alltoall (partition_blocks -> copy -> replace_with_blocks) on a local world of one"}


def _timed_ms(fn, L, check, reps):
    """wall-clock ms of fn() between two drains of the library stream (fn may synchronise internally)"""
    ts = []
    for _ in range(reps):
        check(L.dbhip_stream_sync(None))
        c0 = time.perf_counter()
        fn()
        check(L.dbhip_stream_sync(None))
        ts.append((time.perf_counter() - c0) * 1e3)
    return ts


def bench_operator_plan(li, tpch, D, L, check, fused_result, fused_kernel_ms):
    """The same Q1 through the GENERIC operator kernels — what the physical plan dispatches to without a query-specific
    kernel. Two plans, both bit-exact against the fused kernel:
      literal   the reference's operator-at-a-time shape: cmp -> filter_select -> take x6 -> 4 decimal maps -> add_block
      pushdown  cmp -> Bitmap; the decimal maps and the partial aggregation read the unfiltered columns and the Bitmap
                (tpch.q1_operator_pushdown: dbhip_groupby_add_block_filtered)
      fused_program  the binding flattens the predicate and the maps into ONE register program and the GENERIC fused
                filter -> map -> partial-aggregate kernel runs it (tpch.q1_fused_program: dbhip_groupby_add_block_program) —
                no query-specific device code; the pipeline is PREPAREd first (dbhip_groupby_prepare_program: the library
                specialises the kernel for the program through hiprtc, reported as `prepare_ms`; without PREPARE the compile runs in the
                background and the first launches go through the interpreter: 4.6 ms per 60 M rows)"""
    n = li.n
    out = {}
    plans = [("literal", tpch.q1_operator_at_a_time)]
    if hasattr(tpch, "q1_operator_pushdown"):
        plans.append(("pushdown", tpch.q1_operator_pushdown))
    if hasattr(tpch, "q1_fused_program"):
        plans.append(("fused_program", tpch.q1_fused_program))
    for name, fn in plans:
        try:
            extra = {}
            if name == "fused_program":
                t0 = time.perf_counter()
                fn(li, prepare=True)
                extra["prepare_ms"] = (time.perf_counter() - t0) * 1e3
            g = fn(li)  # warm-up: allocations land in the block cache
            same = tpch.q1_rows(g) == fused_result
            if name != "fused_program":
                # steady state of a plain add_block with a handful of groups = the run-time specialised few-groups kernel, which a
                # cold kernel cache compiles in the BACKGROUND (the block that asked takes the LDS path): warm up until
                # the plan's aggregation really goes through it (bounded), and say which kernel the timed runs used
                import ctypes as _C
                st = (_C.c_uint64 * 3)()
                t0 = time.perf_counter()
                used = False
                while not used and time.perf_counter() - t0 < 20.0:
                    L.dbhip_fagg_stats(st)
                    before = st[0]
                    fn(li)
                    L.dbhip_fagg_stats(st)
                    used = st[0] > before
                    if not used:
                        time.sleep(0.25)
                extra["aggregation_kernel"] = "fagg_jit (run-time specialised)" if used else "LDS pre-aggregation (no specialised kernel within 20 s)"
                extra["kernel_cache_warmup_s"] = time.perf_counter() - t0
            ts = _timed_ms(lambda: fn(li), L, check, 3)
            ms = min(ts)
            if name == "fused_program":   # the plan is ONE launch: its kernel time by HIP events, next to the hand-written kernel's
                import ctypes as _C
                kms = _C.c_float()
                if L.dbhip_last_kernel_ms(_C.byref(kms)) == 0:
                    extra["kernel_ms"] = kms.value
                    extra["kernel_hbm_frac"] = n * BYTES_PER_ROW / (kms.value * 1e-3) / 1e9 / HBM_PEAK_GBS
                    extra["kernel_slowdown_vs_fused_kernel"] = kms.value / fused_kernel_ms if fused_kernel_ms else None
            out[name] = {"ms": ms, "all_ms": ts, "rows_per_s": n / (ms * 1e-3), "hbm_frac_algorithmic": n * BYTES_PER_ROW / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "slowdown_vs_fused_kernel": ms / fused_kernel_ms if fused_kernel_ms else None, "equals_fused_result": bool(same), **extra}
            assert same, f"operator plan '{name}' differs from the fused kernel"
        except Exception as e:  # noqa: BLE001 — a plan that cannot run is reported, not hidden
            out[name] = {"error": repr(e)}
        check(L.dbhip_trim())
    # a Q1 variant whose scales force a rescale (a rounding multiply and a decimal divide per row) stays ONE fused program since round 4
    if hasattr(tpch, "q1_rescale_fused"):
        try:
            import ctypes as _C
            plan = tpch.q1_rescale_program(li)
            t0 = time.perf_counter()
            tpch.q1_rescale_fused(li, prepare=True, plan=plan)
            prep = (time.perf_counter() - t0) * 1e3
            g = tpch.q1_rescale_fused(li, plan=plan)
            ngroups = len(g.result())
            ts = _timed_ms(lambda: tpch.q1_rescale_fused(li, plan=plan), L, check, 3)
            kms = _C.c_float()
            L.dbhip_last_kernel_ms(_C.byref(kms))
            bytes_per_row = 4 + 8 + 8 + 8 + 2
            out["fused_program_rescale"] = {
                "what": "sum(qty), sum(price{15,8} * disc{15,8}) [rounding multiply], sum(qty / price{15,8}) [decimal divide], count(*) "
                        "WHERE shipdate <= cutoff GROUP BY returnflag, linestatus — one dbhip_groupby_add_block_program launch",
                "ms": min(ts), "all_ms": ts, "kernel_ms": kms.value, "prepare_ms": prep, "groups": ngroups,
                "algorithmic_bytes_per_row": bytes_per_row, "hbm_frac_algorithmic": n * bytes_per_row / (min(ts) * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "parity": "tests/test_gpu_fused.py::test_fused_aggregation_over_a_rescaling_q1_variant (oracle), DIV_CASES"}
        except Exception as e:  # noqa: BLE001
            out["fused_program_rescale"] = {"error": repr(e)}
        check(L.dbhip_trim())
    return out


def bench_block_sizes(cpu):
    """The drop-in at the block size the reference hands over: the same Q1 program fed through the C-ABI as blocks of
    65,536 (max_block_size, settings_default.rs:142-148) / 262,144 / 1 Mi / 16 Mi / all rows, from 1 and from 8 host threads with their
    own streams and partial tables — synchronous calls and the pipelined table (dbhip_groupby_set_pipelined) side by side. Runs
    databend_amd/host/block_sweep (C++: a Python loop would measure ctypes) in a process of its own, outside every timed region;
    every line's result equals the whole-table call's. The CPU baseline runs 65,536-row blocks too: quoted beside it."""
    exe = os.path.join(ROOT, "databend_amd", "host", "block_sweep")
    if not os.path.exists(exe):
        return {"skipped": "databend_amd/host/block_sweep is not built"}
    import subprocess
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        outp = os.path.join(td, "sweep.json")
        try:
            r = subprocess.run([exe, "--only-q1", "--rows", str(512 << 20), "--out", outp], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=300)
        except subprocess.TimeoutExpired:
            return {"skipped": "block_sweep timed out"}
        if r.returncode != 0 or not os.path.exists(outp):
            return {"failed": r.stderr.decode("utf-8", "replace")[-400:]}
        j = json.load(open(outp))
    lines = [{k: ln[k] for k in ("op", "block_rows", "threads", "g_rows_per_s", "us_per_call_per_thread", "host_us_inside_call", "equals_whole_table")} for ln in j["lines"]]
    whole = max((ln["g_rows_per_s"] for ln in lines if ln["block_rows"] >= j["rows"] and ln["op"].startswith("q1_")), default=None)
    return {"rows": j["rows"], "whole_table_g_rows_per_s": whole, "lines": lines,
            "cpu_baseline_g_rows_per_s_at_65536_row_blocks": (cpu["value"] / 1e9 if cpu else None), "cpu_threads": (cpu["cores"] if cpu else None),
            "what": "rows/s of TPC-H Q1 (same program as the headline) when the table arrives as blocks of `block_rows` rows on `threads` host threads; "
                    "q1_sync = one synchronous dbhip_groupby_add_block_program per block, q1_pipelined = the same call on a pipelined table "
                    "(blocks queued, up to 512 / 32 Mi rows per launch, one checkpoint at the end; 512 Mi rows so that every one of 8 threads streams several launches)"}


def cpu_baseline(args, li, n, tpch, result):
    """Reference-shaped CPU Q1 on the host cores (kind "port": oracle/q1_typed.c), timed on a prefix sample, and used as the
    checker: device == CPU on the sample, and on the WHOLE shard merged chunk by chunk (sums and counts add)."""
    from tests import oracle_lib
    cn = min(args.cpu_rows, n) if args.cpu_rows else n
    cn &= ~3
    host = li.host(0, cn)
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    # the box may expose more logical CPUs than its cgroup lets run: pick the fastest thread count
    best = (0.0, 1)
    probe_n = min(cn, 8_000_000)
    for tcount in sorted({avail, max(avail // 2, 1), 64, 32, 16, 8}):
        if tcount > avail:
            continue
        c0 = time.perf_counter()
        oracle_lib.q1_run(host, tpch.Q1_CUTOFF, threads=tcount, n=probe_n, typed=True)
        rate = probe_n / (time.perf_counter() - c0)
        if rate > best[0]:
            best = (rate, tcount)
    cores = best[1]
    reps, cdt, cres = 0, 0.0, None
    while cdt < args.cpu_seconds and reps < 512:  # ~10 s of wall time on the chosen cores, whole passes only
        c0 = time.perf_counter()
        cres = oracle_lib.q1_run(host, tpch.Q1_CUTOFF, threads=cores, n=cn, typed=True)
        cdt += time.perf_counter() - c0
        reps += 1
    # parity on the sample: the device over the same prefix
    got = tpch.q1_rows(tpch.q1_fused(li.slice(0, cn)))
    assert got == cres, "GPU result on the sample differs from the CPU restatement"
    checked_rows = cn
    if not args.no_verify_full and cn < n:
        total = {}
        c0 = time.perf_counter()
        step = 1 << 26
        for a in range(0, n, step):
            b = min(a + step, n)
            part = oracle_lib.q1_run(li.host(a, b), tpch.Q1_CUTOFF, threads=cores, typed=True)
            for k, v in part.items():
                acc = total.setdefault(k, dict.fromkeys(v, 0))
                for f, x in v.items():
                    acc[f] += int(x)
        # sums wrap like the device's: i64 states wrap at 2^64, i128 states at 2^128
        def wrap(v, bits):
            v &= (1 << bits) - 1
            return v - (1 << bits) if v >> (bits - 1) else v
        for k, v in total.items():
            for f in ("sum_qty", "sum_base_price", "sum_disc"):
                v[f] = wrap(v[f], 64)
            for f in ("sum_disc_price", "sum_charge"):
                v[f] = wrap(v[f], 128)
        assert total == result, "GPU result on the whole table differs from the CPU restatement (merged over chunks)"
        checked_rows = n
        full_s = time.perf_counter() - c0
    else:
        full_s = 0.0
        if cn == n:
            assert cres == result, "GPU result differs from the CPU restatement"
    return {"value": cn * reps / cdt, "unit": "rows/s", "cores": cores, "kind": "port",
            "sample": f"{reps} passes over the first {cn} rows of the same lineitem table, {cores} threads x 65536-row "
                      f"blocks (filter->take->decimal maps->partial AggregateHashTable->final merge), the reference's pipeline "
                      f"with the column types fixed at compile time (oracle/q1_typed.c, gcc -O2 -march=native; "
                      f"results identical to the generic restatement oracle/oracle.c, which is ~14x slower per row)",
            "device_equals_cpu_on_rows": checked_rows, "full_check_seconds": full_s}


def bench_q3(args, torch, tpch, D, L, check):
    """BASELINE configs[2]: TPC-H Q3 at SF100 on one GPU, the operator-at-a-time plan over the C-ABI (tools/bench_q3.py's
    tables and independent torch statement of the query)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_q3 as BQ
    t0 = time.perf_counter()
    src = BQ.Q3Torch(args.q3_sf)
    torch.cuda.synchronize()
    gen_s = time.perf_counter() - t0
    t = src.table()
    stats = {}
    got = tpch.q3_operator_at_a_time(t, stats=stats)  # warm-up
    ts = _timed_ms(lambda: tpch.q3_operator_at_a_time(t), L, check, 3)
    exp, ngroups, njoined = src.torch_q3(tpch.Q3_DATE, 10)
    ok = [(r[1], r[2]) for r in got] == [(r[1], r[2]) for r in exp] and sorted(got) == sorted(exp) and ngroups == stats["groups"] \
        and njoined == stats["orders_joined"]
    assert ok, "Q3 result differs from the independent torch statement"
    ms = min(ts)
    streamed, nl = 24 * src.nc + 24 * src.no + 28 * src.nl, src.nl
    gather = BQ.gather_bytes(stats)
    del src, t
    check(L.dbhip_trim())
    return {"workload": f"TPC-H Q3 SF{args.q3_sf:g}, 1 MI355X, operator-at-a-time over the C-ABI (filter Bitmaps as probe-key validity -> 2 hash joins "
                        f"-> decimal maps -> 3-key group-by -> ORDER BY revenue DESC, o_orderdate LIMIT 10)",
            "ms": ms, "all_ms": ts, "lineitem_rows": nl, "lineitem_rows_per_s": nl / (ms * 1e-3), "streamed_bytes": streamed, "streamed_GBps": streamed / (ms * 1e-3) / 1e9,
            "frac_of_hbm_peak": streamed / (ms * 1e-3) / 8e12, "gather_bytes": gather, "stages": stats, "matches_independent_torch_statement": bool(ok),
            "cpu_oracle_check": "tools/bench_q3.py --oracle (the same tables through oracle/liboracle.so): profiles/r03_q3_sf100_oracle.json",
            "generate_seconds": gen_s}


def bench_q3_dist(args, rank, world, torch, dist, tpch, D, DX, L, check):
    """BASELINE configs[2] on N GPUs: the broadcast-join plan (databend_amd.dist.q3_broadcast_join — the filtered customers and the
    orders that joined them are all-gathered, every rank probes its own row range of orders / lineitem, the partial group states are
    exchanged by hash % world). Every rank draws the same tables (same seed) and keeps its row range of each; the result on every
    rank must equal the independent torch statement of the whole query. Time = max over the ranks of the plan's wall clock."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_q3 as BQ
    dev = torch.device("cuda", torch.cuda.current_device())
    src = BQ.Q3Torch(args.q3_sf)
    torch.cuda.synchronize()
    exp, ngroups, _ = src.torch_q3(tpch.Q3_DATE, 10)

    class Shard:
        pass
    sh = Shard()
    dec = dict(precision=15, scale=2)

    def rng(n):
        return rank * n // world, (rank + 1) * n // world
    c0, c1 = rng(src.nc); o0, o1 = rng(src.no); l0, l1 = rng(src.nl)
    from databend_amd import _lib as T
    sh.c_custkey, sh.c_mktsegment = BQ.col(src.c_custkey[c0:c1].contiguous(), T.T_I64), BQ.col(src.c_seg[c0:c1].contiguous(), T.T_STRING)
    sh.o_orderkey, sh.o_custkey = BQ.col(src.o_orderkey[o0:o1].contiguous(), T.T_I64), BQ.col(src.o_custkey[o0:o1].contiguous(), T.T_I64)
    sh.o_orderdate, sh.o_shippriority = BQ.col(src.o_orderdate[o0:o1].contiguous(), T.T_DATE), BQ.col(src.o_shipprio[o0:o1].contiguous(), T.T_I32)
    sh.l_orderkey, sh.l_shipdate = BQ.col(src.l_orderkey[l0:l1].contiguous(), T.T_I64), BQ.col(src.l_ship[l0:l1].contiguous(), T.T_DATE)
    sh.l_extendedprice, sh.l_discount = BQ.col(src.l_price[l0:l1].contiguous(), T.T_DEC64, **dec), BQ.col(src.l_disc[l0:l1].contiguous(), T.T_DEC64, **dec)
    sh.n_customer, sh.n_orders, sh.n_lineitem = c1 - c0, o1 - o0, l1 - l0
    ops = tpch.Q3DeviceOps(torch)
    got = DX.q3_broadcast_join(sh, ops, dist, torch, dev)     # warm-up
    ts = []
    for _ in range(3):
        check(L.dbhip_stream_sync(None)); torch.cuda.synchronize(); dist.barrier()
        t0 = time.perf_counter()
        got = DX.q3_broadcast_join(sh, ops, dist, torch, dev)
        check(L.dbhip_stream_sync(None)); torch.cuda.synchronize()
        t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ts.append(float(t.item()) * 1e3)
    ok = [(int(r[0]), int(r[1])) for r in got] == [(r[0], r[1]) for r in exp]
    assert ok, "distributed Q3 differs from the independent torch statement"
    ms = min(ts)
    return {"workload": f"TPC-H Q3 SF{args.q3_sf:g} over {world} GPUs: broadcast hash joins (all-gather of the filtered build sides), row-range sharded "
                        f"probe sides, partial states exchanged by hash % world", "ms": ms, "all_ms": ts, "lineitem_rows": src.nl,
            "lineitem_rows_per_s": src.nl / (ms * 1e-3), "scaling": "strong", "groups": ngroups, "matches_independent_torch_statement": bool(ok)}


def bench_ann(args, rank, world, torch, dist, D, DX, L, check):
    """Secondary headline metric (BASELINE.json configs[4]): exact cosine top-10 over `--ann-rows` x `--ann-dim` f32
    vectors, row-range sharded over the ranks, queries replicated, per-shard top-10 all-gathered over RCCL and merged.
    A step = one batch of `--ann-queries` queries through dbhip_vec_index_search on every rank + the merge."""
    import numpy as np
    from databend_amd import _lib as T
    dev = torch.device("cuda", torch.cuda.current_device())
    n_total, dim, nq, k = args.ann_rows, args.ann_dim, args.ann_queries, 10
    lo, hi = rank * n_total // world, (rank + 1) * n_total // world
    n = hi - lo
    gen = torch.Generator(device=dev)
    gen.manual_seed(5 + rank)
    base = torch.randn((n, dim), device=dev, dtype=torch.float32, generator=gen)   # N(0,1), NOT normalised
    gq = torch.Generator(device=dev)
    gq.manual_seed(505)
    queries = torch.randn((nq, dim), device=dev, dtype=torch.float32, generator=gq)
    torch.cuda.synchronize()
    ix = C.c_void_p()
    t0 = time.perf_counter()
    check(L.dbhip_vec_index_build(T.VEC_COSINE, C.c_void_p(base.data_ptr()), C.c_int64(n), dim, C.byref(ix), None))
    check(L.dbhip_stream_sync(None))
    build_s = time.perf_counter() - t0
    oi = torch.empty((nq, k), dtype=torch.int32, device=dev)
    od = torch.empty((nq, k), dtype=torch.float32, device=dev)

    def merge_dev(d_ptr, i_ptr, nq_, m, k_, out_i_ptr, out_d_ptr):
        check(L.dbhip_vec_topk_merge(C.c_void_p(d_ptr), C.c_void_p(i_ptr), C.c_int64(m), nq_, k_, C.c_void_p(out_i_ptr), C.c_void_p(out_d_ptr), None))

    def step():
        check(L.dbhip_vec_index_search(ix, C.c_void_p(queries.data_ptr()), nq, k, C.c_void_p(oi.data_ptr()), C.c_void_p(od.data_ptr()), None))
        if world == 1:
            return oi, od
        return DX.merge_shard_topk_device(oi, od, lo, k, dist, torch, lambda: check(L.dbhip_stream_sync(None)), merge_dev)

    step()
    check(L.dbhip_stream_sync(None))
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    kms = []
    for _ in range(args.ann_steps):
        res = step()
        ms = C.c_float()
        check(L.dbhip_last_kernel_ms(C.byref(ms)))
        kms.append(ms.value)
    check(L.dbhip_stream_sync(None))
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    # recall@10 of the index against the exact f32 scan (dbhip_vec_topk) of the local shard, first 128 queries, and for 32
    # of them against the CPU oracle's exact top-10 over a 200 k-row window that contains the device's answers
    m = min(128, nq)
    ei = torch.empty((m, k), dtype=torch.int32, device=dev)
    ed = torch.empty((m, k), dtype=torch.float32, device=dev)
    check(L.dbhip_vec_topk(T.VEC_COSINE, C.c_void_p(base.data_ptr()), C.c_int64(n), dim, C.c_void_p(queries.data_ptr()), m, k,
                           C.c_void_p(ei.data_ptr()), C.c_void_p(ed.data_ptr()), None))
    check(L.dbhip_stream_sync(None))
    got = oi[:m].cpu().numpy()
    exp = ei.cpu().numpy()
    recall = float(np.mean([len(set(got[i].tolist()) & set(exp[i].tolist())) / k for i in range(m)]))
    oracle_recall = ann_oracle_recall(torch, base, queries, oi, od, k) if (rank == 0 and not args.no_cpu) else None
    oracle_full = ann_oracle_recall_full(torch, base, queries, oi, k) if (rank == 0 and not args.no_cpu) else None
    check(L.dbhip_vec_index_destroy(ix))
    qps = nq * args.ann_steps / dt
    search_ms = float(np.mean(kms))
    # HBM bytes per search step from the committed PMC passes of THIS configuration (profiles/ann_traffic.json), else null
    traffic, traffic_src = None, None
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "ann_traffic.json")))
        if world == 1 and n_total == 10_000_000 and dim == 768 and nq == 10_000:
            traffic, traffic_src = tj.get("hbm_bytes_per_step"), tj.get("source")
    except Exception:
        traffic = None
    tf = 2.0 * n * dim * nq / (search_ms * 1e-3) / 1e12
    return {"metric": "ANN queries/s @ recall@10", "value": qps, "unit": "queries/s", "recall_at_10": recall,
            "recall_at_10_vs_cpu_oracle": oracle_recall, "recall_at_10_vs_cpu_oracle_full_base": oracle_full, "k": k,
            "ms_per_step": dt / args.ann_steps * 1e3, "steps": args.ann_steps, "scaling": "strong", "index_build_s": build_s,
            "config": {"workload": f"exact cosine top-10, {n_total} x {dim} f32 base (N(0,1), not normalised), {nq} queries per step, "
                                   f"row-range sharded over {world} GPU(s), bf16-MFMA pre-filter + exact f32 re-score"
                                   + (", all-gather of per-shard top-10 over RCCL + merge" if world > 1 else ""),
                       "rows_per_rank": n, "dim": dim, "queries_per_step": nq},
            "roofline": {"bound": "mfma", "achieved": tf, "peak": 2500.0, "unit": "TFLOP/s", "frac": tf / 2500.0,
                         "kernel": "bf16_filter256_kernel (+ exact seed scan, re-score, select)", "search_ms": search_ms,
                         "algorithmic_flops_per_step": 2.0 * n * dim * nq, "traffic": traffic, "traffic_unit": "HBM bytes per search step",
                         "traffic_source": traffic_src,
                         "note": "per-rank dbhip_vec_index_search time (HIP events on the library stream); peak = dense bf16 MFMA"}}


def ann_oracle_recall(torch, base, queries, oi, od, k, n_queries=32, window=1 << 17):
    """recall@10 against the CPU ORACLE (oracle.c orc_vec_distance = the reference's distance.rs restated): for the first
    32 queries the oracle scores a `window`-row slab of the base PLUS the device's own answers; the oracle's top-10 over
    that candidate set is compared with the device's (every slab row that beats a device answer costs recall).
    (A full 10 M x 768 oracle scan is ~1 minute per query on one core; the candidate set keeps the check in seconds while
    still letting the oracle overrule the device on every row it scored. The exact-scan recall above covers all rows.)"""
    import numpy as np
    from tests import oracle_lib
    O = oracle_lib.load()
    n, dim = base.shape
    w = min(window, n)
    slab = base[:w].cpu().numpy()
    hits, total = 0, 0
    for qi in range(min(n_queries, queries.shape[0])):
        q = queries[qi:qi + 1].cpu().numpy()
        dev_ids = oi[qi].cpu().numpy().astype(np.int64)
        extra = base[torch.from_numpy(dev_ids).to(base.device)].cpu().numpy()
        cand = np.ascontiguousarray(np.concatenate([slab, extra], axis=0), dtype=np.float32)
        ids = np.concatenate([np.arange(w, dtype=np.int64), dev_ids])
        out = np.empty(cand.shape[0], dtype=np.float32)
        O.orc_vec_distance(0, cand.ctypes.data_as(C.c_void_p), C.c_int64(cand.shape[0]), C.c_int(dim), q.ctypes.data_as(C.c_void_p), C.c_int(1),
                           out.ctypes.data_as(C.c_void_p))
        order = np.lexsort((ids, out))
        top, seen = [], set()
        for j in order:
            if int(ids[j]) not in seen:
                seen.add(int(ids[j]))
                top.append(int(ids[j]))
            if len(top) == k:
                break
        hits += len(set(top) & set(int(x) for x in dev_ids))
        total += k
    return hits / total if total else None


def ann_oracle_recall_full(torch, base, queries, oi, k, n_queries=8, slab_rows=1 << 20):
    """recall@10 of the first 8 queries against the CPU oracle's exact top-10 over the WHOLE local base (the windowed
    check above only lets the oracle overrule the device on 1.3 % of the rows): the base is brought to the host slab by slab,
    every slab is scored by orc_vec_distance on the host cores (one thread per query: ctypes releases the GIL), a running
    top-10 per query (ties: lower id) is kept. -> {"recall": .., "queries": .., "rows_scored_per_query": n, "seconds": ..}"""
    import threading
    import numpy as np
    from tests import oracle_lib
    O = oracle_lib.load()
    n, dim = base.shape
    nq = min(n_queries, queries.shape[0])
    q_host = [np.ascontiguousarray(queries[i:i + 1].cpu().numpy(), dtype=np.float32) for i in range(nq)]
    best = [(np.empty(0, np.float32), np.empty(0, np.int64)) for _ in range(nq)]
    t0 = time.perf_counter()
    for lo in range(0, n, slab_rows):
        hi = min(n, lo + slab_rows)
        slab = np.ascontiguousarray(base[lo:hi].cpu().numpy(), dtype=np.float32)
        outs = [np.empty(hi - lo, dtype=np.float32) for _ in range(nq)]

        def score(i):
            O.orc_vec_distance(0, slab.ctypes.data_as(C.c_void_p), C.c_int64(hi - lo), C.c_int(dim), q_host[i].ctypes.data_as(C.c_void_p), C.c_int(1),
                               outs[i].ctypes.data_as(C.c_void_p))
        th = [threading.Thread(target=score, args=(i,)) for i in range(nq)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        for i in range(nq):
            d = np.concatenate([best[i][0], outs[i]])
            ids = np.concatenate([best[i][1], np.arange(lo, hi, dtype=np.int64)])
            order = np.lexsort((ids, d))[:k]
            best[i] = (d[order], ids[order])
    got = oi[:nq].cpu().numpy().astype(np.int64)
    hits = sum(len(set(best[i][1].tolist()) & set(got[i].tolist())) for i in range(nq))
    return {"recall": hits / (nq * k), "queries": nq, "rows_scored_per_query": n, "seconds": time.perf_counter() - t0}


if __name__ == "__main__":
    main()

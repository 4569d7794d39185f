"""CPU, world_size 2 over gloo: the multi-rank exchange of serialized partial aggregation states
(databend_amd/dist.py — the RCCL path of bench.py --gpus N runs exactly this code with the nccl
backend). The HIP table is replaced by a host stand-in with the same three methods
(flush_serialized / reset / merge_serialized); the stand-in merges rows word-wise like the device
merge does for COUNT / wrapping SUM states (gb_layout.h)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from databend_amd import dist as DX

W = 4  # [key][hash][sum][count]
HASH_WORD = 1
M64 = (1 << 64) - 1


def mix(x):
    """the integer group hash (aggregate/group_hash.rs:555-570), python ints"""
    x &= M64
    x ^= x >> 32
    x = (x * 0xD6E8FEB86659FD93) & M64
    x ^= x >> 32
    x = (x * 0xD6E8FEB86659FD93) & M64
    x ^= x >> 32
    return x


class HostTable:
    def __init__(self):
        self.groups = {}

    def add(self, keys, vals):
        for k, v in zip(keys.tolist(), vals.tolist()):
            s = self.groups.setdefault(k, [0, 0])
            s[0] = (s[0] + v) & M64
            s[1] += 1

    def flush_serialized(self):
        rows = np.zeros((len(self.groups), W), dtype=np.uint64)
        for i, (k, s) in enumerate(sorted(self.groups.items())):
            rows[i] = (k & M64, mix(k), s[0], s[1])
        return rows

    def reset(self, stream=None):
        self.groups = {}

    def merge_serialized(self, rows):
        for k, h, s, c in np.asarray(rows, dtype=np.uint64).tolist():
            assert h == mix(k)
            st = self.groups.setdefault(k, [0, 0])
            st[0] = (st[0] + s) & M64
            st[1] += c


    # --- the device-block protocol of dbhip_groupby_flush_block / merge_blocks, on host memory ---
    def row_bytes(self):
        return W * 8

    @staticmethod
    def _view(ptr, words):
        import ctypes
        return np.ctypeslib.as_array((ctypes.c_uint64 * words).from_address(ptr))

    def flush_block(self, ptr, max_rows, stream=None):
        block = self._view(ptr, (max_rows + 1) * W).reshape(max_rows + 1, W)
        rows = self.flush_serialized()
        block[0] = 0
        if rows.shape[0] > max_rows:
            block[0, 0] = M64
        else:
            block[0, 0] = rows.shape[0]
            block[1:1 + rows.shape[0]] = rows

    def merge_blocks(self, ptr, n_blocks, max_rows, skip, stream=None):
        blocks = self._view(ptr, n_blocks * (max_rows + 1) * W).reshape(n_blocks, max_rows + 1, W)
        # every header counts, the caller's own block included (dbhip_groupby_merge_blocks): the owner of an overflowed
        # block must leave the fixed-size path together with the ranks that see its header
        if any(int(blocks[b, 0, 0]) == M64 for b in range(n_blocks)):
            raise OverflowError("block overflow")   # decided before the table is touched
        for b in range(n_blocks):
            if b != skip:
                self.merge_serialized(blocks[b, 1:1 + int(blocks[b, 0, 0])])

    # --- dbhip_groupby_partition_blocks / replace_with_blocks / flush_partitioned on host memory ---
    def num_groups(self):
        return len(self.groups)

    def partition_blocks(self, ptr, n_buckets, max_rows, stream=None):
        blocks = self._view(ptr, n_buckets * (max_rows + 1) * W).reshape(n_buckets, max_rows + 1, W)
        parts = DX.route_rows_by_hash(self.flush_serialized(), HASH_WORD, n_buckets)
        over = any(p.shape[0] > max_rows for p in parts)
        for b, p in enumerate(parts):
            blocks[b, 0] = 0
            blocks[b, 0, 0] = M64 if p.shape[0] > max_rows else p.shape[0]
            blocks[b, 0, 1] = 1 if over else 0
            if p.shape[0] <= max_rows:
                blocks[b, 1:1 + p.shape[0]] = p

    def replace_with_blocks(self, ptr, n_blocks, max_rows, stream=None):
        blocks = self._view(ptr, n_blocks * (max_rows + 1) * W).reshape(n_blocks, max_rows + 1, W)
        if any(int(blocks[b, 0, 0]) == M64 or int(blocks[b, 0, 1]) != 0 for b in range(n_blocks)):
            raise OverflowError("block overflow")   # decided before the table is touched
        self.reset()
        for b in range(n_blocks):
            self.merge_serialized(blocks[b, 1:1 + int(blocks[b, 0, 0])])

    def flush_partitioned(self, n_buckets, ptr, max_rows, stream=None):
        parts = DX.route_rows_by_hash(self.flush_serialized(), HASH_WORD, n_buckets)
        allrows = np.concatenate(parts, axis=0)
        if allrows.shape[0]:
            self._view(ptr, allrows.shape[0] * W).reshape(-1, W)[:] = allrows
        return [p.shape[0] for p in parts]

    def merge_serialized_device(self, ptr, n_rows, stream=None):
        if n_rows:
            self.merge_serialized(self._view(ptr, n_rows * W).reshape(n_rows, W).copy())


def shard(rank, world, n, card, seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    keys = rng.integers(0, card, n, dtype=np.int64)
    vals = rng.integers(0, 1 << 40, n, dtype=np.int64)
    lo, hi = rank * n // world, (rank + 1) * n // world
    return keys, vals, lo, hi


def worker(rank, world, port, mode, n, card, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        keys, vals, lo, hi = shard(rank, world, n, card, 11)
        t = HostTable()
        t.add(keys[lo:hi], vals[lo:hi])
        if mode == "fixed":
            DX.exchange_partials_fixed(t, dist, torch, torch.device("cpu"), max_rows=64)
        elif mode == "device":
            DX.exchange_partials_device(t, dist, torch, torch.device("cpu"), max_rows=64,
                                        capacity_error=lambda e: isinstance(e, OverflowError))
        elif mode == "alltoall_device":
            DX.exchange_partials_alltoall_device(t, dist, torch, torch.device("cpu"), max_rows=64,
                                                 capacity_error=lambda e: isinstance(e, OverflowError))
        elif mode == "device_asym":
            # asymmetric cardinality: only rank 0 overflows the 64-row block — BOTH ranks must leave the fixed-size path
            if rank == 1:
                t.reset()
                t.add(np.arange(5, dtype=np.int64), np.arange(5, dtype=np.int64))
            DX.exchange_partials_device(t, dist, torch, torch.device("cpu"), max_rows=64,
                                        capacity_error=lambda e: isinstance(e, OverflowError))
        else:
            DX.exchange_partials(t, dist, torch, torch.device("cpu"), mode=mode, hash_word=HASH_WORD)
        q.put((rank, {k: tuple(v) for k, v in t.groups.items()}))
    finally:
        dist.barrier()
        dist.destroy_process_group()


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def run(mode, n, card, world=2):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=worker, args=(r, world, port, mode, n, card, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    keys, vals, _, _ = shard(0, world, n, card, 11)
    full = HostTable()
    full.add(keys, vals)
    return got, {k: tuple(v) for k, v in full.groups.items()}


@pytest.mark.parametrize("n,card", [(5000, 4), (20000, 3000)])
def test_allgather_exchange_every_rank_gets_the_global_result(n, card):
    got, exp = run("allgather", n, card)
    assert got[0] == exp and got[1] == exp


@pytest.mark.parametrize("n,card", [(5000, 4), (3, 1), (20000, 3000)])
def test_fixed_block_exchange_one_collective(n, card):
    """bench.py's Q1 exchange: one fixed-size all-gather (row 0 of every block = row count); a rank with more rows
    than the block holds makes EVERY rank fall back to the variable-length path in the same step."""
    got, exp = run("fixed", n, card)
    assert got[0] == exp and got[1] == exp


@pytest.mark.parametrize("n,card", [(5000, 4), (3, 1), (20000, 3000)])
def test_device_block_exchange_protocol(n, card):
    """bench.py --gpus N: flush_block -> one all_gather_into_tensor -> merge_blocks(skip = own rank). The stand-in
    implements the block layout of include/dbhip.h on host memory; 3000 groups overflow the 64-row block on both
    ranks, which every rank sees in the gathered headers and answers with the variable-length exchange."""
    got, exp = run("device", n, card)
    assert got[0] == exp and got[1] == exp


@pytest.mark.parametrize("n,card", [(5000, 4), (20000, 3000), (10, 1)])
def test_alltoall_exchange_partitions_groups_by_hash(n, card):
    got, exp = run("alltoall", n, card)
    # every group is finalised on exactly the rank hash % world names (payload.rs:571-577)
    for r in (0, 1):
        for k in got[r]:
            assert mix(k) % 2 == r
    merged = dict(got[0])
    assert not (set(got[0]) & set(got[1]))
    merged.update(got[1])
    assert merged == exp


@pytest.mark.parametrize("n,card", [(5000, 4), (20000, 3000), (10, 1), (3000, 100)])
def test_device_alltoall_exchange_protocol(n, card):
    """bench.py --gpus N --exchange alltoall: partition_blocks -> ONE all_to_all_single of equal blocks ->
    replace_with_blocks; 3000 groups overflow the 64-row blocks, every rank sees the senders' flags and takes the
    variable-length all-to-all (flush_partitioned -> counts -> rows) in the same step."""
    got, exp = run("alltoall_device", n, card)
    for r in (0, 1):
        for k in got[r]:
            assert mix(k) % 2 == r
    assert not (set(got[0]) & set(got[1]))
    merged = dict(got[0])
    merged.update(got[1])
    assert merged == exp


def test_device_block_exchange_with_asymmetric_overflow():
    """rank 0 holds 3000 groups (its block overflows), rank 1 holds 5. The owner of the overflowed block must
    see the overflow in its OWN header and fall back together with rank 1 — otherwise rank 1 enters the all-gather of
    the variable-length path alone and the job hangs."""
    n, card = 20000, 3000
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=worker, args=(r, 2, port, "device_asym", n, card, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    keys, vals, lo, hi = shard(0, 2, n, card, 11)
    full = HostTable()
    full.add(keys[lo:hi], vals[lo:hi])
    full.add(np.arange(5, dtype=np.int64), np.arange(5, dtype=np.int64))
    exp = {k: tuple(v) for k, v in full.groups.items()}
    assert got[0] == exp and got[1] == exp


def test_route_rows_by_hash_is_a_partition():
    rows = np.zeros((1000, W), dtype=np.uint64)
    rows[:, 0] = np.arange(1000)
    rows[:, 1] = [mix(int(k)) for k in range(1000)]
    for world in (1, 2, 3, 8):
        parts = DX.route_rows_by_hash(rows, HASH_WORD, world)
        assert sum(len(p) for p in parts) == 1000
        for r, p in enumerate(parts):
            assert np.all(p[:, 1] % np.uint64(world) == r)
    empty = DX.route_rows_by_hash(np.zeros((0, W), dtype=np.uint64), HASH_WORD, 2)
    assert [p.shape for p in empty] == [(0, W), (0, W)]


def numpy_topk_merge(dists, ids, k):
    """stand-in for dbhip_vec_topk_merge on CPU: ascending distance, NaN last, ties by lower id, 0xFFFFFFFF = empty"""
    out_i = np.full((dists.shape[0], k), 0xFFFFFFFF, np.uint32)
    out_d = np.full((dists.shape[0], k), np.inf, np.float32)
    for q in range(dists.shape[0]):
        cand = [(np.inf if np.isnan(d) else float(d), int(i)) for d, i in zip(dists[q], ids[q]) if i != 0xFFFFFFFF]
        cand.sort()
        for j, (d, i) in enumerate(cand[:k]):
            out_i[q, j], out_d[q, j] = i, d
    return out_i, out_d


def ann_worker(rank, world, port, n, dim, nq, k, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.Generator(np.random.PCG64(21))
        base = rng.standard_normal((n, dim)).astype(np.float32)
        qs = rng.standard_normal((nq, dim)).astype(np.float32)
        lo, hi = rank * n // world, (rank + 1) * n // world
        d = 1.0 - (qs @ base[lo:hi].T) / (np.linalg.norm(qs, axis=1)[:, None] * np.linalg.norm(base[lo:hi], axis=1)[None, :])
        kk = min(k, hi - lo)
        order = np.argsort(d, axis=1, kind="stable")[:, :kk]
        idx = np.full((nq, k), 0xFFFFFFFF, np.uint32)
        dst = np.full((nq, k), np.inf, np.float32)
        idx[:, :kk] = order
        dst[:, :kk] = np.take_along_axis(d, order, axis=1)
        gi, gd = DX.merge_shard_topk(idx, dst, lo, k, dist, torch, torch.device("cpu"), numpy_topk_merge)
        q.put((rank, gi.tolist(), gd.tolist()))
    finally:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.parametrize("n,k", [(500, 10), (7, 5)])
def test_sharded_ann_topk_merge_equals_the_unsharded_topk(n, k):
    """Row-range sharded ANN: all-gather of per-shard top-k + merge == top-k of the whole base,
    with global row ids; a shard with fewer than k rows contributes empty slots."""
    world, dim, nq = 2, 16, 9
    ctx = mp.get_context("spawn")
    qq = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=ann_worker, args=(r, world, port, n, dim, nq, k, qq)) for r in range(world)]
    for p in procs:
        p.start()
    got = {r: (i, d) for r, i, d in (qq.get(timeout=120) for _ in range(world))}
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    rng = np.random.Generator(np.random.PCG64(21))
    base = rng.standard_normal((n, dim)).astype(np.float32)
    qs = rng.standard_normal((nq, dim)).astype(np.float32)
    d = 1.0 - (qs @ base.T) / (np.linalg.norm(qs, axis=1)[:, None] * np.linalg.norm(base, axis=1)[None, :])
    exp = np.argsort(d, axis=1, kind="stable")[:, :min(k, n)]
    for r in range(world):
        gi = np.array(got[r][0], np.uint32)
        assert np.array_equal(gi[:, :min(k, n)], exp.astype(np.uint32))
        assert (gi[:, min(k, n):] == 0xFFFFFFFF).all()


# ---- broadcast hash join across ranks: TPC-H Q3 over row-range shards (databend_amd.dist.q3_broadcast_join) ----
class Q3HostTable:
    """stand-in for the device group-by table of Q3: rows [l_orderkey][o_orderdate | o_shippriority << 32][hash][revenue]"""
    W3, HASH = 4, 2

    def __init__(self):
        self.groups = {}

    def add(self, ok, od, sp, rev):
        for k, d, p, r in zip(ok.tolist(), od.tolist(), sp.tolist(), rev.tolist()):
            key = (k, d, p)
            self.groups[key] = self.groups.get(key, 0) + r

    def flush_serialized(self):
        rows = np.zeros((len(self.groups), self.W3), dtype=np.uint64)
        for i, ((k, d, p), r) in enumerate(sorted(self.groups.items())):
            rows[i] = (k & M64, (d & 0xFFFFFFFF) | (p << 32), mix(k), r & M64)
        return rows

    def reset(self, stream=None):
        self.groups = {}

    def merge_serialized(self, rows):
        for k, dp, h, r in np.asarray(rows, dtype=np.uint64).tolist():
            assert h == mix(k)
            key = (k, dp & 0xFFFFFFFF, dp >> 32)
            self.groups[key] = self.groups.get(key, 0) + r


class Q3NumpyOps:
    """the single-node operators of the plan on numpy (what databend_amd.tpch.Q3DeviceOps does through the C-ABI)"""

    def __init__(self, segment, date):
        from databend_amd import tpch
        self.seg_code, self.date = tpch.SEGMENTS.index(segment.encode()), date

    def filter_customers(self, s):
        c = s["customer"]
        return [torch.from_numpy(c["c_custkey"][c["seg_code"] == self.seg_code].copy())]

    def join_orders(self, all_ck, s):
        o = s["orders"]
        keep = (o["o_orderdate"] < self.date) & np.isin(o["o_custkey"], all_ck.numpy())
        return [torch.from_numpy(o[k][keep].copy()) for k in ("o_orderkey", "o_orderdate", "o_shippriority")]

    def aggregate_lineitem(self, okey, odate, oprio, s):
        li = s["lineitem"]
        okey, odate, oprio = okey.numpy(), odate.numpy(), oprio.numpy()
        order = np.argsort(okey, kind="stable")
        sk = okey[order]
        keep = li["l_shipdate"] > self.date
        pos = np.searchsorted(sk, li["l_orderkey"])
        pos[pos >= len(sk)] = 0
        hit = keep & (len(sk) > 0) & (sk[pos] == li["l_orderkey"]) if len(sk) else np.zeros(len(keep), bool)
        b = order[pos[hit]]
        t = Q3HostTable()
        t.add(li["l_orderkey"][hit], odate[b], oprio[b], li["l_extendedprice"][hit] * (100 - li["l_discount"][hit]))
        return t

    def exchange(self, table, dist_, device):
        return DX.exchange_partials(table, dist_, torch, device, mode="alltoall", hash_word=Q3HostTable.HASH)

    def top_rows(self, table, limit):
        rows = sorted(((k, r, d, p) for (k, d, p), r in table.groups.items()), key=lambda r: (-r[1], r[2], r[0]))
        return rows[:limit] if limit else rows


def q3_host_tables(sf, seed):
    from databend_amd import tpch
    host = tpch.gen_q3(sf, seed=seed)
    # the segment code per customer (the views hold the strings inline: length in the low 4 bytes, then the bytes)
    v = np.ascontiguousarray(host["customer"]["c_mktsegment"]).view(np.uint8).reshape(-1, 16)
    names = [bytes(r[4:4 + int(r[0])]) for r in v]
    host["customer"]["seg_code"] = np.array([tpch.SEGMENTS.index(nm) for nm in names], dtype=np.int64)
    return host


def q3_shard(host, rank, world, skew):
    """row-range shards, cut at DIFFERENT fractions per table (skew) so that an order's lines and its customer live on
    different ranks"""
    out = {}
    for ti, (name, cols) in enumerate(host.items()):
        n = len(next(iter(cols.values())))
        cuts = [0] + [min(n, int(n * ((r + 1) / world) ** (1.0 + skew * ti))) for r in range(world - 1)] + [n]
        out[name] = {k: v[cuts[rank]:cuts[rank + 1]] for k, v in cols.items()}
    return out


def q3_worker(rank, world, port, sf, seed, limit, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from databend_amd import tpch
        host = q3_host_tables(sf, seed)
        got = DX.q3_broadcast_join(q3_shard(host, rank, world, 0.4), Q3NumpyOps(tpch.Q3_SEGMENT, tpch.Q3_DATE), dist, torch,
                                   torch.device("cpu"), limit=limit)
        q.put((rank, got))
    finally:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.parametrize("world,sf,limit", [(2, 0.004, 10), (3, 0.002, 0)])
def test_q3_broadcast_join_over_shards_equals_the_single_node_query(world, sf, limit):
    from databend_amd import tpch
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=q3_worker, args=(r, world, port, sf, 9, limit, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    host = q3_host_tables(sf, 9)
    ops = Q3NumpyOps(tpch.Q3_SEGMENT, tpch.Q3_DATE)       # the same operators on the whole tables = the single-node query
    t = ops.aggregate_lineitem(*ops.join_orders(ops.filter_customers(host)[0], host), host)
    exp = ops.top_rows(t, limit)
    assert len(exp) == (limit or len(t.groups)) and len(exp) >= 10
    # and the single-node statement agrees with the oracle's Q3 (revenue, date sequence; ties are unordered there)
    from tests import oracle_lib as O
    ref = O.q3_run({k: {c: v for c, v in cols.items() if c != "seg_code"} for k, cols in host.items()}, tpch.Q3_SEGMENT, tpch.Q3_DATE,
                   limit=limit, threads=2)
    assert [(r[1], r[2]) for r in exp] == [(r[1], r[2]) for r in ref] and sorted(exp) == sorted(ref)
    for r in range(world):
        assert got[r] == exp, f"rank {r}"


# ---- distributed sort: sample -> Bounds -> range partition -> all-to-all -> local sort (DX.range_partitioned_sort) ----
def sort_row_cmp(a, b, desc, nf):
    """SortCompare's order on host tuples (None = NULL; the NULL order does not depend on asc / desc)"""
    for x, y, d, f in zip(a, b, desc, nf):
        if x is None or y is None:
            if x is None and y is None:
                continue
            a_first = bool(f) if x is None else not f
            return -1 if a_first else 1
        if x != y:
            r = -1 if x < y else 1
            return -r if d else r
    return 0


class SortNumpyOps:
    """host stand-in of databend_amd.sort_ops.SortDeviceOps: the same three operators in plain Python"""

    def _rows(self, flat, kpos, kvpos):
        cols = [flat[p].tolist() for p in kpos]
        vals = [flat[v].tolist() if v is not None else None for v in kvpos]
        return [tuple(cols[k][i] if vals[k] is None or vals[k][i] else None for k in range(len(kpos))) for i in range(len(cols[0]))]

    def ordered_rows(self, key_cols, key_valids, desc, nulls_first):
        import functools
        flat = list(key_cols) + [v for v in key_valids if v is not None]
        kv, at = [], len(key_cols)
        for v in key_valids:
            kv.append(at if v is not None else None)
            at += 1 if v is not None else 0
        rows = self._rows(flat, list(range(len(key_cols))), kv) if len(key_cols[0]) else []
        return sorted(rows, key=functools.cmp_to_key(lambda a, b: sort_row_cmp(a, b, desc, nulls_first)))

    def partition(self, flat, kpos, kvpos, bounds, desc, nulls_first):
        n = len(flat[0])
        if not bounds or n == 0:
            return list(flat), [n]
        rows = self._rows(flat, kpos, kvpos)
        part = np.array([sum(1 for b in bounds if sort_row_cmp(b, r, desc, nulls_first) < 0) for r in rows], dtype=np.int64)
        order = torch.from_numpy(np.argsort(part, kind="stable"))
        return [c[order] for c in flat], np.bincount(part, minlength=len(bounds) + 1).tolist()

    def sort(self, flat, kpos, kvpos, desc, nulls_first):
        import functools
        if len(flat[0]) == 0:
            return list(flat)
        rows = self._rows(flat, kpos, kvpos)
        order = sorted(range(len(rows)), key=functools.cmp_to_key(lambda i, j: sort_row_cmp(rows[i], rows[j], desc, nulls_first)))
        order = torch.tensor(order, dtype=torch.int64)
        return [c[order] for c in flat]


def sort_table(seed, n, card):
    rng = np.random.default_rng(seed)
    return {"k0": rng.integers(-card, card, n).astype(np.int32), "k1": rng.integers(0, 1 << 40, n).astype(np.int64),
            "v0": (rng.random(n) > 0.15).astype(np.uint8), "pay": np.arange(n, dtype=np.int64)}


def sort_cuts(n, world, skew):
    if skew == "empty":                       # one rank holds nothing
        base = np.linspace(0, n, world).astype(np.int64)
        return [0] + base.tolist()
    return np.linspace(0, n, world + 1).astype(np.int64).tolist() if not skew else [0] + [int(n * (0.7 + 0.3 * r / (world - 1))) for r in range(world - 1)] + [n]


def sort_worker(rank, world, port, seed, n, card, skew, samples, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        t = sort_table(seed, n, card)
        cuts = sort_cuts(n, world, skew)
        sh = {k: torch.from_numpy(v[cuts[rank]:cuts[rank + 1]].copy()) for k, v in t.items()}
        cols, valids, bounds = DX.range_partitioned_sort([sh["k0"], sh["k1"], sh["pay"]], [0, 1], SortNumpyOps(), dist, torch, desc=[1, 0],
                                                         nulls_first=[1, 0], valids=[sh["v0"], None, None], samples_per_rank=samples)
        q.put((rank, [c.numpy() for c in cols], valids[0].numpy(), bounds))
    finally:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n,card,skew,samples", [(2, 3000, 50, False, 64), (3, 5000, 5, True, 128), (3, 400, 1000, "empty", 16), (2, 5, 2, False, 64)])
def test_range_partitioned_sort_concatenates_to_the_global_order(world, n, card, skew, samples):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=sort_worker, args=(r, world, port, 21, n, card, skew, samples, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(world):
        r, cols, v0, bounds = q.get(timeout=180)
        got[r] = (cols, v0, bounds)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    t = sort_table(21, n, card)
    desc, nf = [1, 0], [1, 0]
    assert all(got[r][2] == got[0][2] for r in range(world)) and len(got[0][2]) <= world - 1        # the same bounds everywhere
    k0 = np.concatenate([got[r][0][0] for r in range(world)])
    k1 = np.concatenate([got[r][0][1] for r in range(world)])
    pay = np.concatenate([got[r][0][2] for r in range(world)])
    v0 = np.concatenate([got[r][1] for r in range(world)])
    assert sorted(pay.tolist()) == list(range(n))                                                    # every row exactly once
    assert np.array_equal(t["k0"][pay], k0) and np.array_equal(t["k1"][pay], k1) and np.array_equal(t["v0"][pay], v0)   # rows stay whole
    rows = [(int(a) if v else None, int(b)) for a, b, v in zip(k0, k1, v0)]
    assert all(sort_row_cmp(rows[i], rows[i + 1], desc, nf) <= 0 for i in range(n - 1))              # rank order IS the sort order
    # the cut is the reference's (sort_spill.rs partition_point): a row's rank = the number of bounds sorting strictly before it
    at = 0
    for r in range(world):
        m = len(got[r][0][0])
        for row in rows[at:at + m]:
            assert sum(1 for b in got[0][2] if sort_row_cmp(b, row, desc, nf) < 0) == r
        at += m
    if n >= 3000:
        sizes = [len(got[r][0][0]) for r in range(world)]
        assert max(sizes) <= 1.3 * n / world, sizes                                                  # the samples balance the ranges


# ---- shuffle hash join: both sides scattered by siphash64(key) % world, joined locally (DX.shuffle_hash_join) ----
class ShuffleNumpyOps:
    """host stand-in of databend_amd.sort_ops.ShuffleDeviceOps; the routing hash is the independent Python statement of the
    reference's siphash64 (tests/siphash_ref.py)"""

    def scatter(self, flat, kpos, kvpos, world):
        from tests import siphash_ref as R
        n = len(flat[0])
        keys = flat[kpos].tolist()
        valid = flat[kvpos].tolist() if kvpos is not None else [1] * n
        dest = np.array([R.scatter_index([R.siphash64("i64", k) if v else None], world, 0) for k, v in zip(keys, valid)], dtype=np.int64)
        order = torch.from_numpy(np.argsort(dest, kind="stable"))
        return [c[order] for c in flat], np.bincount(dest, minlength=world).tolist()

    def join(self, build_flat, bk, bkv, probe_flat, pk, pkv):
        by_key = {}
        bkeys = build_flat[bk].tolist()
        bval = build_flat[bkv].tolist() if bkv is not None else [1] * len(bkeys)
        for r, (k, v) in enumerate(zip(bkeys, bval)):
            if v:
                by_key.setdefault(k, []).append(r)
        pkeys = probe_flat[pk].tolist()
        pval = probe_flat[pkv].tolist() if pkv is not None else [1] * len(pkeys)
        pi, bi = [], []
        for i, (k, v) in enumerate(zip(pkeys, pval)):
            if v:
                for r in by_key.get(k, []):
                    pi.append(i)
                    bi.append(r)
        pi, bi = torch.tensor(pi, dtype=torch.int64), torch.tensor(bi, dtype=torch.int64)
        return [c[pi] for c in probe_flat], [c[bi] for c in build_flat]


def shuffle_tables(seed, nb, npr, card):
    rng = np.random.default_rng(seed)
    return ({"k": rng.integers(0, card, nb).astype(np.int64), "v": (rng.random(nb) > 0.1).astype(np.uint8), "id": np.arange(nb, dtype=np.int64)},
            {"k": rng.integers(0, card * 2, npr).astype(np.int64), "v": (rng.random(npr) > 0.1).astype(np.uint8), "id": np.arange(npr, dtype=np.int64)})


def shuffle_worker(rank, world, port, seed, nb, npr, card, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        b, p = shuffle_tables(seed, nb, npr, card)
        bc, pc = sort_cuts(nb, world, True), sort_cuts(npr, world, False)
        bs = {k: torch.from_numpy(v[bc[rank]:bc[rank + 1]].copy()) for k, v in b.items()}
        ps = {k: torch.from_numpy(v[pc[rank]:pc[rank + 1]].copy()) for k, v in p.items()}
        out_p, out_b = DX.shuffle_hash_join([bs["k"], bs["id"]], 0, [ps["k"], ps["id"]], 0, ShuffleNumpyOps(), dist, torch,
                                            build_valids=[bs["v"], None], probe_valids=[ps["v"], None])
        q.put((rank, out_p[1].numpy(), out_b[1].numpy(), out_p[0].numpy(), out_b[0].numpy()))
    finally:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.parametrize("world,nb,npr,card", [(2, 2000, 6000, 400), (3, 900, 2500, 5000), (3, 50, 10, 3)])
def test_shuffle_hash_join_union_over_ranks_is_the_join(world, nb, npr, card):
    from tests import siphash_ref as R
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=shuffle_worker, args=(r, world, port, 77, nb, npr, card, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(world):
        r, pid, bid, pk, bk = q.get(timeout=180)
        got[r] = (pid, bid, pk, bk)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    b, p = shuffle_tables(77, nb, npr, card)
    by_key = {}
    for r in range(nb):
        if b["v"][r]:
            by_key.setdefault(int(b["k"][r]), []).append(r)
    exp = sorted((i, r) for i in range(npr) if p["v"][i] for r in by_key.get(int(p["k"][i]), []))
    pairs = sorted((int(a), int(c)) for r in range(world) for a, c in zip(got[r][0], got[r][1]))
    assert pairs == exp and (len(exp) > 0 or nb < 100)
    for r in range(world):      # every pair was produced on the rank its key hashes to, with equal keys on both sides
        assert np.array_equal(got[r][2], got[r][3])
        assert all(R.siphash64("i64", int(k)) % world == r for k in got[r][2])


# ---- aggregation without keys: one state per aggregate and rank, folded in rank order (DX.merge_single_states) ----
def single_state_inputs(rank):
    rng = np.random.default_rng(100 + rank)
    a, b, c = (rng.integers(-2**62, 2**62, 1000) for _ in range(3))
    f = rng.standard_normal(1000) * 10.0 ** rng.integers(-8, 8, 1000)
    d = [int(x) * int(y) for x, y in zip(rng.integers(-2**62, 2**62, 50), rng.integers(0, 2**60, 50))]
    return a, b, c, f, d


def single_state_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        a, b, c, f, d = single_state_inputs(rank)
        wrap = int((a.astype(np.uint64) + b.astype(np.uint64) * c.astype(np.uint64)).sum(dtype=np.uint64))      # sum(a + b * c), wrapping
        fsum = 0.0
        for x in f.tolist():
            fsum += x                                                                                          # sequential, like sum_batch
        empty = rank == 1                                                                                      # a rank without rows
        states = [("sum", "i64", wrap, True), ("sum", "f64", fsum, True), ("sum", "i128", sum(d), True), ("count", "u64", 1000, True),
                  ("min", "i64", int(a.min()), not empty), ("max", "f64", float(f.max()), not empty), ("sum", "i64", 0, False)]
        q.put((rank, DX.merge_single_states(states, dist, torch, torch.device("cpu"))))
    finally:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_single_state_aggregates_merge_in_rank_order(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=single_state_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    M64 = (1 << 64) - 1
    wrap, fs, ds, mins, maxs = 0, [], 0, [], []
    for r in range(world):
        a, b, c, f, d = single_state_inputs(r)
        wrap = (wrap + int((a.astype(np.uint64) + b.astype(np.uint64) * c.astype(np.uint64)).sum(dtype=np.uint64))) & M64
        s = 0.0
        for x in f.tolist():
            s += x
        fs.append(s)
        ds += sum(d)
        if r != 1:
            mins.append(int(a.min()))
            maxs.append(float(f.max()))
    fsum = fs[0]
    for s in fs[1:]:
        fsum += s                                                  # rank order
    ds &= (1 << 128) - 1
    exp = [(wrap - (1 << 64) if wrap >> 63 else wrap, True), (fsum, True), (ds - (1 << 128) if ds >> 127 else ds, True), (1000 * world, True),
           (min(mins), True), (max(maxs), True), (0, False)]
    for r in range(world):
        assert got[r] == exp, r


# ---- the packed collectives of the plans (round 4): ONE collective whatever the width of the block ---------------------------------
def packed_inputs(rank, world):
    rng = np.random.default_rng(900 + rank)
    counts = [int(x) for x in rng.integers(0, 700, world)]
    if rank == 1:
        counts[0] = 0                                            # nothing for rank 0 from rank 1
    n = sum(counts)
    cols = [rng.integers(0, 256, n).astype(np.uint8), rng.integers(-2**15, 2**15, n).astype(np.int16), rng.integers(-2**31, 2**31, n).astype(np.int32),
            rng.integers(-2**62, 2**62, n).astype(np.int64), rng.standard_normal(n)]
    return counts, cols


def packed_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        counts, cols = packed_inputs(rank, world)
        tcols = [torch.from_numpy(c) for c in cols]
        recv, rc = DX.alltoall_columns(tcols, counts, dist, torch)
        gathered = DX.allgather_columns(tcols, dist, torch)
        q.put((rank, ([r.numpy().copy() for r in recv], rc, [g.numpy().copy() for g in gathered])))
    finally:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_packed_collectives_move_every_column_of_a_block_at_once(world):
    """dist.alltoall_columns / allgather_columns: columns of 1 / 2 / 4 / 8-byte elements (odd row counts, an empty slice) packed into ONE
    all_to_all_single / all_gather_into_tensor each: every rank receives, per column, the slices the others addressed to it, by source
    rank; the all-gather gives every rank the rank-major concatenation."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=packed_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    inputs = [packed_inputs(r, world) for r in range(world)]
    for r in range(world):
        recv, rc, gathered = got[r]
        assert rc == [inputs[s][0][r] for s in range(world)]
        for k in range(5):
            exp = []
            for s in range(world):
                counts, cols = inputs[s]
                start = sum(counts[:r])
                exp.append(cols[k][start:start + counts[r]])
            assert np.array_equal(recv[k], np.concatenate(exp)) and recv[k].dtype == inputs[0][1][k].dtype
            assert np.array_equal(gathered[k], np.concatenate([inputs[s][1][k] for s in range(world)]))

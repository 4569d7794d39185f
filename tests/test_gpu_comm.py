"""GPU: the communicator behind the C-ABI (dbhip_comm_*, dbhip_groupby_exchange_*) on the one GPU a test box has — a world
of one, both as the local object (no RCCL) and as a REAL RCCL communicator of one rank (ncclGetUniqueId / ncclCommInitRank /
ncclAllGather / grouped ncclSend + ncclRecv / ncclAllReduce through librccl.so): the exchanges must leave the table's result
unchanged, report an overflowing block before touching the table, and the plain collectives must be identities. The N > 1
behaviour of the same block protocol is covered over gloo by tests/test_dist_gloo.py (world size 2)."""
import ctypes as C

import numpy as np
import pytest

from databend_amd import _lib as T

pytestmark = pytest.mark.gpu


def make_table(D, n_groups, n=200_000, seed=1):
    rng = np.random.default_rng(seed)
    k = rng.integers(0, n_groups, n).astype(np.int64)
    a = rng.integers(-10**6, 10**6, n).astype(np.int64)
    g = D.GroupBy([T.T_I64], [(T.AGG_SUM, T.T_I64, 0, 0, 0), (T.AGG_COUNT, 0, 0, 0, 0)])
    g.add_block([D.Column.from_numpy(k)], [D.Column.from_numpy(a), None], n)
    return g


@pytest.mark.parametrize("real_rccl", [False, True])
def test_exchanges_in_a_world_of_one_leave_the_result_unchanged(gpu, real_rccl):
    D = gpu
    comm = D.Comm(0, 1, D.Comm.unique_id()) if real_rccl else D.Comm.local()
    for groups in (4, 150):
        g = make_table(D, groups)
        before = sorted(g.result())
        comm.exchange_allgather(g, max_rows=256)      # every rank ends with the global result: here, its own
        assert sorted(g.result()) == before
        comm.exchange_alltoall(g, max_rows=256)       # rank 0 of 1 owns every hash class
        assert sorted(g.result()) == before
    # an overflowing block is reported before the table is touched
    g = make_table(D, 1000)
    before = sorted(g.result())
    for fn in (comm.exchange_allgather, comm.exchange_alltoall):
        with pytest.raises(T.DbhipError) as e:
            fn(g, max_rows=256)
        assert e.value.code == T.ERR_CAPACITY
        assert sorted(g.result()) == before
    comm.exchange_alltoall(g, max_rows=2048)
    assert sorted(g.result()) == before
    # plain collectives: identities in a world of one
    x = np.arange(1000, dtype=np.uint64) * np.uint64(977)
    send, recv = D.DeviceBuffer.from_numpy(x), D.DeviceBuffer(8000 + 64)
    for op in ("allgather", "alltoall"):
        recv.zero()
        getattr(comm, op)(send.ptr, recv.ptr, 8000)
        assert np.array_equal(recv.to_numpy(np.uint64, 1000), x)
    recv.zero()
    comm.allreduce_sum_u64(send.ptr, recv.ptr, 1000)
    assert np.array_equal(recv.to_numpy(np.uint64, 1000), x)
    comm.destroy()


# ---- world > 1 on one GPU: the in-process loopback world (dbhip_comm_create_loopback), one host thread per rank ---------------------
def run_ranks(world, fn):
    """fn(rank) on `world` threads; returns the results by rank, re-raises the first failure"""
    import threading
    out, err = [None] * world, [None] * world

    def body(r):
        try:
            out[r] = fn(r)
        except BaseException as e:  # noqa: BLE001
            err[r] = e
    ts = [threading.Thread(target=body, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(300)
    for e in err:
        if e is not None:
            raise e
    return out


@pytest.mark.parametrize("world", [2, 3, 8])
def test_block_exchange_between_ranks_routes_every_row_once(gpu, world):
    """dbhip_exchange_begin / _finish (the shuffle of the hash join and the exchange of the distributed sort behind the ABI): every
    rank scatters its shard by siphash64(key) % world (dbhip_scatter_indices, the reference's HashFlightScatter) and ONE grouped
    all-to-all moves all columns — values, validities, a Boolean column, a Decimal128 column, inline strings. Rank r must end with
    exactly the rows whose destination is r, grouped by source rank, in their order inside a source."""
    D = gpu
    rng = np.random.default_rng(100 + world)
    sizes = [int(x) for x in rng.integers(0, 40_000, world)]
    sizes[0] = 0 if world > 2 else sizes[0]                      # an empty shard
    shards = []
    for r in range(world):
        n = sizes[r]
        key = rng.integers(0, 5000, n).astype(np.int64)
        kv = rng.random(n) > 0.1
        pay = rng.integers(-2**40, 2**40, n).astype(np.int64)
        flag = rng.random(n) > 0.5
        dec = [int(x) * 10**20 + 3 for x in rng.integers(-10**6, 10**6, n)]
        tag = [b"r%d-%d" % (r, i % 997) for i in range(n)]
        shards.append((key, kv, pay, flag, dec, tag))
    gid = 7000 + world

    def rank_fn(r):
        key, kv, pay, flag, dec, tag = shards[r]
        n = len(key)
        comm = D.Comm.loopback(gid, r, world)
        cols = [D.Column.from_numpy(key, validity=kv), D.Column.from_numpy(pay), D.Column.boolean(flag, validity=kv),
                D.Column.decimal128(dec, 38, 2), D.Column.from_views(D.make_views(tag))]
        if n:
            dest, counts = D.scatter_indices([cols[0]], world, default_index=world - 1)
        else:
            dest, counts = D.DeviceBuffer(16), np.zeros(world, np.uint64)
        got, starts = comm.exchange_block(cols, dest)
        d = dest.to_numpy(np.uint32, n)
        res = (d, [c.to_numpy() for c in got], [c.validity_numpy() for c in got], starts)
        comm.destroy()
        return res
    outs = run_ranks(world, rank_fn)
    for r in range(world):
        _, cols, valids, starts = outs[r]
        exp_rows = [(s, i) for s in range(world) for i in np.nonzero(outs[s][0] == r)[0].tolist()]
        assert starts == [sum(int((outs[s][0] == r).sum()) for s in range(q)) for q in range(world + 1)]
        assert len(exp_rows) == starts[-1]
        ek = np.array([shards[s][0][i] for s, i in exp_rows], np.int64)
        ekv = np.array([shards[s][1][i] for s, i in exp_rows], bool)
        assert np.array_equal(cols[0], ek) and np.array_equal(valids[0], ekv)
        assert np.array_equal(cols[1], np.array([shards[s][2][i] for s, i in exp_rows], np.int64))
        assert np.array_equal(cols[2], np.array([shards[s][3][i] for s, i in exp_rows], bool)) and np.array_equal(valids[2], ekv)
        assert [int(x) for x in cols[3]] == [shards[s][4][i] for s, i in exp_rows]
        assert D.view_strings(cols[4]) == [shards[s][5][i] for s, i in exp_rows]
    assert sum(o[3][-1] for o in outs) == sum(sizes)


@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_block_exchange_moves_strings_with_data_buffers(gpu, world):
    """Round 5: String columns whose values live in data buffers (longer than 12 bytes) travel in the exchange — every destination's long
    values packed back to back, the views re-based, the byte counts exchanged with the row counts, all pieces in the one grouped
    all-to-all (the reference ships whole blocks, flight_scatter_hash.rs:57-120 + exchange/serde). Two String columns (one nullable, with
    empty / inline / long values mixed; one all long), an Int64 column beside them; rank r must end with exactly its rows, strings intact."""
    D = gpu
    rng = np.random.default_rng(300 + world)
    sizes = [int(x) for x in rng.integers(1, 30_000, world)]
    if world > 2:
        sizes[1] = 0
    shards = []
    for r in range(world):
        n = sizes[r]
        key = rng.integers(0, 7000, n).astype(np.int64)
        mixed = [(b"" if i % 11 == 0 else b"s%d" % i if i % 3 == 0 else b"rank-%d-row-%07d-" % (r, i) + b"x" * int(i % 50)) for i in range(n)]
        mv = rng.random(n) > 0.15
        long_ = [b"a-long-comment-of-rank-%d-for-row-%09d" % (r, i) for i in range(n)]
        shards.append((key, mixed, mv, long_))
    gid = 9100 + world

    def rank_fn(r):
        key, mixed, mv, long_ = shards[r]
        n = len(key)
        comm = D.Comm.loopback(gid, r, world) if world > 1 else D.Comm.local()
        cols = [D.Column.from_numpy(key), D.Column.strings(mixed, validity=mv), D.Column.strings(long_)]
        if n:
            dest, _ = D.scatter_indices([cols[0]], world)
        else:
            dest = D.DeviceBuffer(16)
        got, starts = comm.exchange_block(cols, dest)
        d = dest.to_numpy(np.uint32, n)
        m = got[0].n
        res = (d, got[0].to_numpy(), got[1].to_strings(), got[1].validity_numpy(), got[2].to_strings(), starts, m)
        comm.destroy()
        return res
    outs = run_ranks(world, rank_fn) if world > 1 else [rank_fn(0)]
    for r in range(world):
        _, k, s1, v1, s2, starts, m = outs[r]
        exp_rows = [(s, i) for s in range(world) for i in np.nonzero(outs[s][0] == r)[0].tolist()]
        assert len(exp_rows) == starts[-1] == m
        assert np.array_equal(k, np.array([shards[s][0][i] for s, i in exp_rows], np.int64))
        ev = np.array([shards[s][2][i] for s, i in exp_rows], bool)
        assert np.array_equal(v1, ev)
        assert [x for x, ok in zip(s1, ev) if ok] == [shards[s][1][i] for s, i in exp_rows if shards[s][2][i]]
        assert s2 == [shards[s][3][i] for s, i in exp_rows]
    assert sum(o[6] for o in outs) == sum(sizes)
    # plain finish refuses a block whose strings need a buffer, before anything is sent
    if world == 1:
        import ctypes as C
        comm = D.Comm.local()
        cols = [D.Column.strings([b"a value that is longer than twelve bytes"] * 5)]
        x, rows = C.c_void_p(), C.c_int64()
        dest = D.DeviceBuffer.from_numpy(np.zeros(5, np.uint32))
        T.check(T.lib().dbhip_exchange_begin(comm.h, D._cols(cols), 1, C.c_void_p(dest.ptr), C.c_int64(5), C.byref(rows), C.byref(x), None))
        out = D.DeviceBuffer(5 * 16 + 64)
        dp, vp, st = (C.c_void_p * 1)(out.ptr), (C.c_void_p * 1)(None), (C.c_int64 * 2)()
        assert T.lib().dbhip_exchange_finish(x, dp, vp, st, None) == T.ERR_INVALID
        T.lib().dbhip_exchange_destroy(x)
        comm.destroy()


@pytest.mark.parametrize("world", [2, 3])
def test_shuffle_and_sort_exchange_plans_behind_the_abi(gpu, world):
    """dbhip_shuffle_exchange_begin / dbhip_sort_exchange_begin: the two distributed plans as single calls, no torch and no Python
    plan logic. Shuffle: every received row hashes to the receiving rank (the reference's siphash64 % world) and the union of what the
    ranks hold is the input. Sort: rank r receives exactly the rows of range partition r (rows <= bound[r] after bound[r - 1], in the
    keys' order incl. NULLs-last), so sorting each rank locally and reading the ranks in order is the globally sorted column."""
    D = gpu
    rng = np.random.default_rng(300 + world)
    gid = 7400 + world
    sizes = [int(x) for x in rng.integers(1000, 30_000, world)]
    keys = [rng.integers(-10**6, 10**6, n).astype(np.int64) for n in sizes]
    kvalid = [rng.random(n) > 0.05 for n in sizes]
    pays = [rng.integers(0, 2**50, n).astype(np.int64) for n in sizes]
    allk = np.concatenate([np.where(v, k, 2**62) for k, v in zip(keys, kvalid)])        # (NULLs sort last: stand-in value for the bounds)
    qs = np.sort(allk)[[len(allk) * (i + 1) // world for i in range(world - 1)]]
    bounds = np.array(qs, np.int64)

    def rank_fn(r):
        comm = D.Comm.loopback(gid, r, world)
        ck, cp = D.Column.from_numpy(keys[r], validity=kvalid[r]), D.Column.from_numpy(pays[r])
        got, starts = comm.shuffle_exchange_block([ck], [ck, cp])
        rk, rkv, rp = got[0].to_numpy(), got[0].validity_numpy(), got[1].to_numpy()
        dest, _ = D.scatter_indices([got[0]], world, default_index=0) if got[0].n else (None, None)
        d = dest.to_numpy(np.uint32, got[0].n) if dest is not None else np.zeros(0, np.uint32)
        got2, _ = comm.sort_exchange_block([ck], [D.Column.from_numpy(bounds)], [ck, cp])
        sk, skv, sp = got2[0].to_numpy(), got2[0].validity_numpy(), got2[1].to_numpy()
        comm.destroy()
        return (rk, rkv, rp, d, starts), (sk, skv, sp)
    outs = run_ranks(world, rank_fn)
    # shuffle: routed by the reference's hash, nothing lost, payload travels with its key
    for r in range(world):
        rk, rkv, rp, d, starts = outs[r][0]
        assert (d == r).all() and starts[-1] == len(rk)
    as_set = lambda ks, vs, ps: sorted((bool(v), int(k) if v else 0, int(p)) for k, v, p in zip(ks, vs, ps))
    assert as_set(np.concatenate([o[0][0] for o in outs]), np.concatenate([o[0][1] for o in outs]), np.concatenate([o[0][2] for o in outs])) == \
        as_set(np.concatenate(keys), np.concatenate(kvalid), np.concatenate(pays))
    # sort: rank r holds range r
    merged = []
    for r in range(world):
        sk, skv, sp = outs[r][1]
        img = np.where(skv, sk, 2**62)
        lo = bounds[r - 1] if r > 0 else -2**63
        hi = bounds[r] if r < world - 1 else 2**63 - 1
        assert ((img > lo) | (r == 0)).all() and (img <= hi).all()
        merged.append(np.sort(img))
    assert np.array_equal(np.concatenate(merged), np.sort(allk))


@pytest.mark.parametrize("world", [2, 5])
def test_shard_topk_allgather_and_partial_state_exchange_between_ranks(gpu, world):
    """dbhip_vec_topk_allgather: every rank's local top-k (local row numbers) -> the global top-k on every rank == the k smallest of
    all shards' candidates with ids made global; and dbhip_groupby_exchange_alltoall between `world` ranks: rank r ends up owning the
    groups with hash % world == r, the union is the single-table result."""
    D = gpu
    rng = np.random.default_rng(world)
    nq, k = 37, 10
    dists = [np.sort(rng.random((nq, k)).astype(np.float32), axis=1) for _ in range(world)]
    ids = [np.stack([rng.permutation(10_000)[:k] for _ in range(nq)]).astype(np.uint32) for _ in range(world)]
    ids[world - 1][:, k - 3:] = 0xFFFFFFFF                        # a shard with fewer than k answers
    dists[world - 1][:, k - 3:] = np.inf
    offs = [r * 1_000_000 for r in range(world)]
    n_rows = 60_000
    keys = [rng.integers(0, 300, n_rows).astype(np.int64) for _ in range(world)]
    vals = [rng.integers(-1000, 1000, n_rows).astype(np.int64) for _ in range(world)]
    gid = 9000 + world

    def rank_fn(r):
        comm = D.Comm.loopback(gid, r, world)
        gi, gd = comm.topk_allgather(D.DeviceBuffer.from_numpy(ids[r]), D.DeviceBuffer.from_numpy(dists[r]), nq, k, offs[r])
        g = D.GroupBy([T.T_I64], [(T.AGG_SUM, T.T_I64, 0, 0, 0), (T.AGG_COUNT, 0, 0, 0, 0)])
        g.add_block([D.Column.from_numpy(keys[r])], [D.Column.from_numpy(vals[r]), None], n_rows)
        comm.exchange_alltoall(g, max_rows=512)
        rows = sorted(g.result())
        g.destroy()
        comm.destroy()
        return gi, gd, rows
    outs = run_ranks(world, rank_fn)
    allc = [[(float(dists[r][q, j]), int(ids[r][q, j]) + offs[r]) for r in range(world) for j in range(k) if ids[r][q, j] != 0xFFFFFFFF] for q in range(nq)]
    for gi, gd, _ in outs:
        for q in range(nq):
            exp = sorted(allc[q])[:k]
            assert [int(x) for x in gi[q]] == [e[1] for e in exp] and np.array_equal(gd[q], np.array([e[0] for e in exp], np.float32))
    merged = {}
    for _, _, rows in outs:
        for key, sm, cnt in rows:
            assert key not in merged
            merged[key] = (sm, cnt)
    allk, allv = np.concatenate(keys), np.concatenate(vals)
    assert merged == {int(u): (int(allv[allk == u].sum()), int((allk == u).sum())) for u in np.unique(allk)}


def test_a_rank_that_gives_up_wakes_the_others(gpu):
    """(a) dbhip_comm_abort while the other rank waits in an all-gather; (b) an exchange that fails on ONE rank (its String column claims
    data buffers it does not have) while the other rank is already in the rendezvous of the counts: nobody hangs, the waiting rank gets
    DBHIP_ERR_INVALID with the failing rank's message, and the group stays failed for later collectives."""
    import threading
    import time
    D = gpu
    for case in ("abort", "failed exchange"):
        gid = 9900 + (1 if case == "abort" else 2)
        n = 1000
        vals = np.arange(n, dtype=np.int64)
        dest = (np.arange(n) % 2).astype(np.uint32)
        both_joined = threading.Barrier(2)          # (the group must hold both ranks before one of them leaves it again)

        def rank_fn(r, both_joined=both_joined):
            comm = D.Comm.loopback(gid, r, 2)
            both_joined.wait(60)
            try:
                if case == "abort":
                    if r == 1:
                        time.sleep(0.5)
                        comm.abort()
                        return "aborted"
                    buf = D.DeviceBuffer(64)
                    out = D.DeviceBuffer(128)
                    comm.allgather(buf.ptr, out.ptr, 64)
                    return "completed"
                col = D.Column.from_numpy(vals)
                if r == 1:
                    time.sleep(0.5)
                    bad = D.Column.strings([b"x" * 20] * n)
                    c = bad.c()
                    c.buffers = None                                   # n_buffers > 0 without the buffers: refused before any collective
                    x, rows = C.c_void_p(), C.c_int64()
                    T.check(T.lib().dbhip_exchange_begin(comm.h, (type(c) * 1)(c), 1, C.c_void_p(D.DeviceBuffer.from_numpy(dest).ptr), C.c_int64(n),
                                                         C.byref(rows), C.byref(x), None))
                    return "completed"
                comm.exchange_block([col], D.DeviceBuffer.from_numpy(dest))
                return "completed"
            except T.DbhipError as e:
                try:                                                    # the group is dead for good
                    buf = D.DeviceBuffer(64)
                    keep = D.DeviceBuffer(128)
                    comm.allgather(buf.ptr, keep.ptr, 64)
                    later = "later collective completed"
                except T.DbhipError as e2:
                    later = str(e2)
                return (e.code, str(e), later)
            finally:
                comm.destroy()
        t0 = time.time()
        outs = run_ranks(2, rank_fn)
        assert time.time() - t0 < 60
        code, msg, later = outs[0]
        assert code == T.ERR_INVALID and "aborted" in msg and "rank 1" in msg and "aborted" in later, (case, outs)
        if case == "failed exchange":
            assert outs[1][0] == T.ERR_INVALID and "needs its buffers" in outs[1][1] and "needs its buffers" in msg


def test_topk_allgather_refuses_ids_past_the_u32_space(gpu):
    D = gpu
    comm = D.Comm.local()
    nq, k = 4, 3
    ids = np.arange(nq * k, dtype=np.uint32).reshape(nq, k)
    ids[0, 2] = 0xFFFFFFFF                                           # "no neighbour" stays what it is
    dist = np.sort(np.random.default_rng(0).random((nq, k)).astype(np.float32), axis=1)
    gi, _ = comm.topk_allgather(D.DeviceBuffer.from_numpy(ids), D.DeviceBuffer.from_numpy(dist), nq, k, 2**32 - 20)
    assert int(gi[1][0]) == 3 + 2**32 - 20
    with pytest.raises(T.DbhipError, match="does not fit the u32 id space"):
        comm.topk_allgather(D.DeviceBuffer.from_numpy(ids), D.DeviceBuffer.from_numpy(dist), nq, k, 2**32 - 8)
    comm.destroy()

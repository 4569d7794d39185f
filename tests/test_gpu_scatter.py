"""GPU: siphash64 and the hash-shuffle scatter indices through the C-ABI (k_scatter.hip) — bit-exact against the reference's golden
values, the oracle, and the shuffle hash join's device operators."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from databend_amd import _lib as T
from databend_amd.device import make_views_general
from tests import oracle_lib as O
from tests.test_siphash_cpu import golden_column, host_values, orc_hash, seeded_columns

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def device_column(gpu, kind, col, extra, n, validity=None):
    if kind == "string":
        return gpu.Column.strings(extra, validity=validity)
    if kind == "bool":
        return gpu.Column.boolean(np.array(host_values(kind, col, None, n), dtype=bool), validity=validity)
    if kind == "decimal128":
        return gpu.Column.decimal128(extra, col.precision, col.scale, validity=validity)
    if kind == "decimal256":
        return gpu.Column.decimal256(extra, col.precision, col.scale, validity=validity)
    return gpu.Column.from_numpy(col.arr[:n], col.dtype, validity=validity, precision=col.precision, scale=col.scale)


def test_golden_values_through_the_c_abi(gpu):
    cases = json.load(open(os.path.join(HERE, "golden", "siphash.json"), encoding="utf-8"))
    checked = 0
    for c in cases:
        t = c["type"]
        if t == "string":
            col = gpu.Column.strings([c["value"].encode("utf-8")])
        elif t == "bytes":
            col = gpu.Column.strings([bytes.fromhex(c["bytes"])])
        elif t == "bool":
            col = gpu.Column.boolean(np.array([c["value"]], dtype=bool))
        elif t == "decimal64":
            col = gpu.Column.from_numpy(np.array([c["value"]], np.int64), T.T_DEC64, precision=c["precision"], scale=c["scale"])
        else:
            code, dt = {"timestamp": (T.T_TIMESTAMP, np.int64), "u32": (T.T_U32, np.uint32), "date": (T.T_DATE, np.int32)}[t]
            col = gpu.Column.from_numpy(np.array([c["value"]], dt), code)
        assert int(gpu.siphash64(col)[0]) == c["expected"], c["what"]
        checked += 1
    assert checked >= 16


@pytest.mark.parametrize("n", [1, 1000, 200_000])
def test_siphash64_matches_the_oracle_on_every_type(gpu, n):
    rng = np.random.default_rng(n)
    valid = rng.integers(0, 7, n) > 0
    for kind, col, extra in seeded_columns(n, n + 3):
        for v in (None, valid):
            got = gpu.siphash64(device_column(gpu, kind, col, extra, n, v))
            h = O.HostCol(col.dtype, col.arr, v, col.precision, col.scale, buffers=col.buffers)
            assert np.array_equal(got, orc_hash(h, n)), (kind, v is not None)


def test_unsupported_decimal_precision_is_refused(gpu):
    col = gpu.Column.decimal256([1, 2, 3], 50, 2)
    with pytest.raises(T.DbhipError) as e:
        gpu.siphash64(col)
    assert e.value.code == T.ERR_UNSUPPORTED
    # long views without their data buffers are refused instead of dereferenced
    views, _buf = make_views_general([b"short", b"a value of more than twelve bytes"])
    with pytest.raises(T.DbhipError) as e2:
        gpu.siphash64(gpu.Column.from_views(views))
    assert e2.value.code == T.ERR_INVALID
    assert gpu.siphash64(gpu.Column.from_views(views[:1])).shape == (1,)


@pytest.mark.parametrize("n,m", [(0, 4), (1, 1), (50_000, 8), (300_000, 3), (100_000, 5000)])
def test_scatter_indices_match_the_oracle(gpu, oracle, n, m):
    """one key with a default scatter index for NULL keys, two and three keys combined through the DefaultHasher; LDS histogram
    (<= 4096 destinations) and global counters"""
    rng = np.random.default_rng(n + m)
    k1 = rng.integers(0, 100_000, max(n, 1)).astype(np.int64)[:n]
    v1 = (rng.integers(0, 6, max(n, 1)) > 0)[:n]
    k2 = rng.integers(-5, 5, max(n, 1)).astype(np.int32)[:n]
    strs = [b"Customer#%09d" % x for x in rng.integers(0, 3000, n)]
    views, buf = make_views_general(strs) if n else (np.zeros((0, 16), np.uint8), np.zeros(16, np.uint8))
    combos = [([("i64", k1, v1)], min(3, m - 1)), ([("i64", k1, None)], 0), ([("i64", k1, v1), ("str", strs, None)], 0),
              ([("i32", k2, None), ("i64", k1, v1), ("str", strs, v1)], 0)]
    for keys, default in combos:
        gcols, hcols = [], []
        for kind, data, v in keys:
            if kind == "str":
                gcols.append(gpu.Column.strings(data, validity=v) if n else gpu.Column.from_views(views))
                hcols.append(O.HostCol(T.T_STRING, views, v, buffers=[buf]))
            else:
                gcols.append(gpu.Column.from_numpy(data, validity=v))
                hcols.append(O.HostCol(T.T_I64 if kind == "i64" else T.T_I32, data, v))
        idx, counts = gpu.scatter_indices(gcols, m, default)
        eidx, ecnt = np.zeros(max(n, 1), np.uint32), np.zeros(m, np.uint64)
        assert oracle.orc_scatter_indices(O.cols(hcols), len(hcols), C.c_int64(n), C.c_uint64(m), C.c_uint64(default), eidx.ctypes.data_as(C.c_void_p),
                                          ecnt.ctypes.data_as(C.c_void_p)) == 0
        assert np.array_equal(idx.to_numpy(np.uint32, n), eidx[:n]) and np.array_equal(counts, ecnt), (len(keys), default)
        if n >= 50_000 and m <= 8:
            assert counts.min() > 0.5 * n / m      # siphash spreads the keys


def test_shuffle_hash_join_operators_and_plan_on_one_rank(gpu):
    """ShuffleDeviceOps driven like three ranks would drive them (scatter both sides, destination d of every shard, join), and the
    whole plan over nccl in a world of one; the union of the pairs is the single-node join."""
    import socket

    import torch
    import torch.distributed as dist

    from databend_amd import dist as DX
    from databend_amd.sort_ops import ShuffleDeviceOps
    rng = np.random.default_rng(41)
    nb, npr, world = 30_000, 80_000, 3
    bk, pk = rng.integers(0, 20_000, nb).astype(np.int64), rng.integers(0, 40_000, npr).astype(np.int64)
    bv, pv = (rng.random(nb) > 0.1).astype(np.uint8), (rng.random(npr) > 0.1).astype(np.uint8)
    by_key = {}
    for r in range(nb):
        if bv[r]:
            by_key.setdefault(int(bk[r]), []).append(r)
    exp = sorted((i, r) for i in range(npr) if pv[i] for r in by_key.get(int(pk[i]), []))
    ops = ShuffleDeviceOps(torch)
    cuts_b, cuts_p = [0, 5000, 5000, nb], [0, 30_000, 60_000, npr]
    t = lambda a: torch.from_numpy(a.copy()).cuda()
    bsh = [[t(bk[cuts_b[r]:cuts_b[r + 1]]), t(np.arange(cuts_b[r], cuts_b[r + 1], dtype=np.int64)), t(bv[cuts_b[r]:cuts_b[r + 1]])] for r in range(world)]
    psh = [[t(pk[cuts_p[r]:cuts_p[r + 1]]), t(np.arange(cuts_p[r], cuts_p[r + 1], dtype=np.int64)), t(pv[cuts_p[r]:cuts_p[r + 1]])] for r in range(world)]
    gb = [ops.scatter(s, 0, 2, world) for s in bsh]
    gp = [ops.scatter(s, 0, 2, world) for s in psh]

    def dest(grouped, d):
        cols = []
        for c in range(3):
            pieces = []
            for flat, counts in grouped:
                at = sum(counts[:d])
                pieces.append(flat[c][at:at + counts[d]])
            cols.append(torch.cat(pieces))
        return cols
    pairs = []
    for d in range(world):
        out_p, out_b = ops.join(dest(gb, d), 0, 2, dest(gp, d), 0, 2)
        assert torch.equal(out_p[0], out_b[0])
        pairs += list(zip(out_p[1].cpu().tolist(), out_b[1].cpu().tolist()))
    assert sorted(pairs) == exp and len(exp) > 10_000
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
    try:
        out_p, out_b = DX.shuffle_hash_join([t(bk), t(np.arange(nb, dtype=np.int64))], 0, [t(pk), t(np.arange(npr, dtype=np.int64))], 0, ops, dist, torch,
                                            build_valids=[t(bv), None], probe_valids=[t(pv), None])
        assert sorted(zip(out_p[1].cpu().tolist(), out_b[1].cpu().tolist())) == exp
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n,m", [(1, 1), (4095, 2), (4097, 8), (300_000, 256), (200_000, 257), (100_000, 5000)])
def test_scatter_block_groups_rows_stably(gpu, n, m):
    """dbhip_scatter_block = DataBlock::scatter (kernels/scatter.rs:20-66): every column grouped by destination, row order kept
    inside a destination — against numpy's stable argsort, for 1 / 2 / 4 / 8 / 16-byte elements, the one-pass path (<= 256
    destinations) and the permutation path (more destinations, 16-byte elements)."""
    rng = np.random.default_rng(n + m)
    index = rng.integers(0, m, n).astype(np.uint32)
    if n > 5000:
        index[:3000] = 0                      # long runs of one destination
    cols = [rng.integers(0, 255, n).astype(np.uint8), rng.integers(-3000, 3000, n).astype(np.int16), rng.standard_normal(n).astype(np.float32),
            rng.integers(-2**62, 2**62, n).astype(np.int64)]
    order = np.argsort(index, kind="stable")
    ibuf = gpu.DeviceBuffer.from_numpy(index)
    out = gpu.scatter_block([gpu.Column.from_numpy(c) for c in cols], ibuf, m)
    for got, src in zip(out, cols):
        assert np.array_equal(got.to_numpy(), src[order])
    dec = [int(x) for x in rng.integers(-2**62, 2**62, n)]
    out = gpu.scatter_block([gpu.Column.decimal128(dec, 38, 0), gpu.Column.from_numpy(cols[3])], ibuf, m)     # 16-byte elements
    assert [int(x) for x in out[0].to_numpy()] == [dec[i] for i in order]
    assert np.array_equal(out[1].to_numpy(), cols[3][order])


# ---- DataBlock::scatter over whole columns and DataBlock::concat (dbhip_scatter_columns / dbhip_concat_columns) --------------------
def _golden_kernel(kind):
    return [c for c in json.load(open(os.path.join(HERE, "golden", "kernel.json"), encoding="utf-8"))["cases"] if c["kind"] == kind]


def _device_cols(gpu, cols):
    out = []
    for kind, vals, valid in cols:
        out.append(gpu.Column.strings(vals, validity=valid) if kind == "str" else gpu.Column.from_numpy(vals, validity=valid))
    return out


def _rendered(col, kind):
    from tests import scatter_cases as SC
    vals = col.string_values() if kind == "str" else col.to_numpy().tolist()
    return SC.render(kind, vals, col.validity_numpy())


def test_kernel_pass_scatter_and_concat_goldens_through_the_c_abi(gpu):
    """kernel-pass.txt 'Scatter' (:211+) and 'Concat' (:21-53): nullable Int32 / UInt8 and nullable String columns through
    dbhip_scatter_columns / dbhip_concat_columns, rendered like the reference renders them."""
    from tests import scatter_cases as SC
    for case in _golden_kernel("scatter"):
        cols = SC.cells_to_columns(case["header"], case["source"])
        S = len(case["results"])
        blocks, starts = gpu.scatter_columns(_device_cols(gpu, cols), gpu.DeviceBuffer.from_numpy(np.array(case["arg"], np.uint32)), S)
        assert starts == [0, 2, 4, 5]
        for d in range(S):
            for c, (kind, _, _) in enumerate(cols):
                assert _rendered(blocks[d][c], kind) == [r[c] for r in case["results"][d]], (d, c)
    checked = 0
    for case in _golden_kernel("concat"):
        for c in range(len(case["header"])):
            blks = [b[c] for b in case["blocks"]]
            if "values" not in blks[0]:
                continue                      # Null / Array(Nothing): outside the path's types
            dcols = []
            for b in blks:
                valid = np.array(b["validity"], bool) if "validity" in b else None
                if isinstance(b["values"][0], str):
                    dcols.append(gpu.Column.strings([x.encode() for x in b["values"]], validity=valid))
                    kind = "str"
                else:
                    dcols.append(gpu.Column.from_numpy(np.array(b["values"], np.int32), validity=valid))
                    kind = "int"
            assert _rendered(gpu.concat_columns(dcols), kind) == [r[c] for r in case["result"]], c
            checked += 1
    assert checked == 3


def _random_block(gpu, rng, n):
    """one column of every type the library accepts, most of them nullable: (device column, host values, host validity, kind)"""
    cols = []
    valid = lambda: rng.random(n) > 0.25

    def add(col, vals, v, kind):
        cols.append((col, vals, v, kind))
    for dt in (np.int8, np.uint8, np.int16, np.uint16, np.int32, np.uint32, np.int64, np.uint64):
        a = rng.integers(0, 100, n).astype(dt)
        v = valid() if dt in (np.int32, np.int64, np.uint8) else None
        add(gpu.Column.from_numpy(a, validity=v), a, v, "num")
    for dt in (np.float32, np.float64):
        a = rng.standard_normal(n).astype(dt)
        v = valid()
        add(gpu.Column.from_numpy(a, validity=v), a, v, "num")
    a = rng.integers(-30000, 30000, n).astype(np.int32)
    add(gpu.Column.from_numpy(a, T.T_DATE), a, None, "num")
    a = rng.integers(-2**50, 2**50, n).astype(np.int64)
    v = valid()
    add(gpu.Column.from_numpy(a, T.T_TIMESTAMP, validity=v), a, v, "num")
    add(gpu.Column.from_numpy(a, T.T_DEC64, validity=v, precision=18, scale=2), a, v, "num")
    d128 = [int(x) * (1 << 40) + 7 for x in rng.integers(-2**62, 2**62, n)]
    v = valid()
    add(gpu.Column.decimal128(d128, 38, 3, validity=v), d128, v, "list")
    d256 = [int(x) * (1 << 150) - 11 for x in rng.integers(-2**62, 2**62, n)]
    add(gpu.Column.decimal256(d256, 70, 3, validity=v), d256, v, "list")
    b = rng.random(n) > 0.5
    v = valid()
    add(gpu.Column.boolean(b, validity=v), b, v, "num")
    add(gpu.Column.boolean(b), b, None, "num")
    strs = [(b"k%d" % x) if x % 3 else (b"a long string value number %09d, longer than twelve bytes" % x) for x in rng.integers(0, 10**6, n)]
    v = valid()
    add(gpu.Column.strings(strs, validity=v), strs, v, "str")
    return cols


def _values(col, kind):
    if kind == "str":
        return col.string_values()
    v = col.to_numpy()
    return [int(x) for x in v] if kind == "list" else v


@pytest.mark.parametrize("n,S", [(1, 2), (63, 3), (64, 2), (299, 24), (5000, 7), (70_000, 256), (70_000, 300), (200_000, 8)])
def test_scatter_then_concat_equals_take_over_every_column_type(gpu, oracle, n, S):
    """tests/it/kernel.rs:519-566 (test_scatter) restated: scatter a block of every column type the library accepts (nullable and
    not; Boolean, String with long values, Decimal128 / Decimal256) by random indices; every destination equals the oracle's
    divide_indices_by_scatter_size + take, and DataBlock::concat of the pieces equals the block taken by the grouped indices."""
    from tests import scatter_cases as SC
    rng = np.random.default_rng(n * 31 + S)
    index = rng.integers(0, S, n).astype(np.uint32)
    if n > 1000:
        index[: n // 3] = S - 1                 # a long run of one destination; destination 0 may stay small
        index[index == 1] = 0                   # an empty destination
    cols = _random_block(gpu, rng, n)
    blocks, starts = gpu.scatter_columns([c[0] for c in cols], gpu.DeviceBuffer.from_numpy(index), S)
    _, estarts, rows = SC.oracle_scatter(oracle, index, S, [])
    assert starts == estarts
    take_indices = [j for d in range(S) for j in np.nonzero(index == d)[0].tolist()] if n <= 5000 else rows.tolist()
    assert rows.tolist() == take_indices
    for c, (col, vals, valid, kind) in enumerate(cols):
        exp_all = [vals[i] for i in take_indices] if kind != "num" else np.asarray(vals)[take_indices]
        exp_valid = valid[take_indices] if valid is not None else np.ones(n, bool)
        # every destination on its own (stand-alone Bitmaps, value slices)
        for d in (range(S) if n <= 5000 else (0, 1, S - 1)):
            lo, hi = starts[d], starts[d + 1]
            piece = blocks[d][c]
            assert piece.n == hi - lo
            got = _values(piece, kind)
            assert (list(got) == list(exp_all[lo:hi])) if kind != "num" else np.array_equal(got, exp_all[lo:hi]), (c, d)
            assert np.array_equal(piece.validity_numpy(), exp_valid[lo:hi]), (c, d)
        # concat of the pieces == take
        cat = gpu.concat_columns([blocks[d][c] for d in range(S)])
        got = _values(cat, kind)
        assert (list(got) == list(exp_all)) if kind != "num" else np.array_equal(got, exp_all), c
        assert np.array_equal(cat.validity_numpy(), exp_valid), c


def test_concat_of_sliced_blocks_and_mixed_validity(gpu, oracle):
    """DataBlock::concat of SLICES (Bitmaps read from a bit offset, value buffers from an element offset), of blocks where only
    some carry a validity (the others count as all valid), of a constant entry, and of > 64 blocks (several launches over one Bitmap)."""
    from tests import scatter_cases as SC
    rng = np.random.default_rng(77)
    n = 1000
    a = rng.integers(-50, 50, n).astype(np.int64)
    v = rng.random(n) > 0.4
    b = rng.random(n) > 0.5
    strs = [b"s%05d-%s" % (i, b"x" * (i % 20)) for i in range(n)]
    ca, cb, cs = gpu.Column.from_numpy(a, validity=v), gpu.Column.boolean(b, validity=v), gpu.Column.strings(strs, validity=v)
    cuts = sorted(set(rng.integers(0, n, 90).tolist() + [0, n, 3, 64, 65, 127, 128]))
    spans = list(zip(cuts[:-1], cuts[1:]))
    order = rng.permutation(len(spans)).tolist()
    rows = [i for k in order for i in range(*spans[k])]
    got = gpu.concat_columns([ca.slice(*spans[k]) for k in order])
    assert np.array_equal(got.to_numpy(), a[rows]) and np.array_equal(got.validity_numpy(), v[rows])
    got = gpu.concat_columns([cb.slice(*spans[k]) for k in order])
    assert np.array_equal(got.to_numpy(), b[rows]) and np.array_equal(got.validity_numpy(), v[rows])
    got = gpu.concat_columns([cs.slice(*spans[k]) for k in order])
    assert got.string_values() == [strs[i] for i in rows] and np.array_equal(got.validity_numpy(), v[rows])
    # the oracle's bit-by-bit concat of the same validity pieces
    exp = SC.oracle_concat_bits(oracle, [v[spans[k][0]:spans[k][1]] for k in order], [spans[k][1] - spans[k][0] for k in order])
    assert np.array_equal(exp, v[rows])
    # mixed validity + a constant entry
    plain = gpu.Column.from_numpy(a[:100])
    const = gpu.Column.scalar(7, T.T_I64)
    const.n = 33
    got = gpu.concat_columns([ca.slice(10, 75), plain, const, ca.slice(900, 1000)])
    assert np.array_equal(got.to_numpy(), np.concatenate([a[10:75], a[:100], np.full(33, 7), a[900:]]))
    assert np.array_equal(got.validity_numpy(), np.concatenate([v[10:75], np.ones(133, bool), v[900:]]))
    # two string columns with their own buffers: the views are rebased onto the concatenated buffer table
    other = [b"another column's long value %04d ........" % i for i in range(50)]
    got = gpu.concat_columns([cs.slice(5, 25), gpu.Column.strings(other), cs.slice(500, 510)])
    assert got.n_buffers == 3 and got.string_values() == strs[5:25] + other + strs[500:510]
    with pytest.raises(T.DbhipError):
        gpu.concat_columns([ca, cb])          # DataBlock::concat checks the schema
    with pytest.raises(T.DbhipError):
        gpu.scatter_columns([ca], gpu.DeviceBuffer.from_numpy(np.full(n, 9, np.uint32)), 4)      # an index outside the destinations


def test_take_of_an_unaligned_column_view_does_not_read_past_the_buffer(gpu):
    """the windowed take assumed 16-byte aligned, padded columns. A view that starts at an odd element of a buffer whose
    last element is the last one of the allocation now reads only 16-byte granules that hold wanted elements."""
    n = 200_000
    rng = np.random.default_rng(3)
    base = rng.integers(-2**40, 2**40, n + 1).astype(np.int64)
    buf = gpu.DeviceBuffer.from_numpy(base)
    view = gpu.Column(T.T_I64, n, gpu.BorrowedBuffer(buf.ptr + 8, n * 8, keep=buf))          # 8 mod 16: not vector aligned
    sel = np.sort(rng.choice(n, n // 3, replace=False)).astype(np.uint32)
    sel[-1] = n - 1                                                                            # the very last element is selected
    got = gpu.take(view, gpu.DeviceBuffer.from_numpy(sel), len(sel))
    assert np.array_equal(got.to_numpy(), base[1:][sel])
    v32 = gpu.Column(T.T_I32, 2 * n, gpu.BorrowedBuffer(buf.ptr + 4, 2 * n * 4, keep=buf))
    sel2 = np.sort(rng.choice(2 * n, n // 2, replace=False)).astype(np.uint32)
    got = gpu.take(v32, gpu.DeviceBuffer.from_numpy(sel2), len(sel2))
    assert np.array_equal(got.to_numpy(), base.view(np.int32)[1:][sel2])
    outs = gpu.take_block([view, v32], gpu.DeviceBuffer.from_numpy(sel), len(sel))
    assert np.array_equal(outs[0].to_numpy(), base[1:][sel]) and np.array_equal(outs[1].to_numpy(), base.view(np.int32)[1:][sel])


def test_stream_scratch_is_released_with_the_stream(gpu):
    """scratch keyed by (thread, stream) was never freed. A stream's scratch now goes with dbhip_stream_destroy /
    dbhip_stream_release_scratch, and a thread keeps at most 8 streams' worth: many short-lived streams do not grow device memory."""
    n = 2_000_000
    keys = gpu.Column.from_numpy(np.random.default_rng(1).integers(0, 1 << 40, n).astype(np.int64))
    L = T.lib()

    def stats():
        out = (C.c_uint64 * 2)()
        T.check(L.dbhip_scratch_stats(out))
        return int(out[0]), int(out[1])
    arr = (T.Col * 1)(keys.c())
    zero = (C.c_uint8 * 1)(0)
    perm = gpu.DeviceBuffer(n * 4)
    e0, b0 = stats()
    for _ in range(24):                       # every iteration: a fresh stream, a sort (~50 MB of scratch), destroy
        s = C.c_void_p()
        T.check(L.dbhip_stream_create(C.byref(s)))
        T.check(L.dbhip_sort_perm(arr, zero, zero, 1, C.c_int64(n), C.c_int64(0), C.c_void_p(perm.ptr), s))
        assert stats()[1] > b0
        T.check(L.dbhip_stream_destroy(s))
        assert stats() == (e0, b0)
    streams = []
    for _ in range(20):                       # foreign streams that are never handed back: the per-thread LRU bounds them
        s = C.c_void_p()
        T.check(L.dbhip_stream_create(C.byref(s)))
        streams.append(s)
        T.check(L.dbhip_sort_perm(arr, zero, zero, 1, C.c_int64(n), C.c_int64(0), C.c_void_p(perm.ptr), s))
        assert stats()[0] <= 8
    assert stats()[1] <= b0 + 8 * (100 << 20)      # (the LRU may also have evicted older entries of this thread: never more than 8 remain)
    for s in streams:
        T.check(L.dbhip_stream_release_scratch(s))
        T.check(L.dbhip_stream_destroy(s))
    assert stats()[0] <= e0 and stats()[1] <= b0
    exp = np.argsort(keys.to_numpy(), kind="stable").astype(np.uint32)
    T.check(L.dbhip_sort_perm(arr, zero, zero, 1, C.c_int64(n), C.c_int64(0), C.c_void_p(perm.ptr), None))
    assert np.array_equal(perm.to_numpy(np.uint32, n), exp)


def test_stream_cancel_stops_multi_launch_operators(gpu):
    """dbhip_stream_cancel (processor.rs:36-41 check_interrupt): a marked stream makes the chunk loops of add_block, the query batches
    of the vector-index search and the sort's passes return DBHIP_ERR_CANCELLED at their next poll; other streams are not affected,
    dbhip_stream_cancel_clear (or destroying the stream) lifts the mark."""
    L = T.lib()
    rng = np.random.default_rng(2)
    n = 1 << 20
    k = gpu.Column.from_numpy(rng.integers(0, 5000, n).astype(np.int64))
    a = gpu.Column.from_numpy(rng.integers(0, 100, n).astype(np.int64))
    s = C.c_void_p()
    T.check(L.dbhip_stream_create(C.byref(s)))
    g = gpu.GroupBy([T.T_I64], [(T.AGG_SUM, T.T_I64, 0, 0, 0), (T.AGG_COUNT, 0, 0, 0, 0)])
    T.check(L.dbhip_stream_cancel(s))
    with pytest.raises(T.DbhipError) as e:
        g.add_block([k], [a, None], n, stream=s)
    assert e.value.code == T.ERR_CANCELLED
    arr = (T.Col * 1)(k.c())
    zero = (C.c_uint8 * 1)(0)
    perm = gpu.DeviceBuffer(n * 4)
    assert L.dbhip_sort_perm(arr, zero, zero, 1, C.c_int64(n), C.c_int64(0), C.c_void_p(perm.ptr), s) == T.ERR_CANCELLED
    # the library stream is another stream: unaffected
    g2 = gpu.GroupBy([T.T_I64], [(T.AGG_COUNT, 0, 0, 0, 0)])
    g2.add_block([k], [None], n)
    assert sum(r[1] for r in g2.result()) == n
    T.check(L.dbhip_stream_cancel_clear(s))
    g.reset(s)
    g.add_block([k], [a, None], n, stream=s)
    T.check(L.dbhip_stream_sync(s))
    assert sum(r[2] for r in g.result()) == n
    T.check(L.dbhip_stream_cancel(s))
    T.check(L.dbhip_stream_destroy(s))          # destroying the stream clears the mark
    g.destroy(); g2.destroy()

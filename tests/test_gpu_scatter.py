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

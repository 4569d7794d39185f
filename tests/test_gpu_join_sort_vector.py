"""GPU: hash join, sort permutation, vector distances / top-k / u8 scoring against the oracle."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from databend_amd import _lib as T
from databend_amd.device import make_views_general
from tests import oracle_lib as O

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("nb,np_,card", [(0, 10, 5), (10, 0, 5), (1000, 5000, 300), (200_000, 500_000, 150_000), (50_000, 50_000, 10)])
def test_inner_join_pairs_match_oracle(gpu, oracle, nb, np_, card):
    rng = np.random.default_rng(nb + np_)
    bk = rng.integers(0, card, nb).astype(np.uint64) * np.uint64(2654435761)
    pk = rng.integers(0, card + card // 2 + 1, np_).astype(np.uint64) * np.uint64(2654435761)
    bvalid = rng.integers(0, 10, nb) > 0
    pvalid = rng.integers(0, 10, np_) > 0
    j = gpu.HashJoin(16)
    # build arrives in chunks (Join::add_block)
    for lo in range(0, nb, 70_000):
        hi = min(nb, lo + 70_000)
        j.add_block(gpu.Column.from_numpy(bk[lo:hi], validity=bvalid[lo:hi]))
    j.final_build()
    gp, gb = j.probe_block(gpu.Column.from_numpy(pk, validity=pvalid))
    cap = len(gp) + 16
    ep, eb = np.zeros(cap, np.uint32), np.zeros(cap, np.uint32)
    bv = np.concatenate([np.packbits(bvalid, bitorder="little"), np.zeros(8, np.uint8)])
    pv = np.concatenate([np.packbits(pvalid, bitorder="little"), np.zeros(8, np.uint8)])
    total = oracle.orc_join_inner_u64(bk.ctypes.data_as(C.c_void_p), bv.ctypes.data_as(C.c_void_p), C.c_int64(nb), pk.ctypes.data_as(C.c_void_p),
                                      pv.ctypes.data_as(C.c_void_p), C.c_int64(np_), ep.ctypes.data_as(C.c_void_p), eb.ctypes.data_as(C.c_void_p), C.c_int64(cap))
    assert total == len(gp)
    assert np.array_equal(gp, ep[:total]) and np.array_equal(gb, eb[:total])
    # every emitted pair joins equal, valid keys
    if total:
        assert (bk[gb] == pk[gp]).all() and bvalid[gb].all() and pvalid[gp].all()


@pytest.mark.parametrize("shape", ["dense", "dense_offset", "wide"])
def test_inner_join_large_tables_behind_the_occupancy_filter(gpu, oracle, shape):
    """Tables of >= 2^19 build rows filter the probe side before a head sector is fetched (one bit per bucket, dbhip_join_finalize; the
    smaller parity cases never build it). Against the oracle's hash join: sparse ids in a bounded span (TPC-H's order keys), the same far
    from zero, and keys all over the u64 range; duplicate build keys, NULLs on both sides, probe keys below the smallest / above the
    largest build key and in the gaps."""
    rng = np.random.default_rng(len(shape))
    nb, np_ = 600_000, 1_500_000
    if shape == "wide":
        universe = rng.integers(0, 2**63, 400_000, dtype=np.uint64)                      # span >> 64 bits per build row
        bk = universe[rng.integers(0, len(universe), nb)]
        pk = np.where(rng.random(np_) < 0.5, universe[rng.integers(0, len(universe), np_)], rng.integers(0, 2**63, np_, dtype=np.uint64))
    else:
        base = np.uint64(0) if shape == "dense" else np.uint64(2**40 + 12345)
        bk = base + np.uint64(1000) + rng.integers(0, 3_000_000, nb).astype(np.uint64) * np.uint64(4)   # sparse ids (TPC-H orderkeys), duplicates
        pk = base + rng.integers(0, 12_008_000, np_).astype(np.uint64)                                    # below, inside (gaps too), above
    bvalid = rng.random(nb) < 0.95
    pvalid = rng.random(np_) < 0.9
    j = gpu.HashJoin(nb)
    for lo in range(0, nb, 250_000):
        hi = min(nb, lo + 250_000)
        j.add_block(gpu.Column.from_numpy(bk[lo:hi], validity=bvalid[lo:hi]))
    j.final_build()
    gp, gb = j.probe_block(gpu.Column.from_numpy(pk, validity=pvalid))
    cap = len(gp) + 16
    ep, eb = np.zeros(cap, np.uint32), np.zeros(cap, np.uint32)
    bv = np.concatenate([np.packbits(bvalid, bitorder="little"), np.zeros(8, np.uint8)])
    pv = np.concatenate([np.packbits(pvalid, bitorder="little"), np.zeros(8, np.uint8)])
    total = oracle.orc_join_inner_u64(bk.ctypes.data_as(C.c_void_p), bv.ctypes.data_as(C.c_void_p), C.c_int64(nb), pk.ctypes.data_as(C.c_void_p),
                                      pv.ctypes.data_as(C.c_void_p), C.c_int64(np_), ep.ctypes.data_as(C.c_void_p), eb.ctypes.data_as(C.c_void_p), C.c_int64(cap))
    assert total == len(gp) and total > 10_000
    assert np.array_equal(gp, ep[:total]) and np.array_equal(gb, eb[:total])
    assert (bk[gb] == pk[gp]).all() and bvalid[gb].all() and pvalid[gp].all()
    # the mark form agrees
    marks = j.probe_mark(gpu.Column.from_numpy(pk, validity=pvalid))
    exp = np.zeros(np_, bool)
    exp[gp] = True
    assert np.array_equal(marks, exp)


@pytest.mark.parametrize("n", [0, 1, 63, 64, 1000, 100_003])
def test_pack_keys_matches_oracle(gpu, oracle, n):
    """dbhip_pack_keys == KeysVec byte layout (method_fixed_keys.rs:310-403) for mixed widths, nullable columns,
    decimals by precision; dbhip_keys_method == choose_hash_method_with_types."""
    rng = np.random.default_rng(n + 3)
    v1, v2 = rng.integers(0, 4, n) > 0, rng.integers(0, 3, n) > 0
    i64 = rng.integers(-2**62, 2**62, n).astype(np.int64)
    i32 = rng.integers(-2**31, 2**31 - 1, n).astype(np.int32)
    u8 = rng.integers(0, 256, n).astype(np.uint8)
    i16 = rng.integers(-2**15, 2**15 - 1, n).astype(np.int16)
    f32 = rng.standard_normal(n).astype(np.float32)
    d128 = [int(x) * 10**7 for x in rng.integers(-10**17, 10**17, n)]
    d128s = [int(x) for x in rng.integers(-10**9, 10**9, n)]
    sets = [
        [(T.T_I64, i64, None, 0, 0)],
        [(T.T_I32, i32, v1, 0, 0), (T.T_U8, u8, None, 0, 0), (T.T_I16, i16, v2, 0, 0)],
        [(T.T_I64, i64, v1, 0, 0), (T.T_DATE, i32, None, 0, 0), (T.T_F32, f32, None, 0, 0)],
        [(T.T_DEC128, d128, v2, 30, 4), (T.T_I16, i16, None, 0, 0)],
        [(T.T_DEC128, d128s, None, 12, 2), (T.T_DEC64, i64, None, 15, 2), (T.T_U8, u8, v1, 0, 0)],
        [(T.T_I64, i64, None, 0, 0), (T.T_TIMESTAMP, i64[::-1].copy(), None, 0, 0), (T.T_DEC128, d128, None, 38, 0)],
    ]
    for spec in sets:
        gcols, hcols = [], []
        for t, arr, v, p_, s_ in spec:
            if t == T.T_DEC128:
                gcols.append(gpu.Column.decimal128(arr, p_, s_, validity=v))
                hcols.append(O.HostCol(t, O.i128_array(arr), v, p_, s_))
            else:
                gcols.append(gpu.Column.from_numpy(arr, t, validity=v, precision=p_, scale=s_))
                hcols.append(O.HostCol(t, arr, v, p_, s_))
        kb = oracle.orc_keys_method(O.cols(hcols), len(hcols))
        assert gpu.keys_method(gcols) == kb and kb > 0
        if n == 0:
            continue
        exp = np.zeros(n * kb, np.uint8)
        assert oracle.orc_pack_keys(O.cols(hcols), len(hcols), C.c_int64(n), kb, exp.ctypes.data_as(C.c_void_p)) == 0
        pk = gpu.pack_keys(gcols)
        assert pk.key_bytes == kb
        assert np.array_equal(pk.to_numpy().reshape(-1), exp)
        allv = np.ones(n, bool)
        for _, _, v, _, _ in spec:
            if v is not None:
                allv &= v
        from databend_amd.device import unpack_bits
        assert np.array_equal(unpack_bits(pk.validity.to_numpy(np.uint8, (n + 7) // 8), n), allv)
    assert gpu.keys_method([gpu.Column.strings([b"a"] * max(n, 1))]) == 0


@pytest.mark.parametrize("nb,np_,card", [(1000, 5000, 300), (120_000, 300_000, 90_000)])
def test_join_on_packed_two_column_keys_u128_and_mark(gpu, oracle, nb, np_, card):
    """Two-column join key (i64, i32 nullable) -> KeysU128 via dbhip_pack_keys; inner pairs == a dictionary join on
    the tuples; rows with a NULL key column never match; probe_mark == 'has at least one pair' (semi/anti join filter)."""
    rng = np.random.default_rng(nb)
    bk1 = rng.integers(0, card, nb).astype(np.int64) * 1_000_003
    bk2 = rng.integers(0, 3, nb).astype(np.int32)
    pk1 = rng.integers(0, card * 2, np_).astype(np.int64) * 1_000_003
    pk2 = rng.integers(0, 3, np_).astype(np.int32)
    bv, pv = rng.integers(0, 10, nb) > 0, rng.integers(0, 10, np_) > 0
    bkeys = gpu.pack_keys([gpu.Column.from_numpy(bk1), gpu.Column.from_numpy(bk2, validity=bv)], key_bytes=16)
    pkeys = gpu.pack_keys([gpu.Column.from_numpy(pk1), gpu.Column.from_numpy(pk2, validity=pv)], key_bytes=16)
    j = gpu.HashJoin(nb, key_bytes=16)
    j.add_block(bkeys)
    j.final_build()
    gp, gb = j.probe_block(pkeys)
    by_key = {}
    for r in range(nb):
        if bv[r]:
            by_key.setdefault((int(bk1[r]), int(bk2[r])), []).append(r)
    exp = [(i, r) for i in range(np_) if pv[i] for r in by_key.get((int(pk1[i]), int(pk2[i])), [])]
    assert list(zip(gp.tolist(), gb.tolist())) == exp and len(exp) > 100
    marks = j.probe_mark(pkeys)
    em = np.zeros(np_, bool)
    em[[i for i, _ in exp]] = True
    assert np.array_equal(marks, em)
    # the same join on 8-byte packed keys of a single narrow column (KeysU32 zero-extended)
    j8 = gpu.HashJoin(nb)
    b8 = gpu.pack_keys([gpu.Column.from_numpy(bk2, validity=bv)], key_bytes=8)
    p8 = gpu.pack_keys([gpu.Column.from_numpy(pk2[:2000], validity=pv[:2000])], key_bytes=8)
    j8.add_block(b8)
    j8.final_build()
    m8 = j8.probe_mark(p8)
    have = set(int(x) for x, ok in zip(bk2, bv) if ok)
    assert np.array_equal(m8, np.array([bool(ok) and int(x) in have for x, ok in zip(pk2[:2000], pv[:2000])]))


def sort_cases(rng, n):
    f = (rng.standard_normal(n) * 3).astype(np.float32)
    if n > 10:
        f[:6] = [np.nan, -0.0, 0.0, np.inf, -np.inf, np.nan]
    return [
        (T.T_I64, rng.integers(-50, 50, n).astype(np.int64)), (T.T_I32, rng.integers(-2**31, 2**31 - 1, n).astype(np.int32)),
        (T.T_U8, rng.integers(0, 4, n).astype(np.uint8)), (T.T_F32, f), (T.T_F64, np.round(rng.standard_normal(n), 1)),
        (T.T_U64, rng.integers(0, 2**64 - 1, n, dtype=np.uint64)), (T.T_I16, rng.integers(-3, 3, n).astype(np.int16)),
    ]


@pytest.mark.parametrize("n", [1, 2, 255, 2049, 4096, 12_289, 100_000])
def test_sort_perm_matches_oracle(gpu, oracle, n):
    rng = np.random.default_rng(n)
    cases = sort_cases(rng, n)
    valid = rng.integers(0, 5, n) > 0
    combos = [([0], [0], [0], 0), ([1], [1], [0], 0), ([2, 0], [0, 1], [0, 0], 0), ([3], [0], [1], 0), ([4, 2], [1, 0], [1, 0], 0),
              ([5], [1], [0], 7), ([6, 3, 1], [0, 1, 0], [0, 1, 0], 0), ([2, 6, 0], [1, 1, 1], [0, 0, 0], 10)]
    for idxs, desc, nf, limit in combos:
        gcols, hcols = [], []
        for pos, i in enumerate(idxs):
            code, arr = cases[i]
            v = valid if pos == 0 and i != 5 else None
            gcols.append(gpu.Column.from_numpy(arr, code, validity=v))
            hcols.append(O.HostCol(code, arr, v))
        got = gpu.sort_perm(gcols, desc, nf, limit)
        m = limit if 0 < limit < n else n
        exp = np.zeros(max(m, 1), np.uint32)
        d = (C.c_uint8 * len(idxs))(*desc)
        f = (C.c_uint8 * len(idxs))(*nf)
        oracle.orc_sort_perm(O.cols(hcols), d, f, len(idxs), C.c_int64(n), C.c_int64(limit), exp.ctypes.data_as(C.c_void_p))
        assert np.array_equal(got, exp[:m]), (idxs, desc, nf, limit)


def _bound_partition_both(gpu, oracle, gkeys, hkeys, gb, hb, desc, nf, n, nb):
    part, counts = gpu.sort_bound_partition(gkeys, gb, desc, nf)
    got = part.to_numpy(np.uint32, n)
    exp, ecnt = np.zeros(max(n, 1), np.uint32), np.zeros(nb + 1, np.uint64)
    d = (C.c_uint8 * len(desc))(*desc)
    f = (C.c_uint8 * len(nf))(*nf)
    assert oracle.orc_sort_bound_partition(O.cols(hkeys), O.cols(hb) if nb else None, d, f, len(desc), C.c_int64(n), C.c_int64(nb),
                                           exp.ctypes.data_as(C.c_void_p), ecnt.ctypes.data_as(C.c_void_p)) == 0
    assert np.array_equal(got, exp[:n]) and np.array_equal(counts, ecnt)
    assert int(counts.sum()) == n
    return got


@pytest.mark.parametrize("n,nb", [(1, 1), (255, 0), (2049, 1), (12_289, 7), (70_001, 63), (30_000, 20), (60_000, 300), (12_000, 5000)])
def test_sort_bound_partition_matches_oracle(gpu, oracle, n, nb):
    """dbhip_sort_bound_partition (the distributed sort's range partition: sort_spill.rs:1008-1040 partition_point over Bounds,
    rows <= bound[i] belong to range i) against the oracle's row-at-a-time statement: every key type of dbhip_sort_perm, asc /
    desc, NULLs first / last on rows AND bounds, duplicate bounds, bounds in LDS (<= 32 KB) and in global memory, the LDS
    histogram (<= 2048 ranges) and the global one. The bounds are rows of the same distribution, ordered by the oracle's sort."""
    rng = np.random.default_rng(n + nb)
    cases, bcases = sort_cases(rng, n), sort_cases(rng, max(nb, 1))
    valid, bvalid = rng.integers(0, 5, n) > 0, rng.integers(0, 5, max(nb, 1)) > 0
    combos = [([0], [0], [0]), ([1], [1], [0]), ([2, 0], [0, 1], [0, 0]), ([3], [0], [1]), ([4, 2], [1, 0], [1, 0]), ([5], [1], [0]),
              ([6, 3, 1], [0, 1, 0], [0, 1, 0]), ([2, 6, 0], [1, 1, 1], [0, 0, 0])]
    for idxs, desc, nf in combos:
        d = (C.c_uint8 * len(idxs))(*desc)
        f = (C.c_uint8 * len(idxs))(*nf)
        raw = []
        for pos, i in enumerate(idxs):
            code, arr = bcases[i]
            raw.append((code, arr[:nb], bvalid[:nb] if pos == 0 and i != 5 else None))
        order = np.zeros(max(nb, 1), np.uint32)
        if nb:
            oracle.orc_sort_perm(O.cols([O.HostCol(c, a, v) for c, a, v in raw]), d, f, len(idxs), C.c_int64(nb), C.c_int64(0), order.ctypes.data_as(C.c_void_p))
        order = order[:nb]
        gkeys, hkeys, gb, hb = [], [], [], []
        for pos, i in enumerate(idxs):
            code, arr = cases[i]
            v = valid if pos == 0 and i != 5 else None
            gkeys.append(gpu.Column.from_numpy(arr, code, validity=v))
            hkeys.append(O.HostCol(code, arr, v))
            bc, ba, bv = raw[pos]
            ba = np.ascontiguousarray(ba[order])
            bv = bv[order] if bv is not None else None
            if nb:
                gb.append(gpu.Column.from_numpy(ba, bc, validity=bv))
                hb.append(O.HostCol(bc, ba, bv))
        got = _bound_partition_both(gpu, oracle, gkeys, hkeys, gb, hb, desc, nf, n, nb)
        assert got.max() <= nb
    # ONE key without NULLs and at most 63 bounds takes the register path (bounds in registers, ballot counting): every key type
    for i, (code, arr) in enumerate(cases if nb <= 63 else []):
        for desc in ([0], [1]):
            d = (C.c_uint8 * 1)(*desc)
            z = (C.c_uint8 * 1)(0)
            ba = bcases[i][1][:nb]
            order = np.zeros(max(nb, 1), np.uint32)
            if nb:
                oracle.orc_sort_perm(O.cols([O.HostCol(code, ba)]), d, z, 1, C.c_int64(nb), C.c_int64(0), order.ctypes.data_as(C.c_void_p))
            ba = np.ascontiguousarray(ba[order[:nb]])
            _bound_partition_both(gpu, oracle, [gpu.Column.from_numpy(arr, code)], [O.HostCol(code, arr)], [gpu.Column.from_numpy(ba, code)] if nb else [],
                                  [O.HostCol(code, ba)] if nb else [], desc, [0], n, nb)


def test_sort_bound_partition_strings_and_decimal128(gpu, oracle):
    """String keys (inline and beyond 12 bytes: memcmp order, a proper prefix first) and Decimal128 keys, NULL bounds, and rows
    equal to a bound (they belong to the bound's own range)."""
    rng = np.random.default_rng(77)
    base = [b"", b"a", b"Customer#000000001", b"Customer#000000002", b"Customer#00000000", b"Customer#000000001\x00", b"x" * 70, b"x" * 69 + b"y",
            b"abcdefghijkl", b"abcdefghijklm", b"\xff" * 13, b"\xff" * 12, b"a\x00b" * 9]
    n = 20_000
    for pool in (base, [b for b in base if len(b) <= 12]):        # long strings (image parts from the data buffers) / inline views only
        strs = [pool[i] + (b"%d" % rng.integers(0, 50) if rng.random() < 0.5 else b"") for i in rng.integers(0, len(pool), n)]
        if pool is not base:
            strs = [x[:12] for x in strs]
        k2 = rng.integers(0, 3, n).astype(np.int32)
        valid = rng.integers(0, 9, n) > 0
        for desc, nf in ((0, 0), (1, 1), (0, 1)):
            # bounds: every 900th row of the ordered table (so many rows EQUAL a bound), incl. a NULL bound when NULLs come first
            perm = gpu.sort_perm([gpu.Column.strings(strs, validity=valid), gpu.Column.from_numpy(k2)], desc=[desc, 0], nulls_first=[nf, 0])
            pick = perm[::900]
            bs, bk, bv = [strs[i] for i in pick], k2[pick], valid[pick]
            v, buf = make_views_general(strs)
            vb, bufb = make_views_general(bs)
            hkeys = [O.HostCol(T.T_STRING, v, valid, buffers=[buf]), O.HostCol(T.T_I32, k2)]
            hb = [O.HostCol(T.T_STRING, vb, bv, buffers=[bufb]), O.HostCol(T.T_I32, bk)]
            gkeys = [gpu.Column.strings(strs, validity=valid), gpu.Column.from_numpy(k2)]
            gb = [gpu.Column.strings(bs, validity=bv), gpu.Column.from_numpy(np.ascontiguousarray(bk))]
            got = _bound_partition_both(gpu, oracle, gkeys, hkeys, gb, hb, [desc, 0], [nf, 0], n, len(pick))
            assert all(got[i] == j for j, i in enumerate(pick) if j == 0 or (strs[pick[j - 1]], k2[pick[j - 1]], valid[pick[j - 1]]) != (strs[i], k2[i], valid[i]))
    ints = [int(x) * int(y) for x, y in zip(rng.integers(-2**62, 2**62, n), rng.integers(0, 2**40, n))]
    bints = sorted(ints[::1500], reverse=True)
    got = _bound_partition_both(gpu, oracle, [gpu.Column.decimal128(ints, 38, 0)], [O.HostCol(T.T_DEC128, O.i128_array(ints))],
                                [gpu.Column.decimal128(bints, 38, 0)], [O.HostCol(T.T_DEC128, O.i128_array(bints))], [1], [0], n, len(bints))
    assert got.tolist() == [sum(1 for b in bints if b > x) for x in ints]


def test_range_partitioned_sort_operators_on_the_device(gpu):
    """databend_amd.sort_ops.SortDeviceOps — the three device operators of dist.range_partitioned_sort — driven like three ranks
    would drive them (samples -> ordered rows -> Bounds -> partition every shard -> range r of every shard -> sort): the ranges
    concatenate to the full sort, and the whole plan over RCCL at world size 1 returns the plain sort."""
    import torch
    from databend_amd.sort_bounds import balanced_cuts
    from databend_amd.sort_ops import SortDeviceOps
    rng = np.random.default_rng(5)
    n, world = 90_000, 3
    k0 = rng.integers(-40, 40, n).astype(np.int32)
    k1 = rng.standard_normal(n)
    v0 = (rng.random(n) > 0.1).astype(np.uint8)
    pay = np.arange(n, dtype=np.int64)
    cuts = [0, 20_000, 20_000, n]                                  # an empty shard in the middle
    ops, desc, nf = SortDeviceOps(torch), [1, 0], [1, 0]
    shards = [[torch.from_numpy(a[cuts[r]:cuts[r + 1]].copy()).cuda() for a in (k0, k1, pay, v0)] for r in range(world)]
    samples = [[c[torch.arange(0, c.shape[0], 97, device="cuda")] for c in (s[0], s[1], s[3])] for s in shards]
    allk0, allk1, allv = (torch.cat([s[i] for s in samples]) for i in range(3))
    rows = ops.ordered_rows([allk0, allk1], [allv, None], desc, nf)
    bounds = balanced_cuts(rows, world)
    assert len(bounds) == world - 1
    grouped = [ops.partition(s, [0, 1], [3, None], bounds, desc, nf) for s in shards]
    outs = []
    for r in range(world):
        recv = []
        for c in range(4):
            pieces = []
            for (flat, counts) in grouped:
                counts = counts + [0] * (world - len(counts))
                at = sum(counts[:r])
                pieces.append(flat[c][at:at + counts[r]])
            recv.append(torch.cat(pieces))
        outs.append(ops.sort(recv, [0, 1], [3, None], desc, nf))
    full = [torch.cat([o[c] for o in outs]).cpu().numpy() for c in range(4)]
    exp = gpu.sort_perm([gpu.Column.from_numpy(k0, validity=v0 != 0), gpu.Column.from_numpy(k1)], desc, nf)
    assert sorted(full[2].tolist()) == list(range(n))
    nulls_k0 = np.where(v0[exp] != 0, k0[exp], 0)
    assert np.array_equal(np.where(full[3] != 0, full[0], 0), nulls_k0) and np.array_equal(full[1], k1[exp]) and np.array_equal(full[3], v0[exp])
    assert np.array_equal(k1[full[2]], full[1]) and np.array_equal(v0[full[2]], full[3])
    sizes = [int(o[0].shape[0]) for o in outs]
    assert max(sizes) < 0.4 * n, sizes


def test_range_partitioned_sort_on_one_rank_over_rccl(gpu):
    """the whole distributed sort plan with the device operators and the nccl (= RCCL) backend in a world of one: the sample
    all-gather and the per-column all-to-all really run on device tensors; the multi-rank logic is covered by
    tests/test_dist_gloo.py with world 2 / 3."""
    import socket

    import torch
    import torch.distributed as dist

    from databend_amd import dist as DX
    from databend_amd.sort_ops import SortDeviceOps
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
    try:
        rng = np.random.default_rng(31)
        n = 300_000
        k0 = rng.integers(0, 1000, n).astype(np.int64)
        k1 = rng.integers(-2**31, 2**31 - 1, n).astype(np.int32)
        pay = rng.standard_normal(n).astype(np.float32)
        cols, valids, bounds = DX.range_partitioned_sort([torch.from_numpy(a).cuda() for a in (k0, k1, pay)], [0, 1], SortDeviceOps(torch), dist, torch,
                                                         desc=[0, 1])
        assert bounds == [] and valids == [None, None, None]
        exp = gpu.sort_perm([gpu.Column.from_numpy(k0), gpu.Column.from_numpy(k1)], [0, 1])
        for got, src in zip(cols, (k0, k1, pay)):
            assert np.array_equal(got.cpu().numpy(), src[exp])
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n", [65_536, 300_001])
def test_sort_limit_radix_select(gpu, oracle, n):
    """LIMIT (sort_compare.rs:197-209): the device path radix-selects a threshold on the most significant key and
    sorts only the candidates; the result must be the first `limit` rows of the full stable sort."""
    rng = np.random.default_rng(n + 17)
    f = (rng.standard_normal(n) * 1e3).astype(np.float32)
    f[:4] = [np.nan, -0.0, 0.0, -np.inf]
    lowcard = rng.integers(0, 5, n).astype(np.int64)          # heavy ties on the first key: candidates = whole groups
    i32 = rng.integers(-2**31, 2**31 - 1, n).astype(np.int32)
    small = rng.integers(1000, 1256, n).astype(np.int64)      # constant upper bytes: their passes are skipped
    combos = [([(T.T_F32, f)], [1], 10), ([(T.T_F32, f)], [0], 10), ([(T.T_I64, lowcard), (T.T_I32, i32)], [0, 1], 100),
              ([(T.T_I32, i32)], [0], 20_000), ([(T.T_I64, small), (T.T_F32, f)], [1, 0], 1), ([(T.T_I32, i32), (T.T_I64, lowcard)], [1, 0], 9000)]
    for cols, desc, limit in combos:
        gcols = [gpu.Column.from_numpy(a, c) for c, a in cols]
        hcols = [O.HostCol(c, a) for c, a in cols]
        got = gpu.sort_perm(gcols, desc, [0] * len(cols), limit)
        exp = np.zeros(limit, np.uint32)
        d = (C.c_uint8 * len(cols))(*desc)
        z = (C.c_uint8 * len(cols))(*([0] * len(cols)))
        oracle.orc_sort_perm(O.cols(hcols), d, z, len(cols), C.c_int64(n), C.c_int64(limit), exp.ctypes.data_as(C.c_void_p))
        assert np.array_equal(got, exp), (desc, limit)
    # Decimal128 most significant key (two 64-bit parts; the select runs on the high part)
    ints = [int(x) * int(y) for x, y in zip(rng.integers(-2**62, 2**62, n), rng.integers(0, 2**40, n))]
    got = gpu.sort_perm([gpu.Column.decimal128(ints, 38, 0)], [1], [0], 10)
    exp = sorted(range(n), key=lambda i: (-ints[i], i))[:10]
    assert got.tolist() == exp


@pytest.mark.parametrize("nruns,limit", [(1, 0), (3, 0), (8, 0), (8, 25), (5, 100_000)])
def test_merge_sorted_runs_like_the_loser_tree(gpu, nruns, limit):
    """Merger semantics (sorts/core/merger.rs, loser_tree.rs basic test :126-156): the merged key sequence of N sorted
    streams == heapq.merge of the runs; ties between streams are unordered in the reference, so key tuples are
    compared, plus: every run's rows appear in their original order."""
    import heapq
    rng = np.random.default_rng(nruns * 7 + limit)
    runs = []
    for r in range(nruns):
        m = int(rng.integers(0, 60_000)) if r % 3 else int(rng.integers(0, 50))
        a = rng.integers(-5, 6, m).astype(np.int32)
        b = rng.integers(0, 2**40, m).astype(np.int64)
        order = np.lexsort((-b, a))  # a asc, b desc
        runs.append((a[order], b[order]))
    offs = np.concatenate([[0], np.cumsum([len(a) for a, _ in runs])]).astype(np.int64)
    ca = np.concatenate([a for a, _ in runs]) if nruns else np.zeros(0, np.int32)
    cb = np.concatenate([b for _, b in runs]) if nruns else np.zeros(0, np.int64)
    perm = gpu.merge_sorted_perm([gpu.Column.from_numpy(ca), gpu.Column.from_numpy(cb)], offs, desc=[0, 1], limit=limit)
    exp = list(heapq.merge(*[[(int(x), -int(y)) for x, y in zip(a, b)] for a, b in runs]))
    n = len(exp)
    m = limit if 0 < limit < n else n
    assert len(perm) == m
    assert [(int(ca[i]), -int(cb[i])) for i in perm] == exp[:m]
    run_of = np.searchsorted(offs, perm, side="right") - 1
    for r in range(nruns):
        mine = perm[run_of == r]
        assert np.all(np.diff(mine.astype(np.int64)) > 0)          # in-run order preserved (stable merge)


def test_sort_decimal128_and_bool(gpu, oracle):
    rng = np.random.default_rng(4)
    n = 5000
    ints = [int(x) * int(y) for x, y in zip(rng.integers(-2**62, 2**62, n), rng.integers(0, 2**40, n))]
    bools = rng.integers(0, 2, n).astype(bool)
    got = gpu.sort_perm([gpu.Column.boolean(bools), gpu.Column.decimal128(ints, 38, 0)], [1, 0])
    exp = sorted(range(n), key=lambda i: (-int(bools[i]), ints[i], i))
    assert got.tolist() == exp


def close(got, exp, tol=1e-5, scale=None):
    """f32 distances: |got - exp| <= tol * max(1, |exp|, scale) — north_star's 1e-5 relative, measured against the
    natural scale of the reduction: a dot product of two vectors is only defined to ~eps * ||a|| * ||b||
    (different summation orders of the SAME f32 products differ by that much), and a cosine distance is a
    difference of O(1) quantities."""
    got, exp = np.asarray(got, np.float64), np.asarray(exp, np.float64)
    both_nan = np.isnan(got) & np.isnan(exp)
    bound = np.maximum(1.0, np.abs(exp))
    if scale is not None:
        bound = np.maximum(bound, scale)
    return bool(np.all(both_nan | (np.abs(got - exp) <= tol * bound)))


@pytest.mark.parametrize("n,dim,nq", [(1, 3, 1), (100, 8, 3), (1000, 128, 17), (5000, 768, 130), (333, 100, 200)])
def test_vec_distance_matches_oracle(gpu, oracle, n, dim, nq):
    rng = np.random.default_rng(n + dim)
    base = rng.standard_normal((n, dim)).astype(np.float32)
    q = rng.standard_normal((nq, dim)).astype(np.float32)
    if n > 2:
        base[1] = q[0]          # identical vectors
        base[2] = 0.0           # zero vector -> cosine NaN
    gb, gq = gpu.VectorColumn(base), gpu.VectorColumn(q)
    for metric in range(4):
        got = gpu.vec_distance(metric, gb, gq)
        exp = np.zeros((nq, n), np.float32)
        oracle.orc_vec_distance(metric, base.ctypes.data_as(C.c_void_p), C.c_int64(n), dim, q.ctypes.data_as(C.c_void_p), nq, exp.ctypes.data_as(C.c_void_p))
        scale = None
        if metric == T.VEC_DOT:
            scale = np.outer(np.linalg.norm(q.astype(np.float64), axis=1), np.linalg.norm(base.astype(np.float64), axis=1))
        assert close(got, exp, 2e-5 if metric == 1 else 1e-5, scale), (metric, np.abs(got - exp).max())


def test_vector_golden_cases(gpu):
    """Known answers from the reference's vector.txt (column cases; tests/golden/vector.json)."""
    cases = json.load(open(os.path.join(HERE, "golden", "vector.json")))["cases"]
    names = {"cosine_distance": T.VEC_COSINE, "l1_distance": T.VEC_L1, "l2_distance": T.VEC_L2, "inner_product": T.VEC_DOT}
    checked = 0
    for c in cases:
        fn = c["ast"].split("(")[0]
        cols = c["columns"]
        if fn not in names or not all(k in cols for k in "abcd"):
            continue
        if any(cols[k]["type"].replace(" NULL", "") not in ("Float32", "Float64") for k in "abcd"):
            continue
        val = lambda k: np.array([float(x) for x in cols[k]["values"]], np.float32)
        n = c["n"]
        exp = np.array([float(x) for x in cols["Output"]["values"]], np.float64)
        for r in range(n):
            a = np.array([[val("a")[r], val("b")[r]]], np.float32)
            b = np.array([[val("c")[r], val("d")[r]]], np.float32)
            got = gpu.vec_distance(names[fn], gpu.VectorColumn(a), gpu.VectorColumn(b))[0, 0]
            assert abs(float(got) - exp[r]) <= 1e-5 * max(1.0, abs(exp[r])), (c["ast"], r, got, exp[r])
        checked += 1
    assert checked >= 4


@pytest.mark.parametrize("n,dim,nq,k", [(5, 16, 2, 10), (3000, 64, 33, 10), (20_000, 128, 4, 16), (70_000, 32, 300, 1)])
def test_vec_topk_matches_exact_oracle_topk(gpu, oracle, n, dim, nq, k):
    rng = np.random.default_rng(n + k)
    base = rng.standard_normal((n, dim)).astype(np.float32)
    q = rng.standard_normal((nq, dim)).astype(np.float32)
    gb, gq = gpu.VectorColumn(base), gpu.VectorColumn(q)
    for metric in (T.VEC_COSINE, T.VEC_L2, T.VEC_DOT):
        idx, dist = gpu.vec_topk(metric, gb, gq, k)
        full = gpu.vec_distance(metric, gb, gq)           # same kernels -> identical floats
        for qi in range(nq):
            order = np.lexsort((np.arange(n), full[qi]))[:k]  # ascending distance, ties by lower row id
            kk = min(k, n)
            assert np.array_equal(idx[qi, :kk], order[:kk].astype(np.uint32)), (metric, qi)
            assert np.array_equal(dist[qi, :kk], full[qi, order[:kk]])
            assert (idx[qi, kk:] == 0xFFFFFFFF).all()
        # recall@k against the oracle's exact ordering (different summation order -> near ties may swap)
        exp = np.zeros((nq, n), np.float32)
        oracle.orc_vec_distance(metric, base.ctypes.data_as(C.c_void_p), C.c_int64(n), dim, q.ctypes.data_as(C.c_void_p), nq, exp.ctypes.data_as(C.c_void_p))
        hits = sum(len(set(idx[qi, :min(k, n)].tolist()) & set(np.argsort(exp[qi], kind="stable")[:k].tolist())) for qi in range(nq))
        assert hits >= 0.99 * nq * min(k, n)


@pytest.mark.parametrize("n,dim,nq,k,kind", [(5, 16, 2, 10, "normal"), (3000, 64, 33, 10, "normal"), (150_000, 64, 70, 10, "normal"),
                                             (131_072, 100, 5, 16, "normal"), (200_000, 48, 300, 3, "clustered"), (140_000, 36, 9, 10, "special"),
                                             # the 256 x 256 8-phase filter kernel (k-tiles pair up, > 128 queries): ragged query / base tiles
                                             (150_000, 128, 300, 10, "normal"), (70_001, 200, 513, 5, "normal"), (40_000, 768, 257, 10, "clustered"),
                                             (140_000, 128, 200, 10, "special")])
def test_vec_index_is_exact(gpu, oracle, n, dim, nq, k, kind):
    """dbhip_vec_index_search (bf16 pre-filter + exact re-score) returns the exact top-k: the same row ids as the
    exact f32 scan dbhip_vec_topk and distances within 1e-5 relative of the oracle's f32 distances (north_star
    tolerance). 'clustered': many near-duplicate rows (large candidate sets, candidate-list overflow fallback);
    'special': zero vectors, huge/tiny magnitudes, NaN and Inf rows."""
    rng = np.random.default_rng(n + dim)
    base = rng.standard_normal((n, dim)).astype(np.float32)
    q = rng.standard_normal((nq, dim)).astype(np.float32)
    if kind == "clustered":
        centers = rng.standard_normal((8, dim)).astype(np.float32)
        base = (centers[rng.integers(0, 8, n)] + 1e-3 * rng.standard_normal((n, dim))).astype(np.float32)
        q[: nq // 2] = centers[rng.integers(0, 8, nq // 2)] + 1e-3 * rng.standard_normal((nq // 2, dim)).astype(np.float32)
    if kind == "special":
        base[7] = 0.0
        base[70_000] = 0.0
        base[11] *= 1e18
        base[100_001] *= 1e-18
        base[13, 3] = np.nan
        base[90_000, 5] = np.inf
        q[0] = 0.0
        q[1] *= 1e15
    gb, gq = gpu.VectorColumn(base), gpu.VectorColumn(q)
    for metric in (T.VEC_COSINE, T.VEC_DOT, T.VEC_L2):
        ix = gpu.VectorIndex(metric, gb)
        idx, dist = ix.search(gq, k)
        eidx, edist = gpu.vec_topk(metric, gb, gq, k)
        kk = min(k, n)
        exp = np.zeros((nq, n), np.float32)
        oracle.orc_vec_distance(metric, base.ctypes.data_as(C.c_void_p), C.c_int64(n), dim, q.ctypes.data_as(C.c_void_p), nq, exp.ctypes.data_as(C.c_void_p))
        mism = 0
        for qi in range(nq):
            if not np.array_equal(idx[qi, :kk], eidx[qi, :kk]):
                # the two exact paths sum in different orders: rows may swap only where their distances tie to rounding
                a, b = np.nan_to_num(edist[qi, :kk], nan=np.inf, posinf=3e38), np.nan_to_num(dist[qi, :kk], nan=np.inf, posinf=3e38)
                assert np.allclose(a, b, rtol=2e-5, atol=2e-6), (metric, qi)
                mism += 1
            for j in range(kk):
                r = int(idx[qi, j])
                ev, gv = float(exp[qi, r]), float(dist[qi, j])
                if np.isfinite(ev) and abs(ev) < 1e30:
                    sc = float(np.abs(base[r].astype(np.float64) * q[qi]).sum()) if metric == T.VEC_DOT else 1.0
                    assert abs(gv - ev) <= 1e-5 * max(1.0, abs(ev), sc) + 2e-7, (metric, qi, j, gv, ev)
            assert (idx[qi, kk:] == 0xFFFFFFFF).all()
        assert mism <= max(1, nq // 4 if kind == "clustered" else nq // 50), (metric, mism)
        ix.destroy()
    with pytest.raises(Exception):
        gpu.VectorIndex(T.VEC_L1, gb)   # l1 has no inner-product form to pre-filter on


def test_score_u8_matches_reference_c_kernels(gpu, oracle):
    """dot / l1 over u8-quantised vectors: exact integers; checked against the oracle restatement and, when
    oracle/_ref/libref_u8.so was built from the reference's own cpp/avx2.c, against the real thing."""
    rng = np.random.default_rng(8)
    # quantised components live in [0, 127] (encoded_vectors_u8.rs:239-246: clamp(0, 127)); the reference's AVX
    # kernel relies on that (maddubs treats one operand as signed bytes)
    for dim in (16, 64, 768, 100):
        n = 2000
        base = rng.integers(0, 128, (n, dim)).astype(np.uint8)
        q = rng.integers(0, 128, dim).astype(np.uint8)
        for is_l1 in (0, 1):
            got = gpu.score_u8(is_l1, q, base)
            exp = np.zeros(n, np.float32)
            oracle.orc_score_u8(is_l1, q.ctypes.data_as(C.c_void_p), base.ctypes.data_as(C.c_void_p), C.c_int64(n), dim, exp.ctypes.data_as(C.c_void_p))
            assert np.array_equal(got, exp)
    ref = os.path.join(os.path.dirname(HERE), "oracle", "_ref", "libref_u8.so")
    if os.path.exists(ref):
        R = C.CDLL(ref)
        R.impl_score_dot_avx.restype = C.c_float
        R.impl_score_l1_avx.restype = C.c_float
        dim, n = 768, 500
        base = rng.integers(0, 128, (n, dim)).astype(np.uint8)
        q = rng.integers(0, 128, dim).astype(np.uint8)
        for is_l1, f in ((0, R.impl_score_dot_avx), (1, R.impl_score_l1_avx)):
            got = gpu.score_u8(is_l1, q, base)
            exp = np.array([f(q.ctypes.data_as(C.c_void_p), base[i].ctypes.data_as(C.c_void_p), C.c_uint32(dim)) for i in range(n)], np.float32)
            assert np.array_equal(got, exp)


def test_vec_topk_known_answers_from_sqllogictest(gpu):
    """Exact (table t1) top-5 of the reference's 09_0000_vector_index_base.test:108-200 through
    dbhip_vec_topk (ids and distances)."""
    g = json.load(open(os.path.join(HERE, "golden", "vector_topk.json")))
    base = gpu.VectorColumn(np.array(g["base"], np.float32))
    names = {"cosine_distance": T.VEC_COSINE, "l1_distance": T.VEC_L1, "l2_distance": T.VEC_L2}
    for q in g["queries"]:
        idx, dist = gpu.vec_topk(names[q["fn"]], base, gpu.VectorColumn(np.array([q["query"]], np.float32)), 5)
        assert [int(i) + 1 for i in idx[0]] == [e[0] for e in q["expected"]], q["fn"]
        for d, (_, ev) in zip(dist[0], q["expected"]):
            assert abs(float(d) - ev) <= 1e-5 * max(1.0, abs(ev)) + 2e-7


def test_sort_short_string_keys(gpu):
    """ORDER BY on inline (<= 12 byte) string keys: memcmp order, a proper prefix first; desc; with a second key."""
    rng = np.random.default_rng(12)
    alphabet = [b"", b"a", b"ab", b"abc", b"abd", b"b", b"ba", b"abcdefghijkl", b"abcdefghijk", b"abcdefgh", b"abcdefghi", b"\xff", b"a\x00", b"zz"]
    n = 5000
    strs = [alphabet[i] for i in rng.integers(0, len(alphabet), n)]
    k2 = rng.integers(0, 5, n).astype(np.int32)
    for desc in (0, 1):
        perm = gpu.sort_perm([gpu.Column.strings(strs), gpu.Column.from_numpy(k2)], desc=[desc, 0])
        got = [(strs[i], int(k2[i])) for i in perm]
        exp = sorted(((s, int(b)) for s, b in zip(strs, k2)), key=lambda t: (t[0], t[1]))
        if desc:
            exp = sorted(exp, key=lambda t: t[1])
            exp = sorted(exp, key=lambda t: t[0], reverse=True)
        assert got == exp


@pytest.mark.parametrize("n", [7, 5000, 200_000])
def test_sort_string_keys_of_any_length(gpu, n):
    """ORDER BY on String keys beyond the 12 inline bytes (sorts/core/row_convert/variable.rs: memcmp order of the bytes, a
    proper prefix first): c_name / c_comment-like values of 0..70 bytes sharing long prefixes, embedded NULs and 0xFF bytes,
    NULLs first / last, asc / desc, a second key behind it, and LIMIT — against Python's bytes ordering."""
    rng = np.random.default_rng(n)
    base = [b"", b"a", b"Customer#000000001", b"Customer#000000002", b"Customer#00000000", b"Customer#000000001\x00", b"x" * 70, b"x" * 69 + b"y",
            b"x" * 69, b"abcdefghijkl", b"abcdefghijklm", b"\xff" * 13, b"\xff" * 12, b"a\x00b" * 9, b"mid-length value 21.."]
    strs = [base[i] + (b"%d" % rng.integers(0, 50) if rng.random() < 0.5 else b"") for i in rng.integers(0, len(base), n)]
    k2 = rng.integers(0, 3, n).astype(np.int32)
    valid = rng.integers(0, 9, n) > 0
    for desc, nulls_first in ((0, 0), (1, 1), (0, 1)):
        perm = gpu.sort_perm([gpu.Column.strings(strs, validity=valid), gpu.Column.from_numpy(k2)], desc=[desc, 0], nulls_first=[nulls_first, 0])
        got = [((strs[i] if valid[i] else None), int(k2[i])) for i in perm]
        rows = sorted(zip(strs, k2.tolist(), valid.tolist()), key=lambda t: t[1])
        nulls = [(None, b) for s_, b, v in rows if not v]
        vals = sorted(((s_, b) for s_, b, v in rows if v), key=lambda t: t[0], reverse=bool(desc))
        # (a stable sort on the string after sorting on k2 keeps k2 ascending inside equal strings)
        exp = nulls + vals if nulls_first else vals + nulls
        assert got == exp
    if n >= 5000:
        top = gpu.sort_perm([gpu.Column.strings(strs)], limit=25)
        assert [strs[i] for i in top] == sorted(strs)[:25]


@pytest.mark.parametrize("kind", ["inner", "left", "left_semi", "left_anti"])
@pytest.mark.parametrize("nb,np_,card", [(0, 10, 5), (10, 0, 5), (1000, 5000, 300), (120_000, 300_000, 90_000), (5000, 5000, 7)])
def test_join_kinds_output_assembly(gpu, oracle, kind, nb, np_, card):
    """Inner / left-outer / left-semi / left-anti output blocks (new_hash_join/memory/{inner_join,left_join,left_join_semi,
    left_join_anti}.rs) assembled on the device from the pairs, the matched Bitmap, take and the nullable take; expected
    rows from the oracle's inner pairs. Row order inside a block is the library's (pairs by probe row, then the unmatched probe
    rows — left_join.rs:196-232 emits them last too); the comparison is on the multiset of rows."""
    rng = np.random.default_rng(nb * 3 + np_)
    bk = rng.integers(0, card, nb).astype(np.uint64) * np.uint64(2654435761)
    pk = rng.integers(0, card + card // 2 + 1, np_).astype(np.uint64) * np.uint64(2654435761)
    bvalid = rng.integers(0, 10, nb) > 0
    pvalid = rng.integers(0, 10, np_) > 0
    bpay = rng.integers(-10**9, 10**9, nb).astype(np.int64)          # a build payload column (nullable itself)
    bpay_valid = rng.integers(0, 5, nb) > 0
    ppay = rng.integers(0, 2**31, np_).astype(np.int32)              # a probe payload column
    j = gpu.HashJoin(16)
    if nb:
        j.add_block(gpu.Column.from_numpy(bk, validity=bvalid))
    j.final_build()
    probe_cols = [gpu.Column.from_numpy(ppay), gpu.Column.from_numpy(np.arange(np_, dtype=np.uint32))]
    build_cols = [gpu.Column.from_numpy(bpay, validity=bpay_valid)]
    pc, bc, rows = j.join(kind, gpu.Column.from_numpy(pk, validity=pvalid), probe_cols, build_cols)
    # expected from the oracle's inner pairs
    cap = max(nb * max(np_, 1) // max(card, 1) * 2 + np_ + 64, 1024)
    ep, eb = np.zeros(cap, np.uint32), np.zeros(cap, np.uint32)
    bv = np.concatenate([np.packbits(bvalid, bitorder="little"), np.zeros(8, np.uint8)])
    pv = np.concatenate([np.packbits(pvalid, bitorder="little"), np.zeros(8, np.uint8)])
    total = oracle.orc_join_inner_u64(bk.ctypes.data_as(C.c_void_p), bv.ctypes.data_as(C.c_void_p), C.c_int64(nb), pk.ctypes.data_as(C.c_void_p),
                                      pv.ctypes.data_as(C.c_void_p), C.c_int64(np_), ep.ctypes.data_as(C.c_void_p), eb.ctypes.data_as(C.c_void_p), C.c_int64(cap))
    assert total <= cap
    ep, eb = ep[:total], eb[:total]
    matched = np.zeros(np_, dtype=bool)
    matched[ep] = True
    got_p = [c.to_numpy() for c in pc]
    if kind == "left_semi":
        exp_rows = sorted(np.nonzero(matched)[0].tolist())
        assert rows == len(exp_rows) and got_p[1].tolist() == exp_rows and np.array_equal(got_p[0], ppay[exp_rows])
        return
    if kind == "left_anti":
        exp_rows = sorted(np.nonzero(~matched)[0].tolist())
        assert rows == len(exp_rows) and got_p[1].tolist() == exp_rows and np.array_equal(got_p[0], ppay[exp_rows])
        return
    gb_vals = bc[0].to_numpy()
    gb_valid = bc[0].validity_numpy() if hasattr(bc[0], "validity_numpy") else gpu.unpack_bits(bc[0].validity.to_numpy(np.uint8, (rows + 7) // 8), rows) \
        if bc[0].validity is not None else np.ones(rows, dtype=bool)
    exp = [(int(p), int(ppay[p]), (int(bpay[b]) if bpay_valid[b] else None)) for p, b in zip(ep, eb)]
    if kind == "left":
        exp += [(int(p), int(ppay[p]), None) for p in np.nonzero(~matched)[0]]
        # ... and the null block's rows (the unmatched probe rows, last) hold zero values
        assert np.all(gb_vals[total:] == 0) and not gb_valid[total:].any()
    got = [(int(got_p[1][i]), int(got_p[0][i]), (int(gb_vals[i]) if gb_valid[i] else None)) for i in range(rows)]
    assert rows == len(exp) and sorted(got, key=repr) == sorted(exp, key=repr)
    if kind == "left" and rows:
        # matched rows first (probe order), the unmatched probe rows last
        assert np.all(np.diff(got_p[1][:total].astype(np.int64)) >= 0) and np.array_equal(got_p[1][total:], np.nonzero(~matched)[0])


@pytest.mark.parametrize("kind", ["inner", "left", "left_semi", "left_anti", "right", "right_semi", "right_anti", "full"])
@pytest.mark.parametrize("nb,np_,card", [(0, 10, 5), (1000, 5000, 300), (40_000, 100_000, 30_000), (3000, 3000, 7)])
def test_join_kinds_with_another_conjunct(gpu, oracle, kind, nb, np_, card):
    """The `CONJUNCT = true` streams (inner_join.rs:278-310, left_join.rs:262-292, left_join_semi.rs / left_join_anti.rs filter
    streams, right_join.rs:256-290): the key matches go through another predicate — here `probe payload < build payload` over a
    NULLABLE build payload, so a NULL comparison drops the pair — and a probe (build) row all of whose pairs were dropped is
    unmatched. Expected rows from the oracle's inner pairs filtered in numpy; compared as multisets."""
    rng = np.random.default_rng(nb * 5 + np_ + 1)
    bk = rng.integers(0, card, nb).astype(np.uint64) * np.uint64(2654435761)
    pk = rng.integers(0, card + card // 2 + 1, np_).astype(np.uint64) * np.uint64(2654435761)
    bvalid, pvalid = rng.integers(0, 10, nb) > 0, rng.integers(0, 10, np_) > 0
    bpay = rng.integers(0, 1000, nb).astype(np.int64)
    bpay_valid = rng.integers(0, 5, nb) > 0
    ppay = rng.integers(0, 1000, np_).astype(np.int64)
    j = gpu.HashJoin(16)
    if nb:
        j.add_block(gpu.Column.from_numpy(bk, validity=bvalid))
    j.final_build()
    probe_cols = [gpu.Column.from_numpy(ppay), gpu.Column.from_numpy(np.arange(np_, dtype=np.uint32))]
    build_cols = [gpu.Column.from_numpy(bpay, validity=bpay_valid), gpu.Column.from_numpy(np.arange(nb, dtype=np.uint32))]
    conj = lambda jp, jb, m: gpu.cmp(T.CMP_LT, jp[0], jb[0], m)
    pc, bc, rows = j.join(kind, gpu.Column.from_numpy(pk, validity=pvalid), probe_cols, build_cols, conjunct=conj)
    cap = max(nb * max(np_, 1) // max(card, 1) * 2 + np_ + 64, 1024)
    ep, eb = np.zeros(cap, np.uint32), np.zeros(cap, np.uint32)
    bv = np.concatenate([np.packbits(bvalid, bitorder="little"), np.zeros(8, np.uint8)])
    pv = np.concatenate([np.packbits(pvalid, bitorder="little"), np.zeros(8, np.uint8)])
    total = oracle.orc_join_inner_u64(bk.ctypes.data_as(C.c_void_p), bv.ctypes.data_as(C.c_void_p), C.c_int64(nb), pk.ctypes.data_as(C.c_void_p),
                                      pv.ctypes.data_as(C.c_void_p), C.c_int64(np_), ep.ctypes.data_as(C.c_void_p), eb.ctypes.data_as(C.c_void_p), C.c_int64(cap))
    assert total <= cap
    ep, eb = ep[:total].astype(np.int64), eb[:total].astype(np.int64)
    keep = bpay_valid[eb] & (ppay[ep] < bpay[eb]) if total else np.zeros(0, dtype=bool)
    if nb >= 1000:
        assert 0 < keep.sum() < total                                   # the conjunct really drops pairs
    ep, eb = ep[keep], eb[keep]
    pmatched, bmatched = np.zeros(np_, dtype=bool), np.zeros(nb, dtype=bool)
    pmatched[ep] = True
    bmatched[eb] = True

    def valid_of(col, n):
        return gpu.unpack_bits(col.validity.to_numpy(np.uint8, (n + 7) // 8), n) if col.validity is not None else np.ones(n, dtype=bool)

    def got_rows(pcols, bcols, n):
        out = []
        pid = pcols[1].to_numpy() if pcols else None
        pvd = valid_of(pcols[1], n) if pcols else None
        bid = bcols[1].to_numpy() if bcols else None
        bvd = valid_of(bcols[1], n) if bcols else None
        bp = bcols[0].to_numpy() if bcols else None
        bpv = valid_of(bcols[0], n) if bcols else None
        for i in range(n):
            p = int(pid[i]) if pcols and pvd[i] else None
            b = int(bid[i]) if bcols and bvd[i] else None
            pay = (int(bp[i]) if bpv[i] else None) if bcols else None
            out.append((p, b, pay))
        return out

    pairs = [(int(p), int(b), int(bpay[b])) for p, b in zip(ep, eb)]      # a surviving pair always has a valid build payload
    got = got_rows(pc, bc, rows)
    if kind == "inner":
        exp = pairs
    elif kind == "left":
        exp = pairs + [(int(p), None, None) for p in np.nonzero(~pmatched)[0]]
    elif kind == "left_semi":
        exp = [(int(p), None, None) for p in np.nonzero(pmatched)[0]]
    elif kind == "left_anti":
        exp = [(int(p), None, None) for p in np.nonzero(~pmatched)[0]]
    elif kind == "right":
        exp = pairs
    elif kind == "full":
        exp = pairs + [(int(p), None, None) for p in np.nonzero(~pmatched)[0]]
    else:
        exp = []
    assert rows == len(exp) and sorted(got, key=repr) == sorted(exp, key=repr), kind
    if kind in ("right", "right_semi", "right_anti", "full"):
        fp, fb, fk = j.final_probe(kind, build_cols, probe_cols_like=probe_cols)
        tail = got_rows(fp, fb, fk)
        want = np.nonzero(bmatched)[0] if kind == "right_semi" else np.nonzero(~bmatched)[0]
        assert sorted(tail, key=repr) == sorted([(None, int(b), (int(bpay[b]) if bpay_valid[b] else None)) for b in want], key=repr), kind


@pytest.mark.parametrize("nb,np_,card", [(3000, 9000, 500), (60_000, 150_000, 20_000)])
def test_join_on_keys_u256(gpu, oracle, nb, np_, card):
    """Three-column join key (i64, i64, i64 nullable = 25 bytes) -> KeysU256 (method_fixed_keys.rs:58-139, 32-byte packed keys):
    inner pairs == a dictionary join on the tuples, the matched Bitmap, and a left-anti assembly on top."""
    rng = np.random.default_rng(nb + 1)
    bk = [rng.integers(0, card, nb).astype(np.int64) * 7_000_003, rng.integers(-2, 2, nb).astype(np.int64), rng.integers(0, 2, nb).astype(np.int64) << 40]
    pk = [rng.integers(0, card * 2, np_).astype(np.int64) * 7_000_003, rng.integers(-2, 2, np_).astype(np.int64), rng.integers(0, 2, np_).astype(np.int64) << 40]
    bv, pv = rng.integers(0, 10, nb) > 0, rng.integers(0, 10, np_) > 0
    assert gpu.keys_method([gpu.Column.from_numpy(bk[0]), gpu.Column.from_numpy(bk[1]), gpu.Column.from_numpy(bk[2], validity=bv)]) == 32
    bkeys = gpu.pack_keys([gpu.Column.from_numpy(bk[0]), gpu.Column.from_numpy(bk[1]), gpu.Column.from_numpy(bk[2], validity=bv)], key_bytes=32)
    pkeys = gpu.pack_keys([gpu.Column.from_numpy(pk[0]), gpu.Column.from_numpy(pk[1]), gpu.Column.from_numpy(pk[2], validity=pv)], key_bytes=32)
    j = gpu.HashJoin(nb, key_bytes=32)
    j.add_block(bkeys)
    j.final_build()
    gp, gb = j.probe_block(pkeys)
    by_key = {}
    for r in range(nb):
        if bv[r]:
            by_key.setdefault((int(bk[0][r]), int(bk[1][r]), int(bk[2][r])), []).append(r)
    exp = [(i, r) for i in range(np_) if pv[i] for r in by_key.get((int(pk[0][i]), int(pk[1][i]), int(pk[2][i])), [])]
    assert list(zip(gp.tolist(), gb.tolist())) == exp and len(exp) > 100
    marks = j.probe_mark(pkeys)
    em = np.zeros(np_, bool)
    em[[i for i, _ in exp]] = True
    assert np.array_equal(marks, em)
    pc, _, rows = j.join("left_anti", pkeys, [gpu.Column.from_numpy(np.arange(np_, dtype=np.uint32))], [])
    assert rows == int((~em).sum()) and np.array_equal(pc[0].to_numpy(), np.nonzero(~em)[0])


@pytest.mark.parametrize("kind", ["right", "right_semi", "right_anti", "full"])
def test_right_and_full_join_kinds_over_two_probe_blocks(gpu, kind):
    """Right / right-semi / right-anti / full outer joins (new_hash_join/memory/{right_join,right_join_semi,right_join_anti,
    full_join}.rs): the table keeps one matched bit per build row across probe blocks (dbhip_join_mark_build), final_probe
    emits the build rows by that bit with a NULL probe side. Expected rows from a plain Python statement of the join; NULL keys
    never match; the comparison is on the multiset of (probe payload | None, build payload | None) rows."""
    rng = np.random.default_rng(11)
    nb, np_, card = 4000, 5000, 1500
    bk = rng.integers(0, card, nb).astype(np.uint64)
    bvalid = rng.integers(0, 10, nb) > 0
    bpay = rng.integers(-10**9, 10**9, nb).astype(np.int64)
    blocks = []
    for _ in range(2):
        pk = rng.integers(0, card * 2, np_).astype(np.uint64)
        pvalid = rng.integers(0, 10, np_) > 0
        ppay = rng.integers(0, 2**31, np_).astype(np.int32)
        blocks.append((pk, pvalid, ppay))
    j = gpu.HashJoin(16)
    j.add_block(gpu.Column.from_numpy(bk, validity=bvalid))
    j.final_build()
    build_cols = [gpu.Column.from_numpy(bpay)]
    got = []

    def rows_of(pc, bc, n):
        pv = [None] * n
        bv = [None] * n
        if pc:
            vals, ok = pc[0].to_numpy(), pc[0].validity_numpy()
            pv = [int(vals[i]) if ok[i] else None for i in range(n)]
        if bc:
            vals, ok = bc[0].to_numpy(), bc[0].validity_numpy()
            bv = [int(vals[i]) if ok[i] else None for i in range(n)]
        return list(zip(pv, bv))
    like = None
    for pk, pvalid, ppay in blocks:
        pcols = [gpu.Column.from_numpy(ppay)]
        like = pcols
        pc, bc, n = j.join(kind, gpu.Column.from_numpy(pk, validity=pvalid), pcols, build_cols)
        got += rows_of(pc, bc, n)
    pc, bc, n = j.final_probe(kind, build_cols, like)
    tail = rows_of(pc, bc, n)
    got += tail
    # the Python statement
    table = {}
    for b in range(nb):
        if bvalid[b]:
            table.setdefault(int(bk[b]), []).append(b)
    matched_b = np.zeros(nb, bool)
    exp = []
    for pk, pvalid, ppay in blocks:
        for p in range(np_):
            bs = table.get(int(pk[p]), []) if pvalid[p] else []
            for b in bs:
                matched_b[b] = True
                if kind in ("right", "full"):
                    exp.append((int(ppay[p]), int(bpay[b])))
            if not bs and kind == "full":
                exp.append((int(ppay[p]), None))
    if kind in ("right", "full", "right_anti"):
        exp += [(None, int(bpay[b])) for b in np.nonzero(~matched_b)[0]]
    if kind == "right_semi":
        exp += [(None, int(bpay[b])) for b in np.nonzero(matched_b)[0]]
    assert sorted(got, key=repr) == sorted(exp, key=repr) and len(tail) == (int(matched_b.sum()) if kind == "right_semi" else int((~matched_b).sum()))
    assert 0 < matched_b.sum() < nb


@pytest.mark.parametrize("shape", ["clustered", "random", "dense"])
def test_inner_join_on_sliced_key_columns_and_clustered_keys(gpu, oracle, shape):
    """r03 probe: (1) a probe key column that starts 8 bytes off a 16-byte boundary (a sliced, borrowed column) takes the plain
    kernel for every tile, a 16-byte aligned one the pipelined kernel plus the plain kernel for the ragged last tile — same pairs;
    (2) keys stored in key order (runs of consecutive buckets), random keys and a dense 0..n range, against the oracle; the build
    side is large enough (> 2^19 rows) for the occupancy bitmap."""
    rng = np.random.default_rng(91)
    nb, np_ = 600_000, 1_000_003
    if shape == "clustered":     # dbgen-like sparse order keys, several probe rows per key, probe side sorted by key
        bk = ((np.arange(nb, dtype=np.uint64) // 8) * 32 + np.arange(nb, dtype=np.uint64) % 8 + 1)[rng.random(nb) < 0.4]
        pk = np.sort(rng.integers(1, int(nb // 8 * 32 + 9), np_).astype(np.uint64))
    elif shape == "random":
        bk = rng.integers(0, 1 << 62, nb).astype(np.uint64)
        pk = np.concatenate([rng.choice(bk, np_ // 3), rng.integers(0, 1 << 62, np_ - np_ // 3).astype(np.uint64)])
        rng.shuffle(pk)
    else:
        bk = rng.permutation(nb).astype(np.uint64)
        pk = rng.integers(0, 2 * nb, np_).astype(np.uint64)
    nb = len(bk)
    pvalid = rng.integers(0, 7, np_) > 0
    j = gpu.HashJoin(nb)
    j.add_block(gpu.Column.from_numpy(bk))
    j.final_build()
    cap = int(np_ * 1.2) + 16
    ep, eb = np.zeros(cap, np.uint32), np.zeros(cap, np.uint32)
    bv = np.full((nb + 7) // 8 + 8, 0xFF, np.uint8)
    pv = np.concatenate([np.packbits(pvalid, bitorder="little"), np.zeros(8, np.uint8)])
    total = oracle.orc_join_inner_u64(bk.ctypes.data_as(C.c_void_p), bv.ctypes.data_as(C.c_void_p), C.c_int64(nb), pk.ctypes.data_as(C.c_void_p),
                                      pv.ctypes.data_as(C.c_void_p), C.c_int64(np_), ep.ctypes.data_as(C.c_void_p), eb.ctypes.data_as(C.c_void_p), C.c_int64(cap))
    assert 0 < total <= cap
    gp, gb = j.probe_block(gpu.Column.from_numpy(pk, validity=pvalid))
    assert len(gp) == total and np.array_equal(gp, ep[:total]) and np.array_equal(gb, eb[:total])
    # the same keys 8 bytes into a buffer: data pointer = base + 8 (validity bits stay at offset 0)
    shifted = gpu.DeviceBuffer.from_numpy(np.concatenate([np.zeros(1, np.uint64), pk]))

    class Off:   # a borrowed view of `shifted` one element in
        ptr, nbytes = shifted.ptr + 8, pk.nbytes
    sliced = gpu.Column(T.T_U64, np_, Off, validity=gpu.DeviceBuffer.from_numpy(gpu.pack_bits(pvalid)))
    assert Off.ptr % 16 == 8
    sp, sb = j.probe_block(sliced)
    assert np.array_equal(sp, gp) and np.array_equal(sb, gb)


@pytest.mark.parametrize("n,dim", [(1, 3), (1000, 16), (3000, 100), (500, 768), (257, 1537)])
def test_vector_functions_row_by_row_over_two_columns(gpu, oracle, n, dim):
    """cosine_distance / l1_distance / l2_distance / inner_product / vector_norm over two COLUMNS (scalars/vector.rs:59-260 Array(Float32) and
    Array(Float64), :490-560 Vector(Float32 | Int8)), one side optionally a constant: within 1e-5 relative of the oracle's statement of
    distance.rs (the ndarray summation order) for f32 / i8 rows, 1e-12 for the *_64 functions."""
    rng = np.random.default_rng(n + dim)
    for elem, dt, tol in ((T.T_F32, np.float32, 1e-5), (T.T_F64, np.float64, 1e-12), (T.T_I8, np.int8, 1e-5)):
        if dt == np.int8:
            a, b = rng.integers(-128, 128, (n, dim)).astype(dt), rng.integers(-128, 128, (n, dim)).astype(dt)
        else:
            a, b = rng.standard_normal((n, dim)).astype(dt), rng.standard_normal((n, dim)).astype(dt)
        odt = np.float64 if elem == T.T_F64 else np.float32
        code = {T.T_F32: 0, T.T_F64: 1, T.T_I8: 2}[elem]
        for metric in (T.VEC_COSINE, T.VEC_L2, T.VEC_DOT, T.VEC_L1, T.VEC_NORM):
            for ls, rs in ((False, False), (False, True), (True, False)):
                if metric == T.VEC_NORM and (ls or rs):
                    continue
                la, rb = (a[0] if ls else a), (b[1 % n] if rs else b)
                got = gpu.vec_distance_rows(metric, la, None if metric == T.VEC_NORM else rb, n, dim, elem, ls, rs)
                exp = np.zeros(n, odt)
                la_c, rb_c = np.ascontiguousarray(la), np.ascontiguousarray(rb)
                oracle.orc_vec_distance_rows(metric, code, la_c.ctypes.data_as(C.c_void_p), int(ls), None if metric == T.VEC_NORM else rb_c.ctypes.data_as(C.c_void_p),
                                             int(rs), C.c_int64(n), dim, exp.ctypes.data_as(C.c_void_p))
                scale = np.maximum(1.0, np.abs(exp))
                if metric == T.VEC_DOT:      # an inner product near zero is judged against the size of its terms
                    scale = np.maximum(scale, (np.abs(np.atleast_2d(la).astype(np.float64)) * np.abs(np.atleast_2d(rb).astype(np.float64))).sum(-1))
                assert np.all(np.abs(got.astype(np.float64) - exp.astype(np.float64)) <= tol * scale + 1e-30), (elem, metric, ls, rs)
    with pytest.raises(T.DbhipError):
        gpu.vec_distance_rows(T.VEC_L2, np.zeros((2, 4), np.int32), np.zeros((2, 4), np.int32), 2, 4, T.T_I32)


@pytest.mark.parametrize("n", [1 << 20, 2_500_001])
def test_sort_perm_onesweep_sizes_match_oracle(gpu, oracle, n):
    """2^20 rows and more go through the onesweep passes (one histogram per key image, decoupled look-back between the tiles of a
    pass): 64- and 32-bit images, nullable keys (the extra pass on the NULL flag), descending keys, several keys, constant bytes
    (skipped passes), a ragged last tile — the permutation is the oracle's, tie order included."""
    rng = np.random.default_rng(n)
    cases = sort_cases(rng, n)
    valid = rng.integers(0, 5, n) > 0
    combos = [([0], [0], [0], 0), ([1], [1], [0], 0), ([2, 0], [0, 1], [0, 0], 0), ([4, 2], [1, 0], [1, 0], 0), ([6, 3, 1], [0, 1, 0], [0, 1, 0], 0)]
    for idxs, desc, nf, limit in combos:
        gcols, hcols = [], []
        for pos, i in enumerate(idxs):
            code, arr = cases[i]
            v = valid if pos == 0 and i != 5 else None
            gcols.append(gpu.Column.from_numpy(arr, code, validity=v))
            hcols.append(O.HostCol(code, arr, v))
        got = gpu.sort_perm(gcols, desc, nf, limit)
        exp = np.zeros(n, np.uint32)
        d = (C.c_uint8 * len(idxs))(*desc)
        f = (C.c_uint8 * len(idxs))(*nf)
        oracle.orc_sort_perm(O.cols(hcols), d, f, len(idxs), C.c_int64(n), C.c_int64(limit), exp.ctypes.data_as(C.c_void_p))
        assert np.array_equal(got, exp), (idxs, desc, nf)

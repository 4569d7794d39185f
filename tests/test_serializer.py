"""HashMethodSerializer (group_by_hash/method_serializer.rs + utils.rs:33-160): CPU — the oracle's restatement
against hand-laid-out bytes and the reference's method choice (tests/it/group_by.rs:30-39); GPU — dbhip_serialize_keys
byte-identical to the oracle, and the hash join on serialized keys (dbhip_join_*_binary) equal to a join on the bytes."""
import ctypes as C
import struct

import numpy as np
import pytest

from databend_amd import _lib as T
from tests import oracle_lib as O


def oracle_serialize(hcols, n):
    L = O.load()
    L.orc_serialize_keys.restype = C.c_int64
    cols = O.cols(hcols)
    off = np.zeros(n + 1, np.uint64)
    total = L.orc_serialize_keys(cols, len(hcols), C.c_int64(n), off.ctypes.data_as(C.c_void_p), None)
    assert total >= 0
    data = np.zeros(max(total, 1), np.uint8)
    assert L.orc_serialize_keys(cols, len(hcols), C.c_int64(n), off.ctypes.data_as(C.c_void_p), data.ctypes.data_as(C.c_void_p)) == total
    return off, data[:total]


def string_col(strs, validity=None):
    from databend_amd.device import make_views_general
    views, buf = make_views_general(strs)
    return O.HostCol(T.T_STRING, views, validity, buffers=[buf])


def rows_of(off, data):
    return [bytes(data[int(off[i]):int(off[i + 1])]) for i in range(len(off) - 1)]


def test_oracle_serializes_rows_like_serialize_column_binary():
    a = np.array([1, -2, 3], np.int8)
    s = [b"x1", b"a string of 27 bytes in all!", b""]
    v = np.array([True, False, True])
    d = np.array([10**20, -1, 7], dtype=object)
    cols = [O.HostCol(T.T_I8, a), string_col(s, validity=v), O.HostCol(T.T_DEC128, O.i128_array(d), None, 30, 2),
            O.HostCol(T.T_BOOL, np.packbits(np.array([1, 0, 1], bool), bitorder="little"))]
    off, data = oracle_serialize(cols, 3)
    exp = []
    for i in range(3):
        row = struct.pack("<b", a[i])
        row += bytes([int(v[i])]) + (struct.pack("<Q", len(s[i])) + s[i] if v[i] else b"")
        row += (int(d[i]) & ((1 << 128) - 1)).to_bytes(16, "little")
        row += bytes([[1, 0, 1][i]])
        exp.append(row)
    assert rows_of(off, data) == exp


def test_method_choice_matches_the_reference_test():
    """tests/it/group_by.rs:30-45: [Int8, String] -> Serializer (key width 0 here), three Int8 -> KeysU32"""
    L = T.load_library()

    def method(types):
        cols = (T.Col * len(types))()
        for i, t in enumerate(types):
            cols[i].type = t
            cols[i].precision = 30 if t == T.T_DEC128 else 0   # (a decimal's key width follows its precision)
        kb = C.c_int32(-1)
        assert L.dbhip_keys_method(cols, len(types), C.byref(kb)) == 0
        return kb.value
    assert method([T.T_I8, T.T_STRING]) == 0
    assert method([T.T_I8, T.T_I8, T.T_I8]) == 4
    assert method([T.T_DEC128, T.T_DEC128, T.T_I64]) == 0   # 40 bytes: beyond KeysU256


def random_key_columns(rng, n, D=None):
    """-> (host cols, device cols or None): Int32 nullable, String (0..40 bytes, some long, shared prefixes) nullable, Decimal128, Bool"""
    i32 = rng.integers(-5, 5, n).astype(np.int32)
    v1 = rng.integers(0, 6, n) > 0
    pool = [b"", b"k", b"Customer#000000001", b"Customer#000000002", b"Customer#0000000", b"x" * 40, b"x" * 39 + b"y", b"abcdefghijkl", b"abcdefghijklm"]
    strs = [pool[i] for i in rng.integers(0, len(pool), n)]
    v2 = rng.integers(0, 8, n) > 0
    dec = [int(x) * 10**20 for x in rng.integers(-3, 3, n)]
    bools = rng.integers(0, 2, n).astype(bool)
    host = [O.HostCol(T.T_I32, i32, v1), string_col(strs, validity=v2), O.HostCol(T.T_DEC128, O.i128_array(dec), None, 30, 2),
            O.HostCol(T.T_BOOL, np.packbits(bools, bitorder="little"))]
    dev = None
    if D is not None:
        dev = [D.Column.from_numpy(i32, validity=v1), D.Column.strings(strs, validity=v2), D.Column.decimal128(dec, 30, 2), D.Column.boolean(bools)]
    return host, dev, (v1 & v2)


@pytest.mark.gpu
@pytest.mark.parametrize("n", [0, 1, 63, 1000, 70_001])
def test_device_serialization_is_byte_identical_to_the_oracle(gpu, n):
    rng = np.random.default_rng(n + 3)
    host, dev, allv = random_key_columns(rng, n, gpu)
    off_o, data_o = oracle_serialize(host, n)
    off, data, av, total = gpu.serialize_keys(dev, n)
    assert total == len(data_o)
    assert np.array_equal(off.to_numpy(np.uint64, n + 1), off_o)
    assert np.array_equal(data.to_numpy(np.uint8, total), data_o)
    assert np.array_equal(gpu.unpack_bits(av.to_numpy(np.uint8, (n + 7) // 8), n), allv)


def expected_pairs(build_rows, probe_rows, build_ok, probe_ok):
    table = {}
    for b, r in enumerate(build_rows):
        if build_ok[b]:
            table.setdefault(r, []).append(b)
    out = []
    for p, r in enumerate(probe_rows):
        if probe_ok[p]:
            out += [(p, b) for b in table.get(r, [])]
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("mask", [None, 0xFF, 0x3])
def test_join_on_serialized_keys_equals_a_join_on_the_bytes(gpu, mask):
    """13..64-byte keys, NULLs (never match), duplicates on both sides, two build blocks; with the hash masked down to 8 / 2 bits
    thousands of different keys share a routing key and the byte-for-byte verification decides every pair"""
    D = gpu
    L = T.lib()
    L.dbhip_join_binary_debug_set_hash_mask.argtypes = [C.c_uint64]
    L.dbhip_join_binary_debug_set_hash_mask(C.c_uint64(0xFFFFFFFFFFFFFFFF if mask is None else mask))
    try:
        rng = np.random.default_rng(5)
        nb, npb = (4000, 9000) if mask is None else (300, 500)
        hb, db, vb = random_key_columns(rng, nb, D)
        hp, dp, vp = random_key_columns(rng, npb, D)
        ob, datab = oracle_serialize(hb, nb)
        op, datap = oracle_serialize(hp, npb)
        exp = expected_pairs(rows_of(ob, datab), rows_of(op, datap), vb, vp)
        j = D.BinaryHashJoin(nb)
        half = nb // 2
        slice_cols = lambda cols, lo, hi: cols   # noqa: E731 (blocks are added through fresh columns below)
        # two build blocks: rebuild the device columns of each half
        for lo, hi in ((0, half), (half, nb)):
            sub_h, sub_d, _ = None, None, None
            i32 = hb[0].arr[lo:hi]
            cols = [D.Column.from_numpy(i32, validity=np.unpackbits(hb[0].validity, bitorder="little")[lo:hi].astype(bool)),
                    D.Column.strings(rows_strings(hb[1])[lo:hi], validity=np.unpackbits(hb[1].validity, bitorder="little")[lo:hi].astype(bool)),
                    D.Column.decimal128(O.i128_list(hb[2].arr)[lo:hi], 30, 2), D.Column.boolean(np.unpackbits(hb[3].arr, bitorder="little")[lo:hi].astype(bool))]
            j.add_block(cols, hi - lo)
        j.final_build()
        pi, bi, matched = j.probe_block(dp, npb)
        got = list(zip(pi.tolist(), bi.tolist()))
        assert got == sorted(exp) and len(got) > 50
        assert np.array_equal(matched, np.isin(np.arange(npb), [p for p, _ in exp]))
    finally:
        L.dbhip_join_binary_debug_set_hash_mask(C.c_uint64(0xFFFFFFFFFFFFFFFF))


def rows_strings(hcol):
    """the strings of a host string column (views + buffer 0)"""
    views = np.ascontiguousarray(hcol.arr).view(np.uint8).reshape(-1, 16)
    buf = hcol.buffers[0]
    out = []
    for v in views:
        ln = int.from_bytes(v[0:4].tobytes(), "little")
        if ln <= 12:
            out.append(v[4:4 + ln].tobytes())
        else:
            o = int.from_bytes(v[12:16].tobytes(), "little")
            out.append(buf[o:o + ln].tobytes())
    return out

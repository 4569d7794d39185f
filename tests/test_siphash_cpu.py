"""siphash64 / scatter indices on the CPU: the oracle (oracle.c orc_siphash64 / orc_scatter_indices) and the independent Python
statement (tests/siphash_ref.py) against the reference's own golden values (tests/golden/siphash.json <- hash.txt and the
bucket_hash_v1 vectors of hash.rs), and against each other on seeded columns of every type."""
import ctypes as C
import json
import os

import numpy as np

from databend_amd import _lib as T
from databend_amd.device import i128_to_bytes, ints_to_limbs, make_views_general
from tests import oracle_lib as O
from tests import siphash_ref as R

HERE = os.path.dirname(os.path.abspath(__file__))


def orc_hash(col, n):
    out = np.zeros(max(n, 1), np.uint64)
    assert O.load().orc_siphash64(C.byref(col.c()), C.c_int64(n), out.ctypes.data_as(C.c_void_p)) == 0
    return out[:n]


def golden_column(case):
    t, v = case["type"], case.get("value")
    if t == "string":
        views, buf = make_views_general([v.encode("utf-8")])
        return O.HostCol(T.T_STRING, views, buffers=[buf])
    if t == "bytes":
        views, buf = make_views_general([bytes.fromhex(case["bytes"])])
        return O.HostCol(T.T_STRING, views, buffers=[buf])
    if t == "bool":
        return O.HostCol(T.T_BOOL, np.packbits(np.array([v], dtype=bool), bitorder="little"))
    if t == "decimal64":
        return O.HostCol(T.T_DEC64, np.array([v], np.int64), precision=case["precision"], scale=case["scale"])
    code, dt = {"timestamp": (T.T_TIMESTAMP, np.int64), "u32": (T.T_U32, np.uint32), "date": (T.T_DATE, np.int32)}[t]
    return O.HostCol(code, np.array([v], dt))


def test_the_references_golden_values():
    cases = json.load(open(os.path.join(HERE, "golden", "siphash.json"), encoding="utf-8"))
    assert len(cases) >= 16
    for c in cases:
        assert int(orc_hash(golden_column(c), 1)[0]) == c["expected"], c["what"]
        kind = c["type"]
        val = bytes.fromhex(c["bytes"]) if kind == "bytes" else c["value"]
        assert R.siphash64(kind if kind != "decimal64" else "decimal", val, c.get("scale", 0)) == c["expected"], c["what"]


def seeded_columns(n, seed):
    rng = np.random.default_rng(seed)
    strs = [bytes(rng.integers(0, 256, int(l), dtype=np.uint8)) for l in rng.integers(0, 40, n)]
    dec = [int(x) * int(y) for x, y in zip(rng.integers(-2**62, 2**62, n), rng.integers(0, 2**40, n))]
    views, buf = make_views_general(strs)
    return [
        ("i8", O.HostCol(T.T_I8, rng.integers(-128, 127, n).astype(np.int8)), None),
        ("u16", O.HostCol(T.T_U16, rng.integers(0, 65535, n).astype(np.uint16)), None),
        ("i32", O.HostCol(T.T_I32, rng.integers(-2**31, 2**31 - 1, n).astype(np.int32)), None),
        ("date", O.HostCol(T.T_DATE, rng.integers(-700000, 2900000, n).astype(np.int32)), None),
        ("u64", O.HostCol(T.T_U64, rng.integers(0, 2**64 - 1, n, dtype=np.uint64)), None),
        ("timestamp", O.HostCol(T.T_TIMESTAMP, rng.integers(-2**60, 2**60, n).astype(np.int64)), None),
        ("f32", O.HostCol(T.T_F32, rng.standard_normal(n).astype(np.float32)), None),
        ("f64", O.HostCol(T.T_F64, rng.standard_normal(n)), None),
        ("bool", O.HostCol(T.T_BOOL, np.concatenate([np.packbits(rng.integers(0, 2, n).astype(bool), bitorder="little"), np.zeros(8, np.uint8)])), None),
        ("string", O.HostCol(T.T_STRING, views, buffers=[buf]), strs),
        ("decimal64", O.HostCol(T.T_DEC64, rng.integers(-10**17, 10**17, n).astype(np.int64), precision=18, scale=3), None),
        ("decimal128", O.HostCol(T.T_DEC128, i128_to_bytes(dec), precision=38, scale=7), dec),
        ("decimal256", O.HostCol(T.T_DEC256, ints_to_limbs(dec, 256), precision=30, scale=2), dec),
    ]


def host_values(kind, col, extra, n):
    if extra is not None:
        return extra
    if kind == "bool":
        return np.unpackbits(col.arr, bitorder="little")[:n].astype(bool).tolist()
    return col.arr[:n].tolist()


def test_oracle_equals_the_python_statement_on_every_type():
    n = 300
    for kind, col, extra in seeded_columns(n, 5):
        got = orc_hash(col, n)
        vals = host_values(kind, col, extra, n)
        py_kind = "decimal" if kind.startswith("decimal") else kind
        exp = [R.siphash64(py_kind, v, col.scale) for v in vals]
        assert got.tolist() == exp, kind


def test_scatter_indices_one_key_and_several_keys():
    n = 500
    rng = np.random.default_rng(8)
    k1 = rng.integers(0, 1000, n).astype(np.int64)
    v1 = rng.integers(0, 6, n) > 0
    strs = [b"Customer#%09d" % x for x in rng.integers(0, 300, n)]
    views, buf = make_views_general(strs)
    cols1 = [O.HostCol(T.T_I64, k1, v1)]
    cols2 = [O.HostCol(T.T_I64, k1, v1), O.HostCol(T.T_STRING, views, buffers=[buf])]
    L = O.load()
    for cols, m, default in ((cols1, 8, 3), (cols1, 1, 0), (cols2, 5, 0), (cols2, 8, 0)):
        idx, cnt = np.zeros(n, np.uint32), np.zeros(m, np.uint64)
        assert L.orc_scatter_indices(O.cols(cols), len(cols), C.c_int64(n), C.c_uint64(m), C.c_uint64(default), idx.ctypes.data_as(C.c_void_p),
                                     cnt.ctypes.data_as(C.c_void_p)) == 0
        exp = []
        for i in range(n):
            hs = [R.siphash64("i64", int(k1[i])) if v1[i] else None]
            if len(cols) == 2:
                hs.append(R.siphash64("string", strs[i]))
            exp.append(R.scatter_index(hs, m, default))
        assert idx.tolist() == exp and cnt.tolist() == np.bincount(exp, minlength=m).tolist()
        if m == 8:
            assert cnt.min() > 0

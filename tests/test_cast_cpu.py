"""CPU: number -> number CAST / TRY_CAST restatement (oracle.c orc_cast_num) against the reference's cast.txt goldens and an
independent Python statement of num_traits::cast's published rules."""
import ctypes as C
import json
import math
import os

import numpy as np
import pytest

from databend_amd import _lib as T
from tests import oracle_lib as O

HERE = os.path.dirname(os.path.abspath(__file__))
NP = {"Int8": np.int8, "Int16": np.int16, "Int32": np.int32, "Int64": np.int64, "UInt8": np.uint8, "UInt16": np.uint16, "UInt32": np.uint32,
      "UInt64": np.uint64, "Float32": np.float32, "Float64": np.float64}
CODE = {"Int8": T.T_I8, "Int16": T.T_I16, "Int32": T.T_I32, "Int64": T.T_I64, "UInt8": T.T_U8, "UInt16": T.T_U16, "UInt32": T.T_U32,
        "UInt64": T.T_U64, "Float32": T.T_F32, "Float64": T.T_F64}
NAME = {v: k for k, v in CODE.items()}


def oracle_cast(L, arr, src, dst, is_try, rounding, validity=None):
    n = len(arr)
    col = O.HostCol(CODE[src], np.ascontiguousarray(arr, dtype=NP[src]), validity)
    out = np.zeros(max(n, 1), dtype=NP[dst])
    bm = np.full((n + 63) // 64 * 8 + 8, 0xFF, dtype=np.uint8)
    nerr = C.c_uint64(0)
    cc = col.c()
    assert L.orc_cast_num(C.byref(cc), CODE[dst], int(is_try), int(rounding), C.c_int64(n), out.ctypes.data_as(C.c_void_p), bm.ctypes.data_as(C.c_void_p),
                          C.byref(nerr)) == 0
    return out[:n], np.unpackbits(bm, bitorder="little")[:n].astype(bool), nerr.value


def test_cast_goldens_of_the_reference():
    L = O.load()
    g = json.load(open(os.path.join(HERE, "golden", "cast.json")))
    assert len(g["cases"]) == 10
    for c in g["cases"]:
        src = np.array([float(x) if "Float" in c["src_type"] else int(x) for x in c["src"]], dtype=NP[c["src_type"]])
        out, ok, nerr = oracle_cast(L, src, c["src_type"], c["dst_type"], c["try"], False)      # FunctionContext::default(): rounding_mode false
        exp = np.array([float(x) if "Float" in c["dst_type"] else int(x) for x in c["out"]], dtype=NP[c["dst_type"]])
        assert np.array_equal(out, exp), c["ast"]
        if c["try"]:
            assert ok.tolist() == c["validity"], c["ast"]
        else:
            assert nerr == 0 and ok.all()
    # the overflow errors name the value the function rejected
    for e in g["overflow_errors"]:
        dst = {"uint8": "UInt8", "uint16": "UInt16", "int32": "Int32", "int8": "Int8", "int16": "Int16"}[e["dst_type"]]
        v = int(e["value"])
        src = "Int64" if v < 0 else "UInt64"
        out, ok, nerr = oracle_cast(L, np.array([v], dtype=NP[src]), src, dst, False, False)
        assert nerr == 1 and not ok[0] and out[0] == 0, e


def py_cast(v, src, dst, rounding):
    """num_traits::cast (0.2.19) + register_{lossless,round,lossy}_cast, value by value -> (Some?, value)"""
    sf, df = "Float" in src, "Float" in dst
    bits = int("".join(ch for ch in dst if ch.isdigit()))
    signed = dst.startswith("Int")
    if df:
        return True, NP[dst](v)
    if not sf:
        v = int(v)
        lo, hi = (-(1 << (bits - 1)), (1 << (bits - 1)) - 1) if signed else (0, (1 << bits) - 1)
        return (lo <= v <= hi), (v if lo <= v <= hi else 0)
    x = float(v)
    fbits = int("".join(ch for ch in src if ch.isdigit()))
    if rounding:
        if math.isfinite(x):
            x = math.copysign(math.floor(abs(x) + 0.5), x) if abs(x) < 2 ** 52 else x      # f64::round: half away from zero
        fbits = 64
    if math.isnan(x):
        return False, 0
    if signed:
        mn, mx1 = -(2.0 ** (bits - 1)), 2.0 ** (bits - 1)
        some = (mn - 1.0 < x < mx1) if fbits > bits else (mn <= x < mx1)
    else:
        some = -1.0 < x < 2.0 ** bits
    return some, (int(x) if some else 0)     # int(): truncation toward zero


def interesting(src, rng):
    if "Float" in src:
        base = [0.0, -0.0, 0.5, -0.5, 1.5, 2.5, -1.5, -2.5, 0.49999997, 127.5, 127.49, 128.0, -128.5, -129.0, 255.5, 256.0, -0.99, -1.0, 32767.5, 32768.0, -32768.9,
                65535.9, 65536.0, 2147483647.0, 2147483648.0, -2147483648.0, -2147483649.0, 4294967295.0, 4294967296.0, 9.223372036854775e18,
                9.223372036854776e18, -9.223372036854776e18, -9.223372036854778e18, 1.8446744073709552e19, 1.8446744073709550e19, 3.4e38, -3.4e38, 1e300,
                float("inf"), float("-inf"), float("nan")]
        arr = np.array(base + list(rng.standard_normal(200) * 10 ** rng.integers(0, 20, 200)), dtype=np.float64)
        with np.errstate(over="ignore"):
            return arr.astype(NP[src]) if src == "Float32" else arr
    info = np.iinfo(NP[src])
    base = [0, 1, info.max, info.min, info.max - 1, info.min + 1 if info.min < 0 else 2]
    for b in (7, 8, 15, 16, 31, 32, 63):
        for d in (-1, 0, 1):
            for sgn in (1, -1):
                v = sgn * ((1 << b) + d)
                if info.min <= v <= info.max:
                    base.append(v)
    return np.array(base + list(rng.integers(info.min, info.max, 200, dtype=NP[src], endpoint=True)), dtype=NP[src])


@pytest.mark.parametrize("rounding", [False, True])
@pytest.mark.parametrize("is_try", [False, True])
def test_every_number_pair_against_the_python_statement(is_try, rounding):
    L = O.load()
    rng = np.random.default_rng(13)
    for src in NP:
        arr = interesting(src, rng)
        valid = rng.integers(0, 6, len(arr)) > 0
        for dst in NP:
            with np.errstate(all="ignore"):
                out, ok, nerr = oracle_cast(L, arr, src, dst, is_try, rounding, validity=valid)
                exp = [py_cast(v, src, dst, rounding) for v in arr.tolist()] if "Float" in src else [py_cast(int(v), src, dst, rounding) for v in arr]
                for i, (some, val) in enumerate(exp):
                    want = NP[dst](val)
                    same = (out[i] == want) or (isinstance(want, np.floating) and np.isnan(want) and np.isnan(out[i]))
                    assert same, (src, dst, arr[i], out[i], want)
                    if is_try:
                        assert ok[i] == (some and valid[i]), (src, dst, arr[i])
                    else:
                        assert ok[i] == (some or not valid[i]), (src, dst, arr[i])
                if not is_try:
                    assert nerr == sum(1 for i, (some, _) in enumerate(exp) if not some and valid[i])

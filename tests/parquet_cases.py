"""The case matrix shared by the CPU (oracle vs pyarrow) and GPU (C-ABI vs pyarrow / oracle) Parquet tests."""
import decimal

import numpy as np

from databend_amd import _lib as T


def _nulls(rng, n, frac):
    return rng.random(n) < frac


def make_cases(seed=0):
    """-> list of (name, pyarrow array, out_type, writer kwargs list)"""
    import pyarrow as pa
    rng = np.random.default_rng(seed)
    cases = []

    def add(name, arr, out_type, **kw):
        cases.append((name, arr, out_type, kw))

    n = 20_000
    # integers of every physical width, with and without nulls, low and high cardinality
    i64 = rng.integers(-2**62, 2**62, n)
    add("i64_random", pa.array(i64, pa.int64()), T.T_I64)
    add("i64_random_nulls", pa.array(i64, pa.int64(), mask=_nulls(rng, n, 0.3)), T.T_I64)
    add("i64_lowcard", pa.array(rng.integers(0, 7, n) * 1000 - 3000, pa.int64(), mask=_nulls(rng, n, 0.05)), T.T_I64)
    add("i64_runs", pa.array(np.repeat(rng.integers(0, 50, n // 100), 100), pa.int64()), T.T_I64)      # long RLE runs
    add("i64_sorted_nullruns", pa.array(np.arange(n), pa.int64(), mask=(np.arange(n) // 700) % 2 == 1), T.T_I64)
    add("i32_date", pa.array(rng.integers(8000, 11000, n).astype(np.int32), pa.int32(), mask=_nulls(rng, n, 0.1)).cast(pa.date32()), T.T_DATE)
    add("i8", pa.array(rng.integers(-128, 128, n).astype(np.int8), pa.int8(), mask=_nulls(rng, n, 0.2)), T.T_I8)
    add("u8", pa.array(rng.integers(0, 256, n).astype(np.uint8), pa.uint8()), T.T_U8)
    add("i16", pa.array(rng.integers(-2**15, 2**15, n).astype(np.int16), pa.int16(), mask=_nulls(rng, n, 0.2)), T.T_I16)
    add("u16", pa.array(rng.integers(0, 2**16, n).astype(np.uint16), pa.uint16()), T.T_U16)
    add("u32", pa.array(rng.integers(0, 2**32, n).astype(np.uint32), pa.uint32(), mask=_nulls(rng, n, 0.5)), T.T_U32)
    add("u64", pa.array(rng.integers(0, 2**63, n).astype(np.uint64) * 2 + 1, pa.uint64()), T.T_U64)
    add("ts", pa.array(rng.integers(0, 2**50, n), pa.int64(), mask=_nulls(rng, n, 0.01)).cast(pa.timestamp("us")), T.T_TIMESTAMP)
    # floats (bit patterns incl. NaN / -0.0 / inf)
    f = rng.standard_normal(n)
    f[::97] = np.nan
    f[1::97] = -0.0
    f[2::97] = np.inf
    add("f64", pa.array(f, pa.float64(), mask=_nulls(rng, n, 0.1)), T.T_F64)
    add("f32", pa.array(f.astype(np.float32), pa.float32()), T.T_F32)
    # booleans
    add("bool", pa.array(rng.random(n) < 0.3, pa.bool_(), mask=_nulls(rng, n, 0.2)), T.T_BOOL)
    add("bool_nonull", pa.array(rng.random(n) < 0.9, pa.bool_()), T.T_BOOL)
    # decimals: FIXED_LEN_BYTE_ARRAY (big endian) of several lengths, and integer-backed ones
    def dec(prec, scale, lo, hi, frac):
        ints = rng.integers(lo, hi, n)
        vals = [decimal.Decimal(int(v)).scaleb(-scale) for v in ints]
        m = _nulls(rng, n, frac)
        return pa.array([None if m[i] else vals[i] for i in range(n)], pa.decimal128(prec, scale))
    add("dec15_2_flba", dec(15, 2, -10**14, 10**14, 0.1), T.T_DEC64)
    add("dec15_2_flba_as128", dec(15, 2, -10**14, 10**14, 0.0), T.T_DEC128)
    add("dec38_6_flba", pa.array([decimal.Decimal(int(a) * 10**19 + int(b)).scaleb(-6) for a, b in
                                  zip(rng.integers(-10**17, 10**17, n), rng.integers(0, 10**18, n))], pa.decimal128(38, 6)), T.T_DEC128)
    add("dec15_2_int64", dec(15, 2, -10**14, 10**14, 0.1), T.T_DEC64, store_decimal_as_integer=True)
    add("dec9_2_int32", dec(9, 2, -10**8, 10**8, 0.1), T.T_DEC64, store_decimal_as_integer=True)
    # strings: inline (<= 12 bytes), long, empty, mixed; low cardinality (dictionary) and unique (PLAIN fallback)
    flags = np.array(["A", "R", "N"])[rng.integers(0, 3, n)]
    add("str_flag", pa.array(flags, pa.string()), T.T_STRING)
    segs = np.array(["AUTOMOBILE", "BUILDING", "FURNITURE", "MACHINERY", "HOUSEHOLD", "", "a much longer market segment name"])[rng.integers(0, 7, n)]
    add("str_segment_nulls", pa.array(segs, pa.string(), mask=_nulls(rng, n, 0.15)), T.T_STRING)
    uniq = ["comment #%d %s" % (i, "x" * int(rng.integers(0, 40))) for i in range(n)]
    add("str_unique", pa.array(uniq, pa.string(), mask=_nulls(rng, n, 0.1)), T.T_STRING)
    add("str_twelve", pa.array(["123456789012", "1234567890123", "12345678901"] * (n // 3), pa.string()), T.T_STRING)
    add("binary", pa.array([bytes(rng.integers(0, 256, int(rng.integers(0, 30))).astype(np.uint8)) for _ in range(2000)], pa.binary()), T.T_STRING)
    # edge sizes
    add("empty", pa.array([], pa.int64()), T.T_I64)
    add("one", pa.array([42], pa.int64()), T.T_I64)
    add("one_null", pa.array([None], pa.int64()), T.T_I64)
    add("all_null", pa.array([None] * 1000, pa.int32()), T.T_I32)
    add("all_null_str", pa.array([None] * 777, pa.string()), T.T_STRING)
    add("n31_33", pa.array(list(range(33)), pa.int32(), mask=np.arange(33) % 31 == 0), T.T_I32)
    add("single_value_col", pa.array([7] * 5000, pa.int64()), T.T_I64)       # dictionary of one entry: bit width 0
    return cases


# writer variants every case is written with: the two shapes of the reference's writer (parquet_rs.rs:139-158) plus the
# mixed ones another writer of the same table could have produced, and small pages (many pages per chunk, runs cut at
# page boundaries, dictionary overflow -> PLAIN fallback pages after dictionary pages)
VARIANTS = [
    dict(dictionary=True, v2=True),
    dict(dictionary=False, v2=False),
    dict(dictionary=True, v2=False),
    dict(dictionary=False, v2=True),
    dict(dictionary=True, v2=True, page_size=1024),
    dict(dictionary=False, v2=False, page_size=512),
    dict(dictionary=True, v2=True, page_size=4096, dictionary_pagesize_limit=2048),
]

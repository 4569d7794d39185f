"""GPU: every kernel behind the C-ABI against the oracle on the same seeded inputs (bit-exact)."""
import ctypes as C

import numpy as np
import pytest

from databend_amd import _lib as T
from tests import oracle_lib as O

pytestmark = pytest.mark.gpu

NUMS = [(T.T_I8, np.int8), (T.T_I16, np.int16), (T.T_I32, np.int32), (T.T_I64, np.int64), (T.T_U8, np.uint8),
        (T.T_U16, np.uint16), (T.T_U32, np.uint32), (T.T_U64, np.uint64), (T.T_F32, np.float32), (T.T_F64, np.float64)]


def rand_col(rng, code, npd, n, edge=True):
    if np.issubdtype(npd, np.integer):
        info = np.iinfo(npd)
        a = rng.integers(info.min, info.max, n, dtype=npd, endpoint=True)
        if edge and n >= 8:
            a[:4] = [info.min, info.max, 0, info.max]
            a[4:8] = [0, 1, info.min, 0]
    else:
        a = (rng.standard_normal(n) * 1000).astype(npd)
        if edge and n >= 8:
            a[:8] = [0.0, -0.0, np.nan, np.inf, -np.inf, 1.5, -2.5, 0.0]
    return a


def same_bits(x, y):
    """Bit-exact, except that any NaN equals any NaN (Rust leaves NaN sign/payload of an arithmetic
    result unspecified; x86 `subsd` and gfx950 `v_add_f64 -y` differ in the propagated sign)."""
    x, y = np.ascontiguousarray(x), np.ascontiguousarray(y)
    if x.dtype.kind == "f":
        nx, ny = np.isnan(x), np.isnan(y)
        if not np.array_equal(nx, ny):
            return False
        x, y = np.where(nx, 0, x), np.where(ny, 0, y)
    return np.array_equal(x.view(np.uint8), y.view(np.uint8))


@pytest.mark.parametrize("n", [0, 1, 7, 1000, 100_003])
def test_arith_all_type_pairs(gpu, oracle, n):
    rng = np.random.default_rng(n + 1)
    for ta, da in NUMS:
        for tb, db in NUMS:
            a, b = rand_col(rng, ta, da, n), rand_col(rng, tb, db, n)
            if n > 20:
                b[10:20] = 0  # division by zero rows
            va = rng.integers(0, 2, n).astype(bool)
            ga = gpu.Column.from_numpy(a, ta, validity=va)
            gb = gpu.Column.from_numpy(b, tb)
            ha, hb = O.HostCol(ta, a, va), O.HostCol(tb, b)
            for op in range(6):
                ot = oracle.orc_arith_result_type(op, ta, tb)
                assert ot == gpu.lib().dbhip_arith_result_type(op, ta, tb)
                npd = dict(NUMS)[ot]
                errs = gpu.RowErrors(n)
                out = gpu.arith(op, ga, gb, n, errors=errs)
                exp = np.zeros(n, dtype=npd)
                eb = np.zeros(((n + 31) // 32) * 4 + 8, np.uint8)
                ec = C.c_uint64(0)
                ca, cb = ha.c(), hb.c()
                assert oracle.orc_arith(op, C.byref(ca), C.byref(cb), C.c_int64(n), ot, exp.ctypes.data_as(C.c_void_p), eb.ctypes.data_as(C.c_void_p), C.byref(ec)) == 0
                got = out.to_numpy()
                assert same_bits(got, exp), (op, ta, tb, got[:12], exp[:12])
                assert errs.num_errors() == ec.value, (op, ta, tb)
                exp_err = np.nonzero(~np.unpackbits(eb, bitorder="little")[:n].astype(bool))[0]
                assert np.array_equal(errs.error_rows(), exp_err)


def test_scalar_operands(gpu, oracle):
    n = 1000
    rng = np.random.default_rng(3)
    a = rand_col(rng, T.T_I32, np.int32, n)
    ga, gs = gpu.Column.from_numpy(a, T.T_I32), gpu.Column.scalar(7, T.T_I64)
    out = gpu.arith(T.OP_MULTIPLY, ga, gs, n).to_numpy()
    assert np.array_equal(out, a.astype(np.int64) * 7)
    out = gpu.arith(T.OP_MODULO, ga, gpu.Column.scalar(7, T.T_U8), n).to_numpy()
    exp = np.fmod(a.astype(np.int64), 7).astype(np.int16)
    assert np.array_equal(out, exp)


@pytest.mark.parametrize("n", [1, 1000, 50_001])
def test_sum_a_plus_b_mul_c_wrapping(gpu, oracle, n):
    rng = np.random.default_rng(n)
    a, b, c = (rng.integers(-2**63, 2**63 - 1, n, dtype=np.int64) for _ in range(3))
    got = gpu.sum_a_plus_b_mul_c(*(gpu.Column.from_numpy(x) for x in (a, b, c)))
    exp = oracle.orc_sum_a_plus_b_mul_c_i64(a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p), c.ctypes.data_as(C.c_void_p), C.c_int64(n), C.c_int64(65536))
    assert got == exp
    # un-fused plan (one kernel per call node) gives the same wrapping sum
    ga, gb, gc = (gpu.Column.from_numpy(x) for x in (a, b, c))
    t = gpu.arith(T.OP_PLUS, ga, gpu.arith(T.OP_MULTIPLY, gb, gc, n), n)
    assert gpu.column_sum(t) == exp


def dec_cases():
    # (lhs (type, p, s), rhs (type, p, s), op)
    D64, D128 = T.T_DEC64, T.T_DEC128
    cs = []
    for op in (T.OP_PLUS, T.OP_MINUS, T.OP_MULTIPLY, T.OP_DIVIDE):
        cs += [((D64, 15, 2), (D64, 15, 2), op), ((D64, 10, 1), (D64, 18, 4), op), ((D64, 18, 6), (D64, 18, 9), op),
               ((D128, 31, 4), (D64, 16, 2), op), ((D128, 38, 10), (D128, 38, 10), op), ((D128, 30, 12), (D128, 20, 3), op),
               ((T.T_U8, 0, 0), (D64, 15, 2), op), ((D64, 15, 2), (T.T_I32, 0, 0), op), ((T.T_I64, 0, 0), (D128, 25, 5), op),
               ((D128, 38, 0), (D64, 18, 18), op)]
    return cs


def rand_dec(rng, spec, n, small=False):
    t, p, s = spec
    if t == T.T_DEC64:
        lim = 10 ** min(p, 18) - 1
        a = rng.integers(-lim, lim, n, dtype=np.int64, endpoint=True)
        if small:
            a = a % 100000
        a[:3] = [0, lim, -lim][:min(n, 3)] if n >= 3 else a[:3]
        return a, O.HostCol(t, a, None, p, s)
    if t == T.T_DEC128:
        lim = 10 ** p - 1
        ints = [int(rng.integers(-2**62, 2**62)) * int(rng.integers(1, 2**62)) % (lim + 1) * (1 if rng.integers(0, 2) else -1) for _ in range(n)]
        if small:
            ints = [v % 1000003 for v in ints]
        if n >= 3:
            ints[:3] = [0, lim, -lim]
        return ints, O.HostCol(t, O.i128_array(ints), None, p, s)
    npd = dict(NUMS)[t]
    a = rand_col(rng, t, npd, n)
    return a, O.HostCol(t, a)


@pytest.mark.parametrize("n", [1, 4097])
def test_decimal_arith(gpu, oracle, n):
    rng = np.random.default_rng(11)
    checked = 0
    for lhs, rhs, op in dec_cases():
        for small in (False, True):
            av, ha = rand_dec(rng, lhs, n, small)
            bv, hb = rand_dec(rng, rhs, n, small)
            if op == T.OP_DIVIDE and n > 10:
                if rhs[0] == T.T_DEC128:
                    bv[5] = 0
                    hb = O.HostCol(rhs[0], O.i128_array(bv), None, rhs[1], rhs[2])
                else:
                    bv[5] = 0
            p, s = C.c_int(), C.c_int()
            props = {T.T_I8: (3, 0), T.T_U8: (3, 0), T.T_I16: (5, 0), T.T_U16: (5, 0), T.T_I32: (10, 0), T.T_U32: (10, 0), T.T_I64: (19, 0), T.T_U64: (20, 0)}
            ap = lhs[1:] if lhs[0] in (T.T_DEC64, T.T_DEC128) else props[lhs[0]]
            bp = rhs[1:] if rhs[0] in (T.T_DEC64, T.T_DEC128) else props[rhs[0]]
            if oracle.orc_decimal_result_size(op, ap[0], ap[1], bp[0], bp[1], C.byref(p), C.byref(s)) != 0:
                continue
            ot = T.T_DEC64 if p.value <= 18 else T.T_DEC128
            exp = np.zeros(n * (2 if ot == T.T_DEC128 else 1), dtype=np.uint64)
            eb = np.zeros(((n + 31) // 32) * 4 + 8, np.uint8)
            ec = C.c_uint64(0)
            ca, cb = ha.c(), hb.c()
            rc = oracle.orc_decimal_arith(op, C.byref(ca), C.byref(cb), C.c_int64(n), ot, p.value, s.value, exp.ctypes.data_as(C.c_void_p), eb.ctypes.data_as(C.c_void_p), C.byref(ec))
            assert rc == 0
            def gcol(spec, v):
                if spec[0] == T.T_DEC128:
                    return gpu.Column.decimal128(v, spec[1], spec[2])
                return gpu.Column.from_numpy(v, spec[0], precision=spec[1], scale=spec[2])
            ga, gb = gcol(lhs, av), gcol(rhs, bv)
            assert gpu.decimal_result_size(op, ga, gb) == (p.value, s.value)
            errs = gpu.RowErrors(n)
            out = gpu.decimal_arith(op, ga, gb, n, errors=errs)
            got = out.data.to_numpy(np.uint64, exp.size)
            assert np.array_equal(got, exp), (lhs, rhs, op, small)
            assert errs.num_errors() == ec.value, (lhs, rhs, op, small, errs.num_errors(), ec.value)
            checked += 1
    assert checked >= 60


@pytest.mark.parametrize("n", [0, 1, 31, 33, 4096, 100_001])
def test_cmp_filter_take(gpu, oracle, n):
    rng = np.random.default_rng(n + 5)
    for code, npd in NUMS:
        a = rand_col(rng, code, npd, n)
        b = rand_col(rng, code, npd, n)
        if n > 16:
            b[8:16] = a[8:16]
        ga, gb = gpu.Column.from_numpy(a, code), gpu.Column.from_numpy(b, code)
        ha, hb = O.HostCol(code, a), O.HostCol(code, b)
        for op in range(6):
            out = gpu.cmp(op, ga, gb, n)
            exp = np.zeros((n + 7) // 8 + 8, np.uint8)
            ca, cb = ha.c(), hb.c()
            oracle.orc_cmp(op, C.byref(ca), C.byref(cb), C.c_int64(n), exp.ctypes.data_as(C.c_void_p))
            got = out.data.to_numpy(np.uint8, (n + 7) // 8)
            assert np.array_equal(got, exp[:(n + 7) // 8]), (code, op)
    # filter: ascending selection vector, then take on several widths
    a = rng.integers(0, 100, n).astype(np.int32)
    pred = gpu.cmp(T.CMP_LTE, gpu.Column.from_numpy(a, T.T_I32), gpu.Column.scalar(37, T.T_I32), n)
    sel, k = gpu.filter_select(pred)
    exp_sel = np.nonzero(a <= 37)[0].astype(np.uint32)
    assert k == len(exp_sel)
    assert np.array_equal(sel.to_numpy(np.uint32, k), exp_sel)
    for npd in (np.uint8, np.int16, np.int32, np.int64):
        src = rng.integers(0, 100, n).astype(npd)
        got = gpu.take(gpu.Column.from_numpy(src), sel, k).to_numpy()
        assert np.array_equal(got, src[exp_sel])
    views = rng.integers(0, 255, (n, 16)).astype(np.uint8)
    got = gpu.take(gpu.Column(T.T_STRING, n, gpu.DeviceBuffer.from_numpy(views)), sel, k).to_numpy()
    assert np.array_equal(got, views[exp_sel])
    bools = rng.integers(0, 2, n).astype(bool)
    got = gpu.take(gpu.Column.boolean(bools), sel, k).to_numpy()
    assert np.array_equal(got, bools[exp_sel])


def test_cmp_strings_and_decimal128(gpu, oracle):
    strs = [b"", b"a", b"ab", b"abc", b"abd", b"a" * 12, b"a" * 13, b"a" * 13 + b"b", b"zzzzzzzzzzzzzzzzzzzzzz", b"b"]
    rng = np.random.default_rng(0)
    A = [strs[i] for i in rng.integers(0, len(strs), 500)]
    B = [strs[i] for i in rng.integers(0, len(strs), 500)]
    ga, gb = gpu.Column.strings(A), gpu.Column.strings(B)
    for op, f in ((T.CMP_EQ, lambda x, y: x == y), (T.CMP_LT, lambda x, y: x < y), (T.CMP_GTE, lambda x, y: x >= y)):
        got = gpu.cmp(op, ga, gb, 500).to_numpy()
        assert got.tolist() == [f(x, y) for x, y in zip(A, B)]
    ints_a = [int(x) for x in rng.integers(-2**62, 2**62, 300)] + [2**100, -2**100, 0]
    ints_b = [int(x) for x in rng.integers(-2**62, 2**62, 300)] + [2**100 + 1, -2**100, -1]
    got = gpu.cmp(T.CMP_LT, gpu.Column.decimal128(ints_a, 38, 0), gpu.Column.decimal128(ints_b, 38, 0), 303).to_numpy()
    assert got.tolist() == [x < y for x, y in zip(ints_a, ints_b)]


def test_group_hash_matches_oracle(gpu, oracle):
    n = 5000
    rng = np.random.default_rng(9)
    cols_np = [(c, rand_col(rng, c, d, n)) for c, d in NUMS]
    valid = rng.integers(0, 2, n).astype(bool)
    strs = [bytes(rng.integers(97, 123, int(l)).astype(np.uint8)) for l in rng.integers(0, 30, n)]
    ints128 = [int(x) * int(y) for x, y in zip(rng.integers(-2**62, 2**62, n), rng.integers(1, 2**60, n))]
    bools = rng.integers(0, 2, n).astype(bool)
    gcols = [gpu.Column.from_numpy(a, c, validity=valid if i % 2 else None) for i, (c, a) in enumerate(cols_np)]
    hcols = [O.HostCol(c, a, valid if i % 2 else None) for i, (c, a) in enumerate(cols_np)]
    gcols += [gpu.Column.strings(strs), gpu.Column.decimal128(ints128, 38, 0), gpu.Column.boolean(bools),
              gpu.Column.from_numpy(rng.integers(-10000, 10000, n).astype(np.int32), T.T_DATE)]
    from databend_amd.device import make_views_general, pack_bits
    v, buf = make_views_general(strs)
    hcols += [O.HostCol(T.T_STRING, v, buffers=[buf]), O.HostCol(T.T_DEC128, O.i128_array(ints128)),
              O.HostCol(T.T_BOOL, pack_bits(bools)), O.HostCol(T.T_DATE, gcols[-1].to_numpy())]
    for k in (1, 2, len(gcols)):
        for start in range(0, len(gcols) - k + 1, max(1, k)):
            got = gpu.group_hash(gcols[start:start + k], n)
            exp = np.zeros(n, np.uint64)
            oracle.orc_group_hash(O.cols(hcols[start:start + k]), k, C.c_int64(n), exp.ctypes.data_as(C.c_void_p))
            assert np.array_equal(got, exp), (start, k)


def oracle_groupby(oracle, key_types, key_nullable, aggs, hkeys, hargs, n):
    kt = (C.c_int32 * len(key_types))(*key_types)
    kn = (C.c_uint8 * len(key_types))(*key_nullable)
    ad = (O.OAgg * max(len(aggs), 1))()
    for i, a in enumerate(aggs):
        ad[i].kind, ad[i].arg_type, ad[i].arg_precision, ad[i].arg_scale, ad[i].arg_nullable = a
    oracle.orc_hashagg_create.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int]
    h = C.c_void_p(oracle.orc_hashagg_create(kt, kn, len(key_types), ad, len(aggs)))
    args = (O.OCol * max(len(aggs), 1))()
    for i, a in enumerate(hargs):
        if a is not None:
            args[i] = a.c()
    assert oracle.orc_hashagg_add_block(h, O.cols(hkeys), args, C.c_int64(n)) == 0
    return h


def oracle_rows(oracle, h, key_types, aggs):
    g = oracle.orc_hashagg_num_groups(h)
    sizes = {T.T_STRING: 16, T.T_DEC128: 16, T.T_BOOL: 1, T.T_DEC256: 32}
    from databend_amd.device import ELEM_SIZE, NP_OF, limbs_to_ints, view_strings
    kb = [np.zeros(max(g, 1) * sizes.get(t, ELEM_SIZE.get(t, 8)) + 16, np.uint8) for t in key_types]
    kv = [np.zeros(max(g, 1) + 8, np.uint8) for _ in key_types]
    res_t = []
    for a in aggs:
        kind, at = a[0], a[1]
        if kind == T.AGG_COUNT: res_t.append(T.T_U64)
        elif kind == T.AGG_SUM: res_t.append(T.T_DEC256 if at == T.T_DEC256 else T.T_DEC128 if at == T.T_DEC128 else (T.T_F64 if at in (T.T_F32, T.T_F64) else (T.T_U64 if at in (T.T_U8, T.T_U16, T.T_U32, T.T_U64) else T.T_I64)))
        else: res_t.append(at)
    ab = [np.zeros(max(g, 1) * 32 + 32, np.uint8) for _ in aggs]
    kp = (C.c_void_p * len(kb))(*[b.ctypes.data for b in kb])
    kvp = (C.c_void_p * len(kv))(*[b.ctypes.data for b in kv])
    ap = (C.c_void_p * max(len(ab), 1))(*[b.ctypes.data for b in ab])
    av = [np.ones(max(g, 1) + 8, np.uint8) for _ in aggs]
    avp = (C.c_void_p * max(len(av), 1))(*[b.ctypes.data for b in av])
    oracle.orc_hashagg_result_nullable(h, kp, kvp, ap, avp, None)
    cols = []
    for t, b, v in zip(key_types, kb, kv):
        if t == T.T_STRING: vals = view_strings(b[:16 * g])
        elif t == T.T_DEC128: vals = O.i128_list(b[:16 * g])
        elif t == T.T_DEC256: vals = limbs_to_ints(b[:32 * g], 256)
        elif t == T.T_BOOL: vals = [bool(x) for x in b[:g]]
        else: vals = b[:g * ELEM_SIZE[t]].view(NP_OF[t]).tolist()
        cols.append([x if ok else None for x, ok in zip(vals, v[:g])])
    for t, b, a, v in zip(res_t, ab, aggs, av):
        if t == T.T_STRING:     # min / max over String: (u32 length, inline bytes | u64 offset into the oracle table's byte store)
            oracle.orc_hashagg_bytes.restype = C.c_void_p
            blen = C.c_int64()
            base = oracle.orc_hashagg_bytes(h, C.byref(blen))
            store = C.string_at(base, blen.value) if blen.value else b""
            vals = []
            for i in range(g):
                ln = int(b[16 * i:16 * i + 4].view(np.uint32)[0])
                off = int(b[16 * i + 8:16 * i + 16].view(np.uint64)[0])
                vals.append(bytes(b[16 * i + 4:16 * i + 4 + ln]) if ln <= 12 else store[off:off + ln])
        elif t == T.T_DEC128: vals = O.i128_list(b[:16 * g])
        elif t == T.T_DEC256: vals = limbs_to_ints(b[:32 * g], 256)
        else: vals = b[:g * ELEM_SIZE[t]].view(NP_OF[t]).tolist()
        if a[4] and a[0] != T.AGG_COUNT:  # sum / min / max over a Nullable argument: NULL for all-NULL groups
            vals = [x if ok else None for x, ok in zip(vals, v[:g])]
        cols.append(vals)
    return [tuple(c[i] for c in cols) for i in range(g)]


def norm(rows):
    return sorted(rows, key=lambda r: tuple((x is None, str(type(x)), x if x is not None else 0) for x in r))


@pytest.mark.parametrize("n,card,pbits", [(1, 1, 0), (100, 4, 0), (10_000, 4, 0), (100_000, 1000, 0), (200_000, 150_000, 0),
                                          (1, 1, 4), (8193, 40, 4), (100_000, 1000, 6), (200_000, 150_000, 10), (200_000, 150_000, 12), (5000, 17, 13), (200_000, 150_000, 13)])
def test_groupby_matches_oracle_as_sorted_sets(gpu, oracle, n, card, pbits):
    """Mirrors tests/it/aggregates/agg_hashtable.rs: several key types incl. NULLs, sum/count/min/max,
    compared as sorted row sets (assert_block_value_sort_eq). pbits > 0 forces the radix-partitioned
    pre-aggregation path (wide layout: 6 key words, 9 aggregates)."""
    rng = np.random.default_rng(n + card)
    k_i64 = rng.integers(0, card, n).astype(np.int64) - card // 2
    k_i16 = (rng.integers(0, min(card, 100), n)).astype(np.int16)
    strs = [b"s%d" % (x % 97) for x in rng.integers(0, card, n)]
    kvalid = rng.integers(0, 8, n) > 0
    k_f32 = rng.integers(0, 3, n).astype(np.float32)
    a_i32 = rng.integers(-2**31, 2**31 - 1, n).astype(np.int32)
    a_u64 = rng.integers(0, 2**64 - 1, n, dtype=np.uint64)
    a_dec64 = rng.integers(-10**15, 10**15, n).astype(np.int64)
    a_dec128 = [int(x) * 10**9 for x in rng.integers(-10**17, 10**17, n)]
    avalid = rng.integers(0, 4, n) > 0
    key_types = [T.T_I64, T.T_I16, T.T_STRING, T.T_F32]
    key_nullable = [1, 0, 0, 0]
    aggs = [(T.AGG_COUNT, 0, 0, 0, 0), (T.AGG_SUM, T.T_I32, 0, 0, 1), (T.AGG_SUM, T.T_U64, 0, 0, 0), (T.AGG_SUM, T.T_DEC64, 15, 2, 0),
            (T.AGG_SUM, T.T_DEC128, 31, 4, 0), (T.AGG_MIN, T.T_I32, 0, 0, 0), (T.AGG_MAX, T.T_U64, 0, 0, 0), (T.AGG_COUNT, T.T_I32, 0, 0, 1),
            (T.AGG_MAX, T.T_F32, 0, 0, 0)]
    gkeys = [gpu.Column.from_numpy(k_i64, validity=kvalid), gpu.Column.from_numpy(k_i16), gpu.Column.strings(strs), gpu.Column.from_numpy(k_f32)]
    gargs = [None, gpu.Column.from_numpy(a_i32, validity=avalid), gpu.Column.from_numpy(a_u64), gpu.Column.from_numpy(a_dec64, T.T_DEC64, precision=15, scale=2),
             gpu.Column.decimal128(a_dec128, 31, 4), gpu.Column.from_numpy(a_i32), gpu.Column.from_numpy(a_u64), gpu.Column.from_numpy(a_i32, validity=avalid),
             gpu.Column.from_numpy(k_f32)]
    from databend_amd.device import make_views_general
    v, buf = make_views_general(strs)
    hkeys = [O.HostCol(T.T_I64, k_i64, kvalid), O.HostCol(T.T_I16, k_i16), O.HostCol(T.T_STRING, v, buffers=[buf]), O.HostCol(T.T_F32, k_f32)]
    hargs = [None, O.HostCol(T.T_I32, a_i32, avalid), O.HostCol(T.T_U64, a_u64), O.HostCol(T.T_DEC64, a_dec64, None, 15, 2),
             O.HostCol(T.T_DEC128, O.i128_array(a_dec128), None, 31, 4), O.HostCol(T.T_I32, a_i32), O.HostCol(T.T_U64, a_u64), O.HostCol(T.T_I32, a_i32, avalid),
             O.HostCol(T.T_F32, k_f32)]
    g = gpu.GroupBy(key_types, aggs, key_nullable)
    if pbits:
        g.debug_set_partition_bits(pbits)
    g.add_block(gkeys, gargs, n)
    got = g.result()
    h = oracle_groupby(oracle, key_types, key_nullable, aggs, hkeys, hargs, n)
    exp = oracle_rows(oracle, h, key_types, aggs)
    oracle.orc_hashagg_destroy(h)
    assert g.num_groups() == len(exp)
    assert norm(got) == norm(exp)


@pytest.mark.parametrize("pbits", [0, 4])
def test_groupby_forced_hash_collisions(gpu, oracle, pbits):
    """hash_index/index.rs:385-404 tests full tag collisions; here distinct keys are forced onto the same
    probe hash so the verify+retry path runs."""
    n = 3000
    rng = np.random.default_rng(5)
    k = rng.integers(0, 40, n).astype(np.int64)
    a = rng.integers(-1000, 1000, n).astype(np.int64)
    g = gpu.GroupBy([T.T_I64], [(T.AGG_SUM, T.T_I64, 0, 0, 0), (T.AGG_COUNT, 0, 0, 0, 0)], capacity=4096)
    g.debug_set_hash_mask(0x3)
    if pbits:
        g.debug_set_partition_bits(pbits)
    g.add_block([gpu.Column.from_numpy(k)], [gpu.Column.from_numpy(a), None], n)
    g.add_block([gpu.Column.from_numpy(k)], [gpu.Column.from_numpy(a), None], n)
    got = sorted(g.result())
    exp = sorted((int(key), int(2 * a[k == key].sum()), int(2 * (k == key).sum())) for key in np.unique(k))
    assert got == exp


def test_groupby_serialized_roundtrip_and_merge(gpu):
    """flush_serialized -> merge_serialized into another table == combine_payload (aggregate_hashtable.rs:349-380)."""
    n = 50_000
    rng = np.random.default_rng(6)
    k = rng.integers(0, 500, n).astype(np.int32)
    a = rng.integers(-10**6, 10**6, n).astype(np.int64)
    spec = ([T.T_I32], [(T.AGG_SUM, T.T_I64, 0, 0, 0), (T.AGG_COUNT, 0, 0, 0, 0), (T.AGG_MIN, T.T_I64, 0, 0, 0)])
    g1, g2, gall = gpu.GroupBy(*spec), gpu.GroupBy(*spec), gpu.GroupBy(*spec)
    g1.add_block([gpu.Column.from_numpy(k[:n // 2])], [gpu.Column.from_numpy(a[:n // 2]), None, gpu.Column.from_numpy(a[:n // 2])], n // 2)
    g2.add_block([gpu.Column.from_numpy(k[n // 2:])], [gpu.Column.from_numpy(a[n // 2:]), None, gpu.Column.from_numpy(a[n // 2:])], n - n // 2)
    gall.add_block([gpu.Column.from_numpy(k)], [gpu.Column.from_numpy(a), None, gpu.Column.from_numpy(a)], n)
    g1.merge_serialized(g2.flush_serialized())
    assert sorted(g1.result()) == sorted(gall.result())
    exp = sorted((int(key), int(a[k == key].sum()), int((k == key).sum()), int(a[k == key].min())) for key in np.unique(k))
    assert sorted(gall.result()) == exp


def test_groupby_state_block_roundtrip_partial_to_final(gpu):
    """Two partial tables flush their serialized-state blocks ([state columns..., group columns...],
    payload_flush.rs:151-181: sum -> value, count -> u64) as HBM columns; a final table batch_merges both blocks
    (count states are ADDED, not counted) and equals one table over all rows — TransformPartialAggregate ->
    exchange -> TransformFinalAggregate with column blocks instead of row payloads."""
    n = 200_000
    rng = np.random.default_rng(61)
    k1 = rng.integers(0, 3000, n).astype(np.int64)
    k2 = rng.integers(0, 3, n).astype(np.int32)
    a = rng.integers(-10**9, 10**9, n).astype(np.int64)
    d = [int(x) * 10**10 for x in rng.integers(-10**15, 10**15, n)]
    f = rng.integers(-100, 100, n).astype(np.float64)
    av = rng.integers(0, 4, n) > 0
    spec = ([T.T_I64, T.T_DATE], [(T.AGG_SUM, T.T_I64, 0, 0, 0), (T.AGG_COUNT, 0, 0, 0, 0), (T.AGG_SUM, T.T_DEC128, 31, 4, 0),
                                  (T.AGG_SUM, T.T_F64, 0, 0, 0), (T.AGG_COUNT, T.T_I64, 0, 0, 1)])

    def run(lo, hi):
        g = gpu.GroupBy(*spec)
        g.add_block([gpu.Column.from_numpy(k1[lo:hi]), gpu.Column.from_numpy(k2[lo:hi], T.T_DATE)],
                    [gpu.Column.from_numpy(a[lo:hi]), None, gpu.Column.decimal128(d[lo:hi], 31, 4), gpu.Column.from_numpy(f[lo:hi]),
                     gpu.Column.from_numpy(a[lo:hi], validity=av[lo:hi])], hi - lo)
        return g

    whole = run(0, n)
    final = gpu.GroupBy(*spec)
    for lo, hi in ((0, 70_000), (70_000, n)):
        cols = run(lo, hi).result_columns()          # [keys..., states...] resident in HBM
        final.merge_state_block(cols[:2], cols[2:], cols[0].n)
    assert final.num_groups() == whole.num_groups()
    assert sorted(final.result()) == sorted(whole.result())
    exp_cnt = {}
    for x, y in zip(k1.tolist(), k2.tolist()):
        exp_cnt[(x, y)] = exp_cnt.get((x, y), 0) + 1
    assert {r[:2]: r[3] for r in final.result()} == exp_cnt


@pytest.mark.parametrize("seed", range(6))
def test_fused_expression_random_register_programs(gpu, seed):
    """dbhip_expr_eval allocates its LDS registers by liveness (results overwrite dead operands in place): random i64
    programs with register reuse, redefinition, x op x, dead stores, constants, results that ARE an input, and more live
    values than the 4-rows-per-lane register file holds (fallback to 2 rows) — against a numpy evaluation of the same
    program (wrapping i64)."""
    rng = np.random.default_rng(100 + seed)
    n = int(rng.integers(1, 70_000))
    n_in = int(rng.integers(1, 6))
    data = [rng.integers(-2**40, 2**40, n).astype(np.int64) for _ in range(n_in)]
    cols = [gpu.Column.from_numpy(d) for d in data]
    pr = gpu.ExprProgram(cols)
    regs = {}
    ins = []
    n_ops = int(rng.integers(0, 24))
    many_live = seed % 3 == 2
    for c in range(n_in):                      # load every input (some stay unused)
        r = c if many_live else int(rng.integers(0, 8))
        ins.append((T.EX_LOAD, r, c, 0, T.T_I64, 0))
        regs[r] = data[c].copy()
    with np.errstate(over="ignore"):
        for _ in range(n_ops):
            live = sorted(regs)
            kind = int(rng.integers(0, 10))
            dst = int(rng.integers(0, 8))
            if kind == 0:
                v = int(rng.integers(-1000, 1000))
                ins.append((T.EX_CONST, dst, 0, 0, T.T_I64, v & ((1 << 64) - 1)))
                regs[dst] = np.full(n, v, dtype=np.int64)
                continue
            a = int(rng.choice(live))
            b = a if kind == 1 else int(rng.choice(live))
            op = [T.EX_PLUS, T.EX_MINUS, T.EX_MULTIPLY][int(rng.integers(0, 3))]
            ins.append((op, dst, a, b, T.T_I64, 0))
            x, y = regs[a], regs[b]
            regs[dst] = x + y if op == T.EX_PLUS else (x - y if op == T.EX_MINUS else x * y)
    pr.ins = ins
    for r in regs:
        pr.types[r] = T.T_I64
    out_reg = int(rng.choice(sorted(regs)))
    got = pr.run(out_reg, n=n, want_sum=True)
    assert np.array_equal(got["values"], regs[out_reg])
    assert got["sum"] == int(regs[out_reg].astype(object).sum() % (1 << 64) if False else np.sum(regs[out_reg].view(np.uint64), dtype=np.uint64).astype(np.int64))


@pytest.mark.parametrize("card,max_rows", [(4, 256), (200, 256), (1, 1), (3000, 256)])
def test_groupby_exchange_blocks_device_resident(gpu, card, max_rows):
    """The multi-GPU exchange of bench.py --gpus N on one GPU: three "ranks" (three tables over disjoint row ranges)
    flush their blocks without a host round trip (dbhip_groupby_flush_block), the blocks are laid out the way
    all_gather_into_tensor lays them out, and every rank merges the OTHER ranks' blocks
    (dbhip_groupby_merge_blocks, skip = own rank): every rank must end with the table one rank computes over all rows.
    Runs through databend_amd.dist.exchange_partials_device with a stand-in communicator, on a torch stream handed
    to the library (the ordering the RCCL path relies on). 3000 groups overflow the 256-row block: reported before
    any table is touched, and the exchange falls back to the variable-length path."""
    import torch
    from databend_amd import dist as DX
    from databend_amd._lib import DbhipError, ERR_CAPACITY
    n, world = 90_000, 3
    rng = np.random.default_rng(card)
    k1 = rng.integers(0, card, n).astype(np.int64)
    a = rng.integers(-10**12, 10**12, n).astype(np.int64)
    d = [int(x) * 10**9 for x in rng.integers(-10**15, 10**15, n)]
    spec = ([T.T_I64], [(T.AGG_SUM, T.T_I64, 0, 0, 0), (T.AGG_COUNT, 0, 0, 0, 0), (T.AGG_SUM, T.T_DEC128, 31, 4, 0),
                        (T.AGG_MIN, T.T_I64, 0, 0, 0)])

    def run(lo, hi):
        g = gpu.GroupBy(*spec)
        g.add_block([gpu.Column.from_numpy(k1[lo:hi])],
                    [gpu.Column.from_numpy(a[lo:hi]), None, gpu.Column.decimal128(d[lo:hi], 31, 4), gpu.Column.from_numpy(a[lo:hi])], hi - lo)
        return g

    whole = sorted(run(0, n).result())
    bounds = [(r * n // world, (r + 1) * n // world) for r in range(world)]
    tables = [run(lo, hi) for lo, hi in bounds]
    dev = torch.device("cuda", 0)
    W = tables[0].row_bytes() // 8
    T.check(T.lib().dbhip_stream_sync(None))
    ts = torch.cuda.Stream()       # (torch's default stream has handle 0, which the C-ABI reads as "the library's stream")
    st = C.c_void_p(ts.cuda_stream)
    # what the collective would deliver: every rank's block, rank-major
    blocks = torch.empty((world * (max_rows + 1), W), dtype=torch.int64, device=dev)
    for r in range(world):
        tables[r].flush_block(blocks[r * (max_rows + 1)].data_ptr(), max_rows, st)
    torch.cuda.synchronize()
    heads = blocks.view(world, max_rows + 1, W)[:, 0, 0].cpu().numpy()
    groups = [len(set(k1[lo:hi].tolist())) for lo, hi in bounds]
    assert heads.tolist() == [g if g <= max_rows else -1 for g in groups]

    class FakeDist:
        def __init__(self, rank):
            self.rank, self.gathers, self.fallback = rank, 0, 0

        def get_world_size(self):
            return world

        def get_rank(self):
            return self.rank

        def all_gather_into_tensor(self, out, inp):
            assert torch.equal(inp[0], blocks.view(world, max_rows + 1, W)[self.rank, 0])
            out.copy_(blocks)   # (on torch's current stream, like the collective)
            self.gathers += 1

        def all_gather(self, outs, inp):   # variable-length fallback: counts, then padded rows
            self.fallback += 1
            per_rank = [tables_rows[r] for r in range(world)]
            if inp.numel() == 1:
                for r in range(world):
                    outs[r].fill_(per_rank[r].shape[0])
            else:
                for r in range(world):
                    outs[r].zero_()
                    outs[r][: per_rank[r].shape[0]] = torch.from_numpy(per_rank[r].view(np.int64)).to(dev)

    overflow = any(g > max_rows for g in groups)
    tables_rows = [t.flush_serialized() for t in tables] if overflow else None
    for r in range(world):
        fd = FakeDist(r)
        with torch.cuda.stream(ts):
            DX.exchange_partials_device(tables[r], fd, torch, dev, max_rows=max_rows, stream=st,
                                        capacity_error=lambda e: isinstance(e, DbhipError) and e.code == ERR_CAPACITY)
        ts.synchronize()
        assert fd.gathers == 1 and (fd.fallback > 0) == overflow
        assert sorted(tables[r].result()) == whole, r
    # the library's own stream + explicit drains (stream=None) gives the same result
    t2 = [run(lo, hi) for lo, hi in bounds]
    if not overflow:
        fd = FakeDist(1)
        DX.exchange_partials_device(t2[1], fd, torch, dev, max_rows=max_rows, stream=None,
                                    lib_sync=lambda: T.check(T.lib().dbhip_stream_sync(None)))
        assert sorted(t2[1].result()) == whole


@pytest.mark.parametrize("n", [1, 127, 128, 129, 1000, 300_007])
def test_q1_fused_and_operator_plans_match_oracle(gpu, oracle, n):
    from databend_amd import tpch
    host = tpch.gen_lineitem(n, seed=n)
    exp = O.q1_run(host, tpch.Q1_CUTOFF, threads=2, block_rows=4096)
    li = tpch.LineitemDevice(host)
    fused = tpch.q1_rows(tpch.q1_fused(li))
    assert fused == exp
    plan = tpch.q1_rows(tpch.q1_operator_at_a_time(li))
    assert plan == exp


def test_q1_wrapping_and_extreme_values(gpu, oracle):
    """Full-range i64 columns: every intermediate wraps exactly like the reference's release build."""
    from databend_amd import tpch
    n = 20_000
    rng = np.random.default_rng(1)
    host = tpch.gen_lineitem(n, seed=1)
    host["l_quantity"] = rng.integers(-2**63, 2**63 - 1, n, dtype=np.int64)     # i64 sums wrap silently
    host["l_discount"] = rng.integers(-2**63, 2**63 - 1, n, dtype=np.int64)     # 100 - d wraps
    host["l_extendedprice"] = rng.integers(-2**40, 2**40, n, dtype=np.int64)
    host["l_tax"] = rng.integers(-2**7, 2**7, n, dtype=np.int64)
    exp = O.q1_run(host, tpch.Q1_CUTOFF, threads=1)
    li = tpch.LineitemDevice(host)
    assert tpch.q1_rows(tpch.q1_fused(li)) == exp
    assert tpch.q1_rows(tpch.q1_operator_at_a_time(li)) == exp


def test_q1_decimal_sum_overflow_is_an_error(gpu, oracle):
    """DecimalSumState<true, i128> leaves [DECIMAL_MIN, DECIMAL_MAX] -> Overflow error in the reference
    (aggregate_sum.rs:203-216); the device table reports DBHIP_ERR_OVERFLOW at merge_result."""
    from databend_amd import tpch
    from databend_amd._lib import DbhipError, ERR_OVERFLOW
    n = 4096
    host = tpch.gen_lineitem(n, seed=1)
    host["l_extendedprice"][:] = 2**62
    host["l_discount"][:] = -(2**62)
    with pytest.raises(OverflowError):
        O.q1_run(host, tpch.Q1_CUTOFF, threads=1)
    li = tpch.LineitemDevice(host)
    for plan in (tpch.q1_fused, tpch.q1_operator_at_a_time):
        with pytest.raises(DbhipError) as e:
            tpch.q1_rows(plan(li))
        assert e.value.code == ERR_OVERFLOW


def test_q1_fused_capacity_fallback(gpu, oracle):
    """More than 8 distinct groups inside a workgroup: the fused kernel refuses (DBHIP_ERR_CAPACITY) and the
    operator-at-a-time plan gives the oracle's answer."""
    from databend_amd import tpch
    from databend_amd._lib import DbhipError, ERR_CAPACITY
    n = 5000
    host = tpch.gen_lineitem(n, seed=3)
    rng = np.random.default_rng(3)
    host["l_returnflag"][:, 4] = rng.integers(65, 91, n)  # 26 flags x 2 statuses
    li = tpch.LineitemDevice(host)
    with pytest.raises(DbhipError) as e:
        tpch.q1_fused(li)
    assert e.value.code == ERR_CAPACITY
    assert tpch.q1_rows(tpch.q1_operator_at_a_time(li)) == O.q1_run(host, tpch.Q1_CUTOFF)


def test_q1_finalize_avg_matches_sf_style_golden_shape(gpu, oracle):
    """avg = sum / count through the decimal divide kernel: Decimal(18,2)/UInt64 -> Decimal(24,8), round half
    away from zero (tests/sqllogictests/suites/tpch/queries.test:52-56 prints 8 fractional digits)."""
    from databend_amd import tpch
    rows = {(b"A", b"F"): dict(sum_qty=3773410700, sum_base_price=5658655440073, sum_disc_price=0, sum_charge=0, sum_disc=7390266, count=1478493)}
    out = tpch.q1_finalize(rows)
    # 37734107.00 / 1478493 = 25.52200585 ; 56586554400.73 / 1478493 = 38273.12973462 (SF1 golden row A,F)
    assert out[0][6] == 2552200585 and out[0][7] == 3827312973462


@pytest.mark.parametrize("n,card,pbits", [(1, 1, 0), (513, 3, 0), (70_000, 4, 0), (300_000, 700, 0), (300_000, 5000, 0), (400_000, 390_000, 0),
                                          (513, 3, 4), (300_000, 5000, 5), (300_000, 60_000, 8), (300_000, 60_000, 12), (1_500_000, 3000, 0), (1_500_000, 40_000, 0)])
def test_groupby_short_layout_lds_preaggregation_matches_oracle(gpu, oracle, n, card, pbits):
    """Short layouts (<= 4 key words, <= 6 aggregates) go through the workgroup-LDS partial aggregation;
    cardinalities below and far above the LDS table capacity (spill to the row path), nullable key,
    Decimal128 sum, min/max — compared with the oracle as sorted row sets."""
    rng = np.random.default_rng(n * 7 + card)
    k_i64 = rng.integers(0, card, n).astype(np.int64) * 7919 - 3
    k_date = rng.integers(0, 3, n).astype(np.int32)
    kvalid = rng.integers(0, 16, n) > 0
    a_i64 = rng.integers(-2**62, 2**62, n).astype(np.int64)
    a_dec128 = [int(x) * 10**12 for x in rng.integers(-10**17, 10**17, n)]
    a_u32 = rng.integers(0, 2**32 - 1, n, dtype=np.uint64).astype(np.uint32)
    avalid = rng.integers(0, 3, n) > 0
    key_types, key_nullable = [T.T_I64, T.T_DATE], [1, 0]
    aggs = [(T.AGG_SUM, T.T_I64, 0, 0, 0), (T.AGG_COUNT, 0, 0, 0, 0), (T.AGG_SUM, T.T_DEC128, 31, 4, 0), (T.AGG_MIN, T.T_U32, 0, 0, 0),
            (T.AGG_MAX, T.T_I64, 0, 0, 0), (T.AGG_COUNT, T.T_U32, 0, 0, 1)]
    gkeys = [gpu.Column.from_numpy(k_i64, validity=kvalid), gpu.Column.from_numpy(k_date, T.T_DATE)]
    gargs = [gpu.Column.from_numpy(a_i64), None, gpu.Column.decimal128(a_dec128, 31, 4), gpu.Column.from_numpy(a_u32), gpu.Column.from_numpy(a_i64),
             gpu.Column.from_numpy(a_u32, validity=avalid)]
    hkeys = [O.HostCol(T.T_I64, k_i64, kvalid), O.HostCol(T.T_DATE, k_date)]
    hargs = [O.HostCol(T.T_I64, a_i64), None, O.HostCol(T.T_DEC128, O.i128_array(a_dec128), None, 31, 4), O.HostCol(T.T_U32, a_u32), O.HostCol(T.T_I64, a_i64),
             O.HostCol(T.T_U32, a_u32, avalid)]
    g = gpu.GroupBy(key_types, aggs, key_nullable)
    if pbits:
        g.debug_set_partition_bits(pbits)  # force the radix-partitioned path (adaptive when n >= 1.5 M)
    g.add_block(gkeys, gargs, n)
    got = g.result()
    h = oracle_groupby(oracle, key_types, key_nullable, aggs, hkeys, hargs, n)
    exp = oracle_rows(oracle, h, key_types, aggs)
    oracle.orc_hashagg_destroy(h)
    assert g.num_groups() == len(exp)
    assert norm(got) == norm(exp)
    # a second block doubles every sum / count and leaves min / max
    g.add_block(gkeys, gargs, n)
    got2 = {r[:2]: r for r in g.result()}
    for r in exp:
        r2 = got2[r[:2]]
        assert r2[2] == ((r[2] * 2 + 2**63) % 2**64) - 2**63 and r2[3] == 2 * r[3] and r2[4] == 2 * r[4] and r2[5:7] == r[5:7] and r2[7] == 2 * r[7]


@pytest.mark.parametrize("n,card,strkey", [(65_536, 3, False), (300_001, 3, False), (300_001, 5, False), (200_000, 3, True)])
def test_groupby_few_groups_short_count_sum_layout_matches_oracle(gpu, oracle, n, card, strkey):
    """A handful of groups (8 with card = 3, more with card = 5) on a short COUNT / SUM layout, two blocks: nullable
    key, nullable count argument, f64 sum of integers (order free), short string key — equal to the oracle."""
    rng = np.random.default_rng(n + card)
    k_i64 = rng.integers(0, card, n).astype(np.int64) * 1_000_003 - 7
    k_date = rng.integers(0, 2, n).astype(np.int32)
    kvalid = rng.integers(0, 16, n) > 0
    a_i64 = rng.integers(-2**62, 2**62, n).astype(np.int64)
    a_f64 = rng.integers(-1000, 1000, n).astype(np.float64)
    a_u16 = rng.integers(0, 2**16 - 1, n).astype(np.uint16)
    avalid = rng.integers(0, 3, n) > 0
    aggs = [(T.AGG_SUM, T.T_I64, 0, 0, 0), (T.AGG_COUNT, 0, 0, 0, 0), (T.AGG_SUM, T.T_F64, 0, 0, 0), (T.AGG_COUNT, T.T_U16, 0, 0, 1)]
    if strkey:
        strs = [b"grp-%d" % (x % card) for x in rng.integers(0, 100, n)]
        key_types, key_nullable = [T.T_STRING, T.T_DATE], [0, 0]
        gkeys = [gpu.Column.strings(strs), gpu.Column.from_numpy(k_date, T.T_DATE)]
        from databend_amd.device import make_views_general
        v, buf = make_views_general(strs)
        hkeys = [O.HostCol(T.T_STRING, v, buffers=[buf]), O.HostCol(T.T_DATE, k_date)]
    else:
        key_types, key_nullable = [T.T_I64, T.T_DATE], [1, 0]
        gkeys = [gpu.Column.from_numpy(k_i64, validity=kvalid), gpu.Column.from_numpy(k_date, T.T_DATE)]
        hkeys = [O.HostCol(T.T_I64, k_i64, kvalid), O.HostCol(T.T_DATE, k_date)]
    gargs = [gpu.Column.from_numpy(a_i64), None, gpu.Column.from_numpy(a_f64), gpu.Column.from_numpy(a_u16, validity=avalid)]
    hargs = [O.HostCol(T.T_I64, a_i64), None, O.HostCol(T.T_F64, a_f64), O.HostCol(T.T_U16, a_u16, avalid)]
    g = gpu.GroupBy(key_types, aggs, key_nullable)
    g.add_block(gkeys, gargs, n)
    g.add_block(gkeys, gargs, n)  # a second block lands on existing groups
    got = g.result()
    h = oracle_groupby(oracle, key_types, key_nullable, aggs, hkeys, hargs, n)
    assert oracle.orc_hashagg_add_block(h, O.cols(hkeys), (O.OCol * 4)(*[a.c() if a is not None else O.OCol() for a in hargs]), C.c_int64(n)) == 0
    exp = oracle_rows(oracle, h, key_types, aggs)
    oracle.orc_hashagg_destroy(h)
    assert g.num_groups() == len(exp)
    assert norm(got) == norm(exp)


def test_groupby_short_layout_string_key_and_f64_sum(gpu, oracle):
    n = 120_000
    rng = np.random.default_rng(77)
    strs = [b"k%03d" % x for x in rng.integers(0, 400, n)]
    a_f32 = rng.integers(-1000, 1000, n).astype(np.float32)  # integers: the f64 sum is order independent
    g = gpu.GroupBy([T.T_STRING], [(T.AGG_SUM, T.T_F32, 0, 0, 0), (T.AGG_COUNT, 0, 0, 0, 0)])
    g.add_block([gpu.Column.strings(strs)], [gpu.Column.from_numpy(a_f32), None], n)
    exp = {}
    for s_, v in zip(strs, a_f32.tolist()):
        e = exp.setdefault(s_, [0.0, 0])
        e[0] += v
        e[1] += 1
    assert sorted(g.result()) == sorted((k, v[0], v[1]) for k, v in exp.items())


# ---------------------------------------------------------------------------------------------------------------
# fused expression evaluation (dbhip_expr_eval): one launch per Expr tree == the chain of per-node kernels, which
# are themselves checked against the oracle above
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n", [1, 63, 64, 129, 100_003])
def test_fused_expression_equals_operator_at_a_time(gpu, oracle, n):
    rng = np.random.default_rng(n)
    a64 = rng.integers(-2**63, 2**63 - 1, n, dtype=np.int64)
    b64 = rng.integers(-2**63, 2**63 - 1, n, dtype=np.int64)
    c64 = rng.integers(-2**63, 2**63 - 1, n, dtype=np.int64)
    a32 = rng.integers(-2**31, 2**31 - 1, n).astype(np.int32)
    u8 = rng.integers(0, 256, n).astype(np.uint8)
    i16 = rng.integers(-2**15, 2**15 - 1, n).astype(np.int16)
    f32 = rng.standard_normal(n).astype(np.float32)
    f64 = rng.standard_normal(n)
    if n > 10:
        f32[3], f64[4] = np.nan, np.nan
        u8[5:9] = 0
    va, vb = rng.integers(0, 4, n) > 0, rng.integers(0, 5, n) > 0
    X = gpu.ExprProgram

    # 1. sum(a + b * c), Int64 wrapping (BASELINE configs[0]) - values and the fused sum
    cols = [gpu.Column.from_numpy(a64), gpu.Column.from_numpy(b64), gpu.Column.from_numpy(c64)]
    p = X(cols)
    r = p.arith(T.EX_PLUS, p.load(0), p.arith(T.EX_MULTIPLY, p.load(1), p.load(2)))
    out = p.run(r, want_sum=True)
    exp = (a64.astype(np.uint64) + b64.astype(np.uint64) * c64.astype(np.uint64)).astype(np.int64)
    assert np.array_equal(out["values"], exp) and out["type"] == T.T_I64
    assert out["sum"] == int(exp.astype(np.uint64).sum(dtype=np.uint64).astype(np.int64))
    assert out["sum"] == gpu.sum_a_plus_b_mul_c(*cols)

    # 2. (a32 * u8) - i16 with nullable inputs: types Int32*UInt8 -> Int64, - Int16 -> Int64; validity = AND
    cols = [gpu.Column.from_numpy(a32, validity=va), gpu.Column.from_numpy(u8), gpu.Column.from_numpy(i16, validity=vb)]
    p = X(cols)
    r = p.arith(T.EX_MINUS, p.arith(T.EX_MULTIPLY, p.load(0), p.load(1)), p.load(2))
    out = p.run(r, want_sum=True)
    step = gpu.arith(T.OP_MINUS, gpu.arith(T.OP_MULTIPLY, cols[0], cols[1]), cols[2])
    assert out["type"] == step.dtype and np.array_equal(out["values"], step.to_numpy())
    assert np.array_equal(out["validity"], va & vb)
    assert out["sum"] == int(step.to_numpy()[va & vb].astype(np.int64).sum())

    # 3. f32 * f64 + i64 -> Float64; f32 (+,-) f32 in the reference's result type, rounded per node like the per-node kernels
    cols = [gpu.Column.from_numpy(f32), gpu.Column.from_numpy(f64), gpu.Column.from_numpy(a32.astype(np.int64))]
    p = X(cols)
    r = p.arith(T.EX_PLUS, p.arith(T.EX_MULTIPLY, p.load(0), p.load(1)), p.load(2))
    out = p.run(r)
    step = gpu.arith(T.OP_PLUS, gpu.arith(T.OP_MULTIPLY, cols[0], cols[1]), cols[2])
    assert out["type"] == T.T_F64 and np.array_equal(out["values"], step.to_numpy(), equal_nan=True)
    p = X(cols)
    r = p.arith(T.EX_MINUS, p.arith(T.EX_PLUS, p.load(0), p.load(0)), p.load(0))
    out = p.run(r)
    step = gpu.arith(T.OP_MINUS, gpu.arith(T.OP_PLUS, cols[0], cols[0]), cols[0])
    assert out["type"] == step.dtype and np.array_equal(out["values"], step.to_numpy(), equal_nan=True)

    # 4. a32 / u8 with zeros in u8: per-row "divided by zero" errors only on valid rows, same rows as dbhip_arith
    cols = [gpu.Column.from_numpy(a32, validity=va), gpu.Column.from_numpy(u8)]
    p = X(cols)
    r = p.arith(T.EX_DIVIDE, p.load(0), p.load(1))
    e1, e2 = gpu.RowErrors(n), gpu.RowErrors(n)
    out = p.run(r, errors=e1)
    step = gpu.arith(T.OP_DIVIDE, cols[0], cols[1], errors=e2)
    assert np.array_equal(out["values"], step.to_numpy(), equal_nan=True)
    assert e1.num_errors() == e2.num_errors() == int(((u8 == 0) & va).sum())
    assert np.array_equal(e1.error_rows(), e2.error_rows())

    # 5. predicate (i16 <= 100) AND (f64 > f64') OR NOT(u8 = 0): Boolean result as a Bitmap
    g64 = rng.standard_normal(n)
    cols = [gpu.Column.from_numpy(i16), gpu.Column.from_numpy(f64), gpu.Column.from_numpy(g64), gpu.Column.from_numpy(u8)]
    p = X(cols)
    c1 = p.cmp(T.EX_LTE, p.load(0), p.const(100, T.T_I16))
    c2 = p.cmp(T.EX_GT, p.load(1), p.load(2))
    c3 = p.logic(T.EX_NOT, p.cmp(T.EX_EQ, p.load(3), p.const(0, T.T_U8)))
    r = p.logic(T.EX_OR, p.logic(T.EX_AND, c1, c2), c3)
    out = p.run(r)
    m1 = gpu.cmp(T.CMP_LTE, cols[0], gpu.Column.scalar(100, T.T_I16), n).to_numpy()
    m2 = gpu.cmp(T.CMP_GT, cols[1], cols[2], n).to_numpy()
    m3 = gpu.cmp(T.CMP_NOTEQ, cols[3], gpu.Column.scalar(0, T.T_U8), n).to_numpy()
    assert np.array_equal(out["values"], (m1 & m2) | m3)

    # 6. CAST(i32 AS Int64) = i64 ; a cast that can overflow is refused (the checked CPU cast stays)
    cols = [gpu.Column.from_numpy(a32), gpu.Column.from_numpy(a32.astype(np.int64) + (rng.integers(0, 2, n)))]
    p = X(cols)
    r = p.cmp(T.EX_EQ, p.cast(p.load(0), T.T_I64), p.load(1))
    assert np.array_equal(p.run(r)["values"], a32.astype(np.int64) == cols[1].to_numpy())
    p = X(cols)
    with pytest.raises(Exception):
        p.run(p.cast(p.load(1), T.T_I32))


def test_fused_expression_rejects_ill_typed_programs(gpu):
    from databend_amd._lib import DbhipError
    cols = [gpu.Column.from_numpy(np.arange(10, dtype=np.int32)), gpu.Column.from_numpy(np.arange(10, dtype=np.int64))]
    p = gpu.ExprProgram(cols)
    a, b = p.load(0), p.load(1)
    with pytest.raises(DbhipError):   # comparison of different operand types: the planner must insert a CAST
        p.run(p.cmp(T.EX_LT, a, b))
    p = gpu.ExprProgram(cols)
    r = p.arith(T.EX_PLUS, p.load(0), p.load(1))
    p.ins[-1] = p.ins[-1][:4] + (T.T_I32,) + p.ins[-1][5:]   # wrong result type for Int32 + Int64
    with pytest.raises(DbhipError):
        p.run(r)


def test_div0_and_divnull(gpu):
    """div0: x / 0 = 0 without an error; divnull: x / 0 = NULL (numeric_basic_arithmetic.rs:441-457, registered on
    Float64 :524-543; other numeric arguments are cast by the planner — the kernel converts like `/` does)."""
    rng = np.random.default_rng(17)
    n = 10_007
    a = rng.integers(-1000, 1000, n).astype(np.int32)
    b = rng.integers(-3, 4, n).astype(np.int64)
    va = rng.integers(0, 5, n) > 0
    ca, cb = gpu.Column.from_numpy(a, validity=va), gpu.Column.from_numpy(b)
    e0 = gpu.RowErrors(n)
    r0 = gpu.arith(T.OP_DIV0, ca, cb, errors=e0)
    exp = np.where(b == 0, 0.0, a.astype(np.float64) / np.where(b == 0, 1, b))
    assert r0.dtype == T.T_F64 and np.array_equal(r0.to_numpy(), exp) and e0.num_errors() == 0
    e1 = gpu.RowErrors(n)
    r1 = gpu.arith(T.OP_DIVNULL, ca, cb, errors=e1)
    assert np.array_equal(r1.to_numpy(), exp)
    # rows with a zero divisor AND valid inputs become NULL (NULL inputs stay NULL through the operands' validity)
    assert np.array_equal(e1.error_rows(), np.nonzero((b == 0) & va)[0]) and e1.num_errors() == int(((b == 0) & va).sum())


@pytest.mark.parametrize("density", ["dense", "tenth", "sparse", "unordered"])
def test_take_block_equals_column_by_column_take(gpu, density):
    """dbhip_take_block (DataBlock::take over several columns with ONE selection, one launch) against numpy indexing, for the
    shapes its per-wave decision distinguishes: >= half of the rows wanted (plain gather), ~10 % (LDS window), < 1 % and an
    unordered selection (plain gather again); every element size, a Boolean column and a nullable one."""
    D = gpu
    rng = np.random.default_rng(17)
    n = 700_003
    if density == "dense":
        sel = np.nonzero(rng.random(n) < 0.97)[0]
    elif density == "tenth":
        sel = np.nonzero(rng.random(n) < 0.1)[0]
    elif density == "sparse":
        sel = np.nonzero(rng.random(n) < 0.004)[0]
    else:
        sel = rng.integers(0, n, 90_001)
    sel = sel.astype(np.uint32)
    k = len(sel)
    cols_np = [rng.integers(-2**62, 2**62, n).astype(np.int64), rng.integers(0, 2**31, n).astype(np.int32), rng.integers(0, 255, n).astype(np.uint8),
               rng.integers(0, 2**15, n).astype(np.int16), rng.random(n), rng.integers(0, 2, n).astype(bool)]
    valid = rng.integers(0, 5, n) > 0
    cols = [D.Column.from_numpy(c) for c in cols_np[:5]] + [D.Column(T.T_BOOL, n, D.DeviceBuffer.from_numpy(D.pack_bits(cols_np[5])))]
    cols[0] = D.Column.from_numpy(cols_np[0], validity=valid)
    dec = D.Column.from_numpy(np.stack([cols_np[0].view(np.uint64), cols_np[0].view(np.uint64) >> np.uint64(3)], axis=1).reshape(-1), dtype=T.T_DEC128, precision=38, scale=2)
    dec.n = n
    dsel = D.DeviceBuffer.from_numpy(sel)
    out = D.take_block(cols + [dec], dsel, k)
    for c, o in zip(cols_np[:5], out[:5]):
        assert np.array_equal(o.to_numpy()[:k], c[sel])
    assert np.array_equal(D.unpack_bits(out[0].validity.to_numpy(np.uint8), k), valid[sel])
    assert np.array_equal(D.unpack_bits(out[5].data.to_numpy(np.uint8), k), cols_np[5][sel])
    got = out[6].data.to_numpy(np.uint64, 2 * k).reshape(k, 2)
    assert np.array_equal(got[:, 0], cols_np[0].view(np.uint64)[sel]) and np.array_equal(got[:, 1], (cols_np[0].view(np.uint64) >> np.uint64(3))[sel])
    # and the single-column entry point gives the same values
    assert np.array_equal(D.take(cols[1], dsel, k).to_numpy()[:k], cols_np[1][sel])


def _np_select_tree(e, rows, data):
    """numpy statement of Selector::select: -> (true rows in the reference's order, false rows)"""
    kind = e[0]
    if kind in ("cmp", "bool"):
        m = data[e[-1]](rows)
        return rows[m], rows[~m]
    if kind == "and":
        cur, fall = rows, []
        for c in e[1]:
            t, f = _np_select_tree(c, cur, data)
            fall.append(f)
            cur = t
            if len(cur) == 0:
                break
        return cur, (np.concatenate(fall) if fall else rows[:0])
    cur, tall = rows, []
    for c in e[1]:
        t, f = _np_select_tree(c, cur, data)
        tall.append(t)
        cur = f
        if len(cur) == 0:
            break
    return (np.concatenate(tall) if tall else rows[:0]), cur


@pytest.mark.parametrize("n", [1, 63, 64, 1000, 300_007])
def test_selector_lists_match_the_reference_walk(gpu, n):
    """dbhip_select_cmp / dbhip_select_bool (the Selector's leaves) and the AND / OR walk over true and false lists
    (device.select_tree = process_and / process_or) against a numpy statement of the same walk: the TRUE list comes out in
    exactly the reference's order (ascending inside a conjunction, branch after branch for a disjunction), NULL rows never pass,
    later predicates only see the rows still undecided."""
    D = gpu
    rng = np.random.default_rng(n)
    a = rng.integers(-50, 50, n).astype(np.int64)
    b = rng.integers(-50, 50, n).astype(np.int32)
    c = rng.random(n)
    d = rng.integers(0, 2, n).astype(bool)
    va, vd = rng.integers(0, 5, n) > 0, rng.integers(0, 7, n) > 0
    ca, cb, cc = D.Column.from_numpy(a, validity=va), D.Column.from_numpy(b), D.Column.from_numpy(c)
    cd = D.Column(T.T_BOOL, n, D.DeviceBuffer.from_numpy(D.pack_bits(d)), D.DeviceBuffer.from_numpy(D.pack_bits(vd)))
    s10, s0, shalf = D.Column.scalar(10, T.T_I64), D.Column.scalar(0, T.T_I32), D.Column.scalar(0.5, T.T_F64)
    data = {"a>10": lambda r: va[r] & (a[r] > 10), "b<=0": lambda r: b[r] <= 0, "c<0.5": lambda r: c[r] < 0.5,
            "d": lambda r: vd[r] & d[r], "a!=b": lambda r: va[r] & (a[r] != b[r].astype(np.int64)), "10<a": lambda r: va[r] & (10 < a[r])}
    leaf = {"a>10": ("cmp", T.CMP_GT, ca, s10, "a>10"), "b<=0": ("cmp", T.CMP_LTE, cb, s0, "b<=0"), "c<0.5": ("cmp", T.CMP_LT, cc, shalf, "c<0.5"),
            "d": ("bool", cd, "d"), "10<a": ("cmp", T.CMP_LT, s10, ca, "10<a"),
            "a!=b": ("cmp", T.CMP_NOTEQ, ca, D.Column.from_numpy(b.astype(np.int64)), "a!=b")}
    trees = [leaf["a>10"], leaf["d"], leaf["10<a"],
             ("and", [leaf["a>10"], leaf["b<=0"]]), ("or", [leaf["a>10"], leaf["b<=0"]]),
             ("and", [leaf["c<0.5"], ("or", [leaf["a>10"], leaf["d"], leaf["b<=0"]]), leaf["a!=b"]]),
             ("or", [("and", [leaf["a>10"], leaf["c<0.5"]]), ("and", [leaf["b<=0"], leaf["d"]]), leaf["a!=b"]]),
             ("and", [leaf["a>10"], leaf["10<a"], ("and", [leaf["a>10"]])])]
    rows = np.arange(n, dtype=np.uint32)
    for e in trees:
        dev_tree = _strip(e)
        t, k = D.select_tree(dev_tree, n)
        exp, _ = _np_select_tree(e, rows, data)
        assert k == len(exp), e
        assert np.array_equal(t.to_numpy(np.uint32, k), exp), e
    # a leaf on an explicit selection, both lists
    sel = np.sort(rng.choice(n, size=max(n // 3, 1), replace=False)).astype(np.uint32)
    t, k, f = D.select_cmp(T.CMP_GT, ca, s10, D.DeviceBuffer.from_numpy(sel), len(sel), want_false=True)
    m = data["a>10"](sel)
    assert k == int(m.sum()) and np.array_equal(t.to_numpy(np.uint32, k), sel[m]) and np.array_equal(f.to_numpy(np.uint32, len(sel) - k), sel[~m])
    # shapes outside the kernel are refused, not guessed
    with pytest.raises(T.DbhipError) as ei:
        D.select_cmp(T.CMP_EQ, cb, ca)
    assert ei.value.code == T.ERR_UNSUPPORTED


def _strip(e):
    """drop the numpy key (last element) of the leaves"""
    if e[0] in ("cmp", "bool"):
        return e[:-1]
    return (e[0], [_strip(c) for c in e[1]])

"""CPU: oracle/decimal256.c (C, 640-bit integers, bit-serial division) against tests/dec256_ref.py (Python integers) on
seeded random operands — binary arithmetic with Decimal256 sides, unary minus, comparisons across DecimalSizes, decimal ->
decimal and integer -> decimal CAST — and against the i64 / i128 restatement of oracle.c where both apply."""
import ctypes as C

import numpy as np
import pytest

from databend_amd import _lib as T
from tests import dec256_cases as K
from tests import dec256_ref as R
from tests import oracle_lib as O


def host_col(kind, bits, size, values, is_scalar=False, validity=None):
    if kind == "dec":
        return O.HostCol(K.DEC_TYPE[bits], K.limbs_array(values, bits), validity, size[0], size[1], is_scalar)
    code, npd = K.INTS[kind]
    return O.HostCol(code, np.array(values, dtype=npd), validity, is_scalar=is_scalar)


def oracle_binary(case):
    L = O.load()
    n = len(case["expected"])
    a, b = host_col(*case["x"]), host_col(*case["y"])
    rp, rs = case["ret"]
    bits = R.storage_bits(rp)
    out = np.zeros(n * bits // 64, np.uint64)
    err = np.zeros((n + 31) // 32 * 4, np.uint8)
    cnt = C.c_uint64(0)
    ca, cb = a.c(), b.c()
    rc = L.orc_decimal_arith(case["op"], C.byref(ca), C.byref(cb), C.c_int64(n), K.DEC_TYPE[bits], rp, rs, out.ctypes.data_as(C.c_void_p),
                             err.ctypes.data_as(C.c_void_p), C.byref(cnt))
    assert rc == 0
    ok = np.unpackbits(err, bitorder="little")[:n].astype(bool)
    vals = K.limbs_list(out, bits)
    return [v if o else None for v, o in zip(vals, ok)], cnt.value


@pytest.mark.parametrize("seed", [11, 12, 13])
def test_binary_arithmetic_with_decimal256_sides_equals_the_python_statement(seed):
    cases = K.binary_cases(seed, 60)
    n_err = n_ok = 0
    for case in cases:
        got, cnt = oracle_binary(case)
        assert got == case["expected"], (case["op"], case["x"][:3], case["y"][:3], case["ret"], got, case["expected"])
        assert cnt == sum(e is None for e in case["expected"])
        n_err += cnt
        n_ok += len(got) - cnt
    assert n_err > 50 and n_ok > 300   # both branches are exercised


def test_i256_mul_div_special_branches():
    """the BigInt fallbacks and from_bigint's quirk (decimal.rs:1343-1404,1460-1487), constructed"""
    mx = 10 ** 76 - 1
    # multiply Decimal(76,20) x Decimal(76,20) -> Decimal(76,20): scale_mul = 20, overflow = true
    rs = R.result_size(R.OP_MULTIPLY, (76, 20), (76, 20))
    assert rs[2] == (76, 20)
    xv = [mx, 10 ** 60, -(10 ** 60), 10 ** 48, 3 * 10 ** 47, -(2 ** 200)]
    yv = [mx, 10 ** 30, 10 ** 30, 10 ** 48, -7 * 10 ** 47, 2 ** 60]
    exp = []
    for a, b in zip(xv, yv):
        try:
            exp.append(R.binary(R.OP_MULTIPLY, a, "dec", (76, 20), b, "dec", (76, 20))[0])
        except R.RowError:
            exp.append(None)
    assert exp[0] is None and exp[1] is not None   # the product overflows 256 bits; the quotient of [1] fits again
    case = dict(op=R.OP_MULTIPLY, x=("dec", 256, (76, 20), xv), y=("dec", 256, (76, 20), yv), ret=rs[2], expected=exp)
    assert oracle_binary(case)[0] == exp
    # divide Decimal(76,0) / Decimal(76,30): mul_scale = 30 + 6 - 0 = 36; a * 10^36 overflows for big a -> fallback
    rs = R.result_size(R.OP_DIVIDE, (76, 0), (76, 30))
    xv = [mx, 10 ** 70, -(10 ** 70), 12345, 0, 10 ** 75]
    yv = [10 ** 40, 3, 7 * 10 ** 35, -(10 ** 30), 5, 0]
    exp = []
    for a, b in zip(xv, yv):
        try:
            exp.append(R.binary(R.OP_DIVIDE, a, "dec", (76, 0), b, "dec", (76, 30))[0])
        except R.RowError:
            exp.append(None)
    assert exp[5] is None and exp[3] is not None
    case = dict(op=R.OP_DIVIDE, x=("dec", 256, (76, 0), xv), y=("dec", 256, (76, 30), yv), ret=rs[2], expected=exp)
    assert oracle_binary(case)[0] == exp
    assert R.from_bigint(-(1 << 255)) == -(10 ** 76 - 1) and R.from_bigint(1 << 255) is None and R.from_bigint(-(1 << 255) - 1) is None


def test_narrow_results_agree_with_the_i64_i128_restatement():
    """where T is i64 / i128 the Python statement must reproduce oracle.c's orc_decimal_arith (pinned on arithmetic.txt)"""
    rng = np.random.default_rng(5)
    L = O.load()
    checked = 0
    while checked < 120:
        a, b = K.rand_size(rng, 1, 38), K.rand_size(rng, 1, 38)
        op = int(rng.integers(0, 4))
        rs = R.result_size(op, a, b)
        if rs is None:
            continue
        n = 16
        av, bv = K.rand_values(rng, a[0], n), K.rand_values(rng, b[0], n)
        if checked % 2:
            av, bv = [v % 10 ** 9 for v in av], [(v % 10 ** 6) or 1 for v in bv]
        case = dict(op=op, x=("dec", R.storage_bits(a[0]), a, av), y=("dec", R.storage_bits(b[0]), b, bv), ret=rs[2], expected=[None] * n)
        got, _ = oracle_binary(case)
        exp = []
        for x, y in zip(av, bv):
            try:
                exp.append(R.binary(op, x, "dec", a, y, "dec", b)[0])
            except R.RowError:
                exp.append(None)
        assert got == exp, (op, a, b, rs, got, exp)
        checked += 1
    assert L is not None


def test_unary_minus_in_every_storage_class():
    L = O.load()
    for bits, p in ((64, 18), (128, 38), (256, 76)):
        vals = K.rand_values(np.random.default_rng(bits), p, 32) + [-(1 << (bits - 1))]
        col = host_col("dec", bits, (p, 2), vals)
        out = np.zeros(len(vals) * bits // 64, np.uint64)
        c = col.c()
        assert L.orc_decimal_neg(C.byref(c), C.c_int64(len(vals)), out.ctypes.data_as(C.c_void_p)) == 0
        assert K.limbs_list(out, bits) == [R.negate(v, bits) for v in vals]


def oracle_cmp3(case):
    L = O.load()
    (ab, asz, av), (bb, bsz, bv) = case["a"], case["b"]
    n = len(av)
    a, b = host_col("dec", ab, asz, av), host_col("dec", bb, bsz, bv)
    ca, cb = a.c(), b.c()
    res = {}
    for op in range(6):
        bm = np.zeros((n + 7) // 8, np.uint8)
        assert L.orc_cmp_decimal_any(op, C.byref(ca), C.byref(cb), C.c_int64(n), bm.ctypes.data_as(C.c_void_p)) == 0
        res[op] = np.unpackbits(bm, bitorder="little")[:n].astype(bool)
    return res


APPLY = {T.CMP_EQ: lambda c: c == 0, T.CMP_NOTEQ: lambda c: c != 0, T.CMP_LT: lambda c: c < 0, T.CMP_LTE: lambda c: c <= 0,
         T.CMP_GT: lambda c: c > 0, T.CMP_GTE: lambda c: c >= 0}


def test_comparisons_across_decimal_sizes_incl_256():
    eq = 0
    for case in K.cmp_cases(21, 80):
        res = oracle_cmp3(case)
        for op, f in APPLY.items():
            assert list(res[op]) == [f(c) for c in case["cmp3"]], (case["a"][:2], case["b"][:2], op)
        eq += sum(c == 0 for c in case["cmp3"])
    assert eq > 40
    # where both sides are <= 38 digits the i64 / i128 statement of oracle.c must agree
    L = O.load()
    for case in K.cmp_cases(22, 80):
        (ab, asz, av), (bb, bsz, bv) = case["a"], case["b"]
        if asz[0] > 38 or bsz[0] > 38:
            continue
        a, b = host_col("dec", ab, asz, av), host_col("dec", bb, bsz, bv)
        ca, cb = a.c(), b.c()
        n = len(av)
        for op in range(6):
            m1, m2 = np.zeros((n + 7) // 8, np.uint8), np.zeros((n + 7) // 8, np.uint8)
            assert L.orc_cmp_decimal(op, C.byref(ca), C.byref(cb), C.c_int64(n), m1.ctypes.data_as(C.c_void_p)) == 0
            assert L.orc_cmp_decimal_any(op, C.byref(ca), C.byref(cb), C.c_int64(n), m2.ctypes.data_as(C.c_void_p)) == 0
            assert np.array_equal(m1, m2)


def oracle_cast(case, is_try=False, validity=None):
    L = O.load()
    kind, bits, size, vals = case["src"]
    n = len(vals)
    src = host_col(kind, bits, size, vals, validity=validity)
    dp, ds = case["dst"]
    dbits = R.storage_bits(dp)
    out = np.zeros(n * dbits // 64, np.uint64)
    bm = np.zeros((n + 63) // 64 * 8, np.uint8)
    cnt = C.c_uint64(0)
    c = src.c()
    assert L.orc_decimal_cast(C.byref(c), dp, ds, int(is_try), int(case["rounding"]), C.c_int64(n), out.ctypes.data_as(C.c_void_p),
                              bm.ctypes.data_as(C.c_void_p), C.byref(cnt)) == 0
    ok = np.unpackbits(bm, bitorder="little")[:n].astype(bool)
    return K.limbs_list(out, dbits), ok, cnt.value


@pytest.mark.parametrize("seed", [31, 32])
def test_decimal_and_integer_casts_to_decimal_equal_the_python_statement(seed):
    n_err = n_ok = 0
    for case in K.cast_cases(seed, 150):
        vals, ok, cnt = oracle_cast(case)
        got = [v if o else None for v, o in zip(vals, ok)]
        assert got == case["expected"], (case["src"][:3], case["dst"], case["rounding"], got, case["expected"])
        assert cnt == sum(e is None for e in case["expected"])
        n_err += cnt
        n_ok += len(got) - cnt
        # try_to_decimal: errors become NULL, NULL inputs stay NULL and never count
        validity = np.arange(len(vals)) % 3 != 1
        _, tok, tcnt = oracle_cast(case, is_try=True, validity=validity)
        assert list(tok) == [e is not None and bool(v) for e, v in zip(case["expected"], validity)] and tcnt == 0
    assert n_err > 100 and n_ok > 1000


def golden_cast_cases():
    import json
    import os
    return json.load(open(os.path.join(os.path.dirname(__file__), "golden", "decimal_cast.json")))["cases"]


def test_reference_decimal_cast_goldens_through_the_oracle_and_the_python_statement():
    """decimal_to_decimal_cast.txt (tests/golden/make_golden_decimal_cast.py): CAST / TRY_CAST, rounding on / off, the three
    storage classes as the golden prints them"""
    cases = golden_cast_cases()
    assert len(cases) >= 86
    kinds = set()
    for c in cases:
        src = c["src"]
        vals = [int(v) for v in src["values"]]
        exp = [None if e is None else int(e) for e in c["expected"]]
        case = dict(src=("dec", src["kind"], (src["p"], src["s"]), vals), dst=tuple(c["dst"]), rounding=c["rounding"])
        ref = []
        for v in vals:
            try:
                ref.append(R.cast_decimal(v, src["kind"], (src["p"], src["s"]), tuple(c["dst"]), c["rounding"]))
            except R.RowError:
                ref.append(None)
        assert ref == exp, (c["sql"], ref, exp)
        got, ok, cnt = oracle_cast(case, is_try=c["is_try"])
        assert [g if o else None for g, o in zip(got, ok)] == exp, (c["sql"], got, ok, exp)
        if c["error"]:
            assert cnt == 1 and not c["is_try"]
        kinds.add((src["kind"], R.storage_bits(c["dst"][0])))
    assert {(64, 64), (64, 128), (128, 64), (128, 128), (256, 64), (256, 128)} <= kinds


# ---- the PRODUCT's 256-bit arithmetic on the host: databend_amd/csrc/dev_i256.h compiled with g++ ----------------------------
@pytest.fixture(scope="module")
def host_twin(tmp_path_factory):
    import os
    import subprocess
    exe = str(tmp_path_factory.mktemp("i256") / "i256_host_check")
    src = os.path.join(os.path.dirname(__file__), "i256_host_check.cpp")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-Wno-unknown-pragmas", "-o", exe, src])

    def run(lines):
        out = subprocess.run([exe], input="\n".join(lines) + "\n", capture_output=True, text=True, check=True).stdout.split("\n")
        return out[:len(lines)]
    return run


def hx(v):
    return "%064x" % (int(v) & ((1 << 256) - 1))


def unhx(line):
    if not line.startswith("ok "):
        return None
    v = int(line[3:], 16)
    return v - (1 << 256) if v >> 255 else v


def test_device_header_binary_rows_equal_the_python_statement_on_the_host(host_twin):
    lines, exp = [], []
    for seed in (11, 12, 13, 14):
        for case in K.binary_cases(seed, 60):
            (xk, _, xs, xv), (yk, _, ys, yv) = case["x"], case["y"]
            if case["ret"][0] <= 38:   # a wide divisor under a narrow result: T = i64 / i128, the older row function (dev_decimal.h)
                continue
            for a, b, e in zip(xv, yv, case["expected"]):
                lines.append(f"B {case['op']} {int(xk == 'dec')} {xs[0]} {xs[1]} {int(yk == 'dec')} {ys[0]} {ys[1]} {hx(a)} {hx(b)}")
                exp.append(e)
    got = [unhx(line) for line in host_twin(lines)]
    bad = [(l, g, e) for l, g, e in zip(lines, got, exp) if g != e]
    assert not bad, bad[:3]
    assert sum(e is None for e in exp) > 200 and sum(e is not None for e in exp) > 2000


def test_device_header_special_branches_on_the_host(host_twin):
    mx = 10 ** 76 - 1
    lines, exp = [], []
    for a, b in [(mx, mx), (10 ** 60, 10 ** 30), (-(10 ** 60), 10 ** 30), (10 ** 48, 10 ** 48), (3 * 10 ** 47, -7 * 10 ** 47), (-(2 ** 200), 2 ** 60),
                 (2 ** 254, 2), (-(2 ** 254), 2), (2 ** 127, 2 ** 128), (-(2 ** 127), 2 ** 128), (0, -5), (-5, 0)]:
        lines.append(f"B 2 1 76 20 1 76 20 {hx(a)} {hx(b)}")
        try:
            exp.append(R.binary(R.OP_MULTIPLY, a, "dec", (76, 20), b, "dec", (76, 20))[0])
        except R.RowError:
            exp.append(None)
    for a, b in [(mx, 10 ** 40), (10 ** 70, 3), (-(10 ** 70), 7 * 10 ** 35), (12345, -(10 ** 30)), (0, 5), (10 ** 75, 0), (mx, 1), (-mx, 1), (mx, -1),
                 (2 ** 255 - 1, 3), (-(2 ** 255), 1), (7, 2), (-7, 2), (7, -2), (-7, -2)]:
        for sizes in (((76, 0), (76, 30)), ((76, 2), (76, 76)), ((60, 10), (50, 3))):
            lines.append(f"B 3 1 {sizes[0][0]} {sizes[0][1]} 1 {sizes[1][0]} {sizes[1][1]} {hx(a)} {hx(b)}")
            try:
                exp.append(R.binary(R.OP_DIVIDE, a, "dec", sizes[0], b, "dec", sizes[1])[0])
            except R.RowError:
                exp.append(None)
    got = [unhx(line) for line in host_twin(lines)]
    bad = [(l, g, e) for l, g, e in zip(lines, got, exp) if g != e]
    assert not bad, bad[:3]
    assert None in exp and any(e is not None and abs(e) > 2 ** 200 for e in exp)


def test_device_header_comparisons_and_casts_on_the_host(host_twin):
    lines, exp = [], []
    for case in K.cmp_cases(21, 120):
        (_, asz, av), (_, bsz, bv) = case["a"], case["b"]
        for a, b, c in zip(av, bv, case["cmp3"]):
            lines.append(f"K 0 {asz[0]} {asz[1]} {bsz[0]} {bsz[1]} {hx(a)} {hx(b)}")
            exp.append(c)
    got = [int(x) for x in host_twin(lines)]
    assert got == exp
    lines, exp, wrapb = [], [], []
    for seed in (31, 32, 33):
        for case in K.cast_cases(seed, 150):
            kind, bits, size, vals = case["src"]
            for v, e in zip(vals, case["expected"]):
                if kind == "dec":
                    lines.append(f"C {bits} {size[0]} {size[1]} {case['dst'][0]} {case['dst'][1]} {int(case['rounding'])} {hx(v)}")
                else:
                    lines.append(f"C 0 0 0 {case['dst'][0]} {case['dst'][1]} {int(case['rounding'])} {hx(v)}")
                exp.append(e)
    got = [unhx(line) for line in host_twin(lines)]
    bad = [(l, g, e) for l, g, e in zip(lines, got, exp) if g != e]
    assert not bad, bad[:3]
    # the reference's own cast goldens
    lines, exp = [], []
    for c in golden_cast_cases():
        s = c["src"]
        for v, e in zip(s["values"], c["expected"]):
            lines.append(f"C {s['kind']} {s['p']} {s['s']} {c['dst'][0]} {c['dst'][1]} {int(c['rounding'])} {hx(int(v))}")
            exp.append(None if e is None else int(e))
    assert [unhx(line) for line in host_twin(lines)] == exp

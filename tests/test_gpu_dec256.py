"""GPU: the Decimal256 class and the decimal functions that span storage classes, through the C-ABI, bit-exact against the
oracle (oracle/decimal256.c) — binary arithmetic with Decimal256 sides, unary minus, comparisons across DecimalSizes, decimal
-> decimal and integer -> decimal CAST / TRY_CAST — on the seeded cases of tests/dec256_cases.py (which tests/test_dec256_cpu.py
pins on the Python statement), the constructed overflow / fallback branches, the reference's decimal_to_decimal_cast.txt
goldens, and a 1 M-row column for the grid-stride paths."""
import numpy as np
import pytest

from databend_amd import _lib as T
from tests import dec256_cases as K
from tests import dec256_ref as R
from tests.test_dec256_cpu import APPLY, golden_cast_cases, oracle_binary, oracle_cast, oracle_cmp3

pytestmark = pytest.mark.gpu


def dev_col(D, kind, bits, size, values, validity=None):
    if kind == "dec":
        return D.Column.decimal(values, size[0], size[1], validity, bits=bits)
    code, npd = K.INTS[kind]
    return D.Column.from_numpy(np.array(values, dtype=npd), code, validity)


def device_binary(D, case):
    n = len(case["expected"])
    a, b = dev_col(D, *case["x"]), dev_col(D, *case["y"])
    err = D.RowErrors(n)
    out = D.decimal_arith(case["op"], a, b, n, errors=err)
    assert (out.precision, out.scale) == tuple(case["ret"])
    bad = set(err.error_rows().tolist())
    assert err.num_errors() == len(bad)
    vals = out.to_numpy()
    return [None if i in bad else int(v) for i, v in enumerate(vals)], [int(v) for v in vals]


@pytest.mark.parametrize("seed", [11, 12, 13, 14])
def test_binary_arithmetic_with_decimal256_sides_is_bit_exact(gpu, seed):
    for case in K.binary_cases(seed, 60):
        got, raw = device_binary(gpu, case)
        exp, _ = oracle_binary(case)
        assert got == exp, (case["op"], case["x"][:3], case["y"][:3], case["ret"])
        assert all(r == 1 for r, e in zip(raw, exp) if e is None)   # error rows hold T::one()


def test_i256_overflow_and_bigint_fallback_branches(gpu):
    mx = 10 ** 76 - 1
    xv = [mx, 10 ** 60, -(10 ** 60), 10 ** 48, 3 * 10 ** 47, -(2 ** 200), 2 ** 254, -(2 ** 254), 2 ** 127, -(2 ** 127), 0, -5]
    yv = [mx, 10 ** 30, 10 ** 30, 10 ** 48, -7 * 10 ** 47, 2 ** 60, 2, 2, 2 ** 128, 2 ** 128, -5, 0]
    rs = R.result_size(R.OP_MULTIPLY, (76, 20), (76, 20))
    case = dict(op=R.OP_MULTIPLY, x=("dec", 256, (76, 20), xv), y=("dec", 256, (76, 20), yv), ret=rs[2], expected=[None] * len(xv))
    exp, _ = oracle_binary(case)
    assert device_binary(gpu, case)[0] == exp and None in exp
    xv = [mx, 10 ** 70, -(10 ** 70), 12345, 0, 10 ** 75, mx, -mx, mx, 2 ** 255 - 1, -(2 ** 255), 7, -7, 7, -7]
    yv = [10 ** 40, 3, 7 * 10 ** 35, -(10 ** 30), 5, 0, 1, 1, -1, 3, 1, 2, 2, -2, -2]
    for sizes in (((76, 0), (76, 30)), ((76, 2), (76, 76)), ((60, 10), (50, 3))):
        rs = R.result_size(R.OP_DIVIDE, *sizes)
        case = dict(op=R.OP_DIVIDE, x=("dec", 256, sizes[0], xv), y=("dec", 256, sizes[1], yv), ret=rs[2], expected=[None] * len(xv))
        exp, _ = oracle_binary(case)
        assert device_binary(gpu, case)[0] == exp, sizes


def test_null_rows_never_raise_and_scalars_broadcast(gpu):
    D = gpu
    n = 40
    vals = K.rand_values(np.random.default_rng(3), 76, n)
    validity = np.arange(n) % 4 != 0
    a = D.Column.decimal256(vals, 76, 0, validity)
    b = D.Column.scalar(10 ** 40, T.T_DEC256, 76, 0)
    err = D.RowErrors(n)
    out = D.decimal_arith(T.OP_MULTIPLY, a, b, n, errors=err)
    exp = []
    for v in vals:
        try:
            exp.append(R.binary(R.OP_MULTIPLY, v, "dec", (76, 0), 10 ** 40, "dec", (76, 0))[0])
        except R.RowError:
            exp.append(None)
    bad = set(err.error_rows().tolist())
    assert bad == {i for i, e in enumerate(exp) if e is None and validity[i]}
    got = out.to_numpy()
    assert all(int(g) == e for g, e in zip(got, exp) if e is not None)
    assert list(out.validity_numpy()) == list(validity)


def test_unary_minus_in_every_storage_class(gpu):
    D = gpu
    for bits, p in ((64, 18), (128, 38), (256, 76)):
        vals = K.rand_values(np.random.default_rng(bits), p, 1000) + [-(1 << (bits - 1))]
        out = D.decimal_neg(D.Column.decimal(vals, p, 2, bits=bits))
        assert [int(v) for v in out.to_numpy()] == [R.negate(v, bits) for v in vals]
        assert (out.precision, out.scale, out.dtype) == (p, 2, K.DEC_TYPE[bits])


def test_comparisons_with_decimal256_sides(gpu):
    D = gpu
    checked = 0
    for case in K.cmp_cases(21, 80) + K.cmp_cases(23, 40):
        (ab, asz, av), (bb, bsz, bv) = case["a"], case["b"]
        if 256 not in (ab, bb):
            ab = 256   # a legacy wide column: same values, Decimal256 storage
            case = dict(case, a=(ab, asz, av))
        ref = oracle_cmp3(case)
        a, b = D.Column.decimal(av, asz[0], asz[1], bits=ab), D.Column.decimal(bv, bsz[0], bsz[1], bits=bb)
        for op in APPLY:
            assert list(D.cmp(op, a, b).to_numpy()) == list(ref[op]), (asz, bsz, op)
        checked += 1
    assert checked == 120


@pytest.mark.parametrize("seed", [31, 32, 33])
def test_decimal_and_integer_casts_to_decimal(gpu, seed):
    D = gpu
    for case in K.cast_cases(seed, 120):
        kind, bits, size, vals = case["src"]
        n = len(vals)
        validity = np.arange(n) % 5 != 2
        for is_try in (False, True):
            src = dev_col(D, kind, bits, size or (0, 0), vals, validity)
            out, ok, cnt = D.decimal_cast(src, case["dst"][0], case["dst"][1], is_try=is_try, rounding_mode=case["rounding"])
            exp_vals, exp_ok, exp_cnt = oracle_cast(case, is_try=is_try, validity=validity)
            assert [int(v) for v in out.to_numpy()] == exp_vals, (case["src"][:3], case["dst"], is_try)
            assert list(ok) == list(exp_ok) and cnt == exp_cnt


def test_reference_decimal_cast_goldens_through_the_c_abi(gpu):
    D = gpu
    for c in golden_cast_cases():
        s = c["src"]
        vals = [int(v) for v in s["values"]]
        src = D.Column.decimal(vals, s["p"], s["s"], bits=s["kind"])
        out, ok, cnt = D.decimal_cast(src, c["dst"][0], c["dst"][1], is_try=c["is_try"], rounding_mode=c["rounding"])
        got = [int(v) if o else None for v, o in zip(out.to_numpy(), ok)]
        assert got == [None if e is None else int(e) for e in c["expected"]], c["sql"]
        assert cnt == (1 if c["error"] else 0)


def test_200k_rows_match_the_python_statement(gpu):
    """grid-stride paths: Decimal(60,4) * Decimal(50,3) -> Decimal(76,7) and the cast back to Decimal(38,2)"""
    D = gpu
    rng = np.random.default_rng(9)
    n = 200_003
    hi = rng.integers(-10 ** 17, 10 ** 17, n).astype(object)
    lo = rng.integers(0, 10 ** 17, n).astype(object)
    av = [int(h) * 10 ** 20 + int(l) for h, l in zip(hi, lo)]
    bv = [int(x) for x in rng.integers(-10 ** 9, 10 ** 9, n)]
    a, b = D.Column.decimal256(av, 60, 4), D.Column.decimal256(bv, 50, 3)
    err = D.RowErrors(n)
    out = D.decimal_arith(T.OP_MULTIPLY, a, b, n, errors=err)
    assert (out.precision, out.scale) == (76, 7) and err.num_errors() == 0
    got = out.to_numpy()
    assert all(int(g) == x * y for g, x, y in zip(got[::997], av[::997], bv[::997]))
    assert sum(int(g) for g in got) == sum(x * y for x, y in zip(av, bv))
    back, ok, cnt = D.decimal_cast(out, 38, 2, rounding_mode=True)
    exp = []
    for x, y in zip(av, bv):
        try:
            exp.append(R.cast_decimal(x * y, 256, (76, 7), (38, 2), True))
        except R.RowError:
            exp.append(None)
    g = back.to_numpy()
    assert cnt == sum(e is None for e in exp)
    assert all((e is None and not o) or (o and int(v) == e) for v, o, e in zip(g[::499], ok[::499], exp[::499]))

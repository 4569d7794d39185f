"""CPU: the oracle against EVERY reference golden vector that exists for the hot path — the fixtures of
tests/golden/ (extracted from the reference's testdata by make_golden.py / make_golden2.py) driven through the oracle
backend of tests/golden_eval.py. The same cases run through the C-ABI in tests/test_gpu_golden.py."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from databend_amd import _lib as T
from tests import golden_eval as G
from tests import oracle_lib as O

HERE = os.path.dirname(os.path.abspath(__file__))


def golden(name):
    return json.load(open(os.path.join(HERE, "golden", name)))["cases"]


class OracleBackend:
    """tests/golden_eval.py backend over liboracle.so"""

    def __init__(self):
        self.L = O.load()

    @staticmethod
    def host(v):
        return O.HostCol(v.dtype, v.arr, None, v.precision, v.scale, is_scalar=v.is_scalar)

    def arith(self, op, a, b, n):
        L = self.L
        code = L.orc_arith_result_type(op, a.dtype, b.dtype)
        assert code > 0, (op, a.dtype, b.dtype)
        out = np.zeros(max(n, 1), dtype=G.NP_OF_CODE[code])
        err = np.full(((n + 31) // 32) * 4 + 8, 0xFF, dtype=np.uint8)
        ca, cb = self.host(a).c(), self.host(b).c()
        assert L.orc_arith(op, C.byref(ca), C.byref(cb), C.c_int64(n), code, out.ctypes.data_as(C.c_void_p), err.ctypes.data_as(C.c_void_p), None) == 0
        return G.Val(code, out[:n])

    def decimal(self, op, a, b, n):
        L = self.L
        ap = (a.precision, a.scale) if a.is_decimal else G.INT_PROPS[a.dtype]
        bp = (b.precision, b.scale) if b.is_decimal else G.INT_PROPS[b.dtype]
        p, s = C.c_int(), C.c_int()
        assert L.orc_decimal_result_size(op, ap[0], ap[1], bp[0], bp[1], C.byref(p), C.byref(s)) == 0
        ot = T.T_DEC64 if p.value <= 18 else (T.T_DEC128 if p.value <= 38 else T.T_DEC256)
        out = np.zeros(max(n, 1) * {T.T_DEC64: 1, T.T_DEC128: 2, T.T_DEC256: 4}[ot], dtype=np.uint64)
        err = np.full(((n + 31) // 32) * 4 + 8, 0xFF, dtype=np.uint8)
        ca, cb = self.host(a).c(), self.host(b).c()
        assert L.orc_decimal_arith(op, C.byref(ca), C.byref(cb), C.c_int64(n), ot, p.value, s.value, out.ctypes.data_as(C.c_void_p),
                                   err.ctypes.data_as(C.c_void_p), None) == 0
        arr = out.view(np.int64) if ot == T.T_DEC64 else out
        return G.Val(ot, arr, None, p.value, s.value)

    def decimal_neg(self, a, n):
        words = {T.T_DEC64: 1, T.T_DEC128: 2, T.T_DEC256: 4}[a.dtype]
        out = np.zeros(max(n, 1) * words, dtype=np.uint64)
        ca = self.host(a).c()
        assert self.L.orc_decimal_neg(C.byref(ca), C.c_int64(n), out.ctypes.data_as(C.c_void_p)) == 0
        return G.Val(a.dtype, out.view(np.int64) if a.dtype == T.T_DEC64 else out, None, a.precision, a.scale)

    def cmp(self, op, a, b, n):
        L = self.L
        out = np.zeros((n + 7) // 8 + 8, dtype=np.uint8)
        ca, cb = self.host(a).c(), self.host(b).c()
        if a.is_decimal and b.is_decimal and (a.dtype != b.dtype or a.scale != b.scale):
            fn = L.orc_cmp_decimal_any if T.T_DEC256 in (a.dtype, b.dtype) else L.orc_cmp_decimal
            assert fn(op, C.byref(ca), C.byref(cb), C.c_int64(n), out.ctypes.data_as(C.c_void_p)) == 0
        else:
            if a.dtype != b.dtype:
                raise G.Skip("comparison of different physical types without a CAST")
            assert L.orc_cmp(op, C.byref(ca), C.byref(cb), C.c_int64(n), out.ctypes.data_as(C.c_void_p)) == 0
        return np.unpackbits(out, bitorder="little")[:n].astype(bool)


def test_every_arithmetic_golden_of_the_hot_path_functions():
    """arithmetic.txt: plus / minus / multiply / divide / div / modulo and unary minus over numbers and decimals of all
    three storage classes (Decimal(76,x) included), with literal / scalar operands, CASTs, nested calls and nullable inputs.
    What is skipped is skipped by name."""
    checked, skipped = G.run_cases(golden("arithmetic.json"), OracleBackend())
    out_of_scope = {k: v for k, v in skipped.items() if k.startswith("function")}
    assert len(checked) >= 75, (len(checked), skipped)
    # everything not checked is one of: another SQL function (pow, sqrt, cbrt, abs, factorial, bit_*: not in SURVEY §8a)
    # or a constant-folded expression
    allowed = ("function", "constant-folded expression", "decimal div / modulo")
    assert all(k.startswith(allowed) for k in skipped), skipped
    assert sum(out_of_scope.values()) == 49, out_of_scope     # pow 5, sqrt 6, cbrt 6, abs 6, factorial 2, bit_* 24


def test_every_comparison_golden():
    checked, skipped = G.run_cases(golden("comparison.json"), OracleBackend())
    assert len(checked) >= 21, (len(checked), skipped)        # all but the two constant-folded cases
    assert all(k.startswith("constant-folded") for k in skipped), skipped


# ---- aggregates: {sum,count,avg,min,max}[_group_by].txt ------------------------------------------------------------------
AGG_KIND = {"sum": T.AGG_SUM, "count": T.AGG_COUNT, "min": T.AGG_MIN, "max": T.AGG_MAX}
KIND_CODE = {"Int64": T.T_I64, "Int32": T.T_I32, "UInt64": T.T_U64, "UInt8": T.T_U8, "Float64": T.T_F64, "Decimal64": T.T_DEC64, "Decimal128": T.T_DEC128,
             "Int8": T.T_I8, "Int16": T.T_I16, "UInt16": T.T_U16, "UInt32": T.T_U32, "Float32": T.T_F32, "String": T.T_STRING}


def agg_argument(case):
    """-> (func, arg spec or None for count()/count(1)/sum(1), n) or raises G.Skip"""
    m = __import__("re").fullmatch(r"(\w+)\((.*)\)", case["ast"])
    if not m or m.group(1) not in ("sum", "count", "avg", "min", "max"):
        raise G.Skip("function " + case["ast"].split("(")[0])
    func, argtxt = m.group(1), m.group(2).strip()
    cols = {k: v for k, v in case["columns"].items() if k != "Output"}
    n = max((len(c["values"]) if "values" in c and c["kind"] != "Boolean" else c.get("n", 0)) for c in cols.values())
    if argtxt in ("", "1"):
        return func, ("lit1" if argtxt == "1" else None), n
    if argtxt == "NULL" or argtxt not in cols:
        raise G.Skip("argument " + argtxt)
    return func, cols[argtxt], n


def agg_column(spec, n):
    """numpy value array (ints scaled for decimals), validity, type code, precision, scale"""
    if spec == "lit1":
        return np.ones(n, np.uint8), None, T.T_U8, 0, 0
    kind = spec["kind"]
    if kind not in KIND_CODE:
        raise G.Skip("column kind " + kind)
    code = KIND_CODE[kind]
    if "const" in spec:
        if spec["const"] is None:
            return np.zeros(n, G.NP_OF_CODE[code]), np.zeros(n, bool), code, 0, 0
        return np.full(n, int(spec["const"]), G.NP_OF_CODE[code]), None, code, 0, 0
    vals = spec["values"]
    validity = np.array(spec["validity"][:n], bool) if "validity" in spec else None
    if kind == "String":     # min(s) / max(s) (aggregate_min_max_any.rs StringState): the values as bytes
        return [str(v).encode() for v in vals], validity, code, 0, 0
    if kind in ("Decimal64", "Decimal128"):
        scale = max((len(str(v).split(".")[1]) if "." in str(v) else 0) for v in vals)
        from decimal import Decimal
        ints = [int(Decimal(str(v)).scaleb(scale)) for v in vals]
        return ints, validity, code, (18 if kind == "Decimal64" else 38), scale
    if kind in ("Float64", "Float32"):
        return np.array([float(v) for v in vals], G.NP_OF_CODE[code]), validity, code, 0, 0
    return np.array(vals, G.NP_OF_CODE[code]), validity, code, 0, 0


def expected_agg(case):
    out = case["columns"]["Output"]
    if out["kind"] not in KIND_CODE:
        raise G.Skip("output kind " + out["kind"])
    vals = out["values"]
    validity = out.get("validity")
    return out["kind"], vals, validity


def run_agg_case_oracle(L, case):
    func, spec, n = agg_argument(case)
    kind, exp_vals, exp_valid = expected_agg(case)
    aggs = []
    hargs = []
    if func == "avg":
        plan = ["sum", "count"]
    else:
        plan = [func]
    arr = validity = None
    code = prec = scale = 0
    if spec is not None:
        arr, validity, code, prec, scale = agg_column(spec, n)
    for f in plan:
        if spec is None:
            if f != "count":
                raise G.Skip("argument-less " + f)
            aggs.append((T.AGG_COUNT, 0, 0, 0, 0))
            hargs.append(None)
            continue
        if f in ("min", "max") and code in (T.T_DEC128,):
            raise G.Skip("min/max on Decimal128")
        aggs.append((AGG_KIND[f], code, prec, scale, 1 if validity is not None else 0))
        if code == T.T_STRING:
            from databend_amd.device import make_views_general
            v, buf = make_views_general(arr)
            hargs.append(O.HostCol(T.T_STRING, v, validity, buffers=[buf]))
            continue
        data = O.i128_array(arr) if code == T.T_DEC128 else (np.array(arr, np.int64) if code == T.T_DEC64 else arr)
        hargs.append(O.HostCol(code, data, validity, prec, scale))
    groups = (np.arange(n) % 2).astype(np.uint8) if case["grouped"] else np.zeros(n, np.uint8)
    from tests.test_gpu_parity import oracle_groupby, oracle_rows
    h = oracle_groupby(L, [T.T_U8], [0], aggs, [O.HostCol(T.T_U8, groups)], hargs, n)
    rows = sorted(oracle_rows(L, h, [T.T_U8], aggs))
    L.orc_hashagg_destroy(h)
    return func, rows, kind, exp_vals, exp_valid, scale


def compare_agg(func, rows, kind, exp_vals, exp_valid, scale, ast):
    from decimal import Decimal
    ng = len(exp_vals)
    assert len(rows) == ng, (ast, rows)
    for g in range(ng):
        valid = True if exp_valid is None else exp_valid[g]
        if func == "avg":
            s, c = rows[g][1], rows[g][2]
            if not valid:
                assert s is None or c == 0, (ast, rows)
                continue
            got = (s / 10 ** scale if kind.startswith("Decimal") else s) / c
            exp = float(exp_vals[g])
            assert abs(got - exp) <= 1e-9 * max(1.0, abs(exp)), (ast, got, exp)
            continue
        got = rows[g][1]
        if not valid:
            assert got is None or func == "count", (ast, rows)
            continue
        if kind == "String":
            assert got == str(exp_vals[g]).encode(), (ast, got, exp_vals)
        elif kind.startswith("Decimal"):
            assert got == int(Decimal(str(exp_vals[g])).scaleb(scale)), (ast, got, exp_vals)
        elif kind.startswith("Float"):
            assert got == float(exp_vals[g]), (ast, got, exp_vals)
        else:
            assert got == int(exp_vals[g]), (ast, got, exp_vals)


def test_aggregate_goldens_sum_count_avg_min_max():
    """{sum,count,avg,min,max}.txt and *_group_by.txt (two groups by row parity): values and NULL results (a group or a table
    whose argument is NULL everywhere). avg is checked as sum / count (the planner's rewrite, aggregate_rewriter.rs:62-66)."""
    L = O.load()
    checked, skipped = [], {}
    for case in golden("aggregates.json"):
        try:
            res = run_agg_case_oracle(L, case)
            compare_agg(*res, case["ast"])
            checked.append((case["file"], case["ast"]))
        except G.Skip as e:
            skipped[e.args[0].split(" ")[0] + " " + e.args[0].split(" ")[-1].split("(")[0]] = skipped.get(e.args[0], 0) + 1
    assert len(checked) >= 54, (len(checked), skipped)
    for f in ("min.txt", "min_group_by.txt", "max.txt", "max_group_by.txt"):        # min(s) / max(s): the String states (round 4)
        assert (f, f.split(".")[0].split("_")[0] + "(s)") in checked


def test_oracle_min_max_over_strings_against_python():
    """aggregate_min_max_any.rs:62-110 (StringState) in the oracle: byte order then length (Rust's Ord on [u8]), NULL rows skipped, a
    group without a value is NULL; values of any length (the bytes are copied into the table); also through combine()."""
    from databend_amd.device import make_views_general
    from tests.test_gpu_parity import oracle_groupby, oracle_rows
    L = O.load()
    rng = np.random.default_rng(21)
    n = 20_000
    keys = rng.integers(0, 300, n).astype(np.int64)
    alphabet = [bytes([c]) for c in b"ab\x00\xffz"]
    vals = [b"".join(alphabet[int(x)] for x in rng.integers(0, len(alphabet), int(ln))) for ln in rng.integers(0, 30, n)]
    valid = rng.random(n) > 0.2
    valid[keys == 7] = False                                                        # a group whose argument is NULL everywhere
    aggs = [(T.AGG_MIN, T.T_STRING, 0, 0, 1), (T.AGG_MAX, T.T_STRING, 0, 0, 1), (T.AGG_COUNT, 0, 0, 0, 0)]
    halves = []
    for lo, hi in ((0, n // 2), (n // 2, n)):
        v, buf = make_views_general(vals[lo:hi])
        col = O.HostCol(T.T_STRING, v, valid[lo:hi], buffers=[buf])
        halves.append(oracle_groupby(L, [T.T_I64], [0], aggs, [O.HostCol(T.T_I64, keys[lo:hi])], [col, col, None], hi - lo))
    assert L.orc_hashagg_combine(halves[0], halves[1]) == 0
    rows = {r[0]: r[1:] for r in oracle_rows(L, halves[0], [T.T_I64], aggs)}
    exp = {}
    for k, s, ok in zip(keys.tolist(), vals, valid.tolist()):
        e = exp.setdefault(k, [None, None, 0])
        e[2] += 1
        if ok:
            e[0] = s if e[0] is None or s < e[0] else e[0]
            e[1] = s if e[1] is None or s > e[1] else e[1]
    assert rows == {k: tuple(v) for k, v in exp.items()} and rows[7][0] is None
    for h in halves:
        L.orc_hashagg_destroy(h)


def test_oracle_decimal256_keys_and_sums_against_python_integers():
    """aggregate_sum.rs:183-300 with T = i256 and group_hash.rs:593-597 in the oracle: Decimal256 group keys (32-byte rows, the byte hash)
    and sum(Decimal256) with its range check, against Python's integers; also through combine()."""
    from databend_amd.device import ints_to_limbs
    from tests.test_gpu_parity import oracle_groupby, oracle_rows
    L = O.load()
    rng = np.random.default_rng(1)
    n = 6000
    keys = [int(x) * 10**50 for x in rng.integers(-20, 20, n)]
    vals = [int(a) * 10**52 + int(b) for a, b in zip(rng.integers(-10**17, 10**17, n), rng.integers(0, 10**9, n))]
    valid = rng.random(n) > 0.2
    aggs = [(T.AGG_SUM, T.T_DEC256, 76, 4, 1), (T.AGG_COUNT, 0, 0, 0, 0)]
    hs = []
    for lo, hi in ((0, n // 2), (n // 2, n)):
        hk = O.HostCol(T.T_DEC256, ints_to_limbs(keys[lo:hi], 256), None, 76, 0)
        ha = O.HostCol(T.T_DEC256, ints_to_limbs(vals[lo:hi], 256), valid[lo:hi], 76, 4)
        hs.append(oracle_groupby(L, [T.T_DEC256], [0], aggs, [hk], [ha, None], hi - lo))
    assert L.orc_hashagg_combine(hs[0], hs[1]) == 0
    rows = oracle_rows(L, hs[0], [T.T_DEC256], aggs)
    exp = {}
    for k, v, ok in zip(keys, vals, valid.tolist()):
        e = exp.setdefault(k, [None, 0])
        e[1] += 1
        if ok:
            e[0] = (e[0] or 0) + v
    assert {r[0]: [r[1], r[2]] for r in rows} == exp
    for h in hs:
        L.orc_hashagg_destroy(h)
    # the range check: +-(10^76 - 1)
    mx = 10**76 - 1
    import ctypes as C
    h = oracle_groupby(L, [T.T_I64], [0], [(T.AGG_SUM, T.T_DEC256, 76, 0, 0)], [O.HostCol(T.T_I64, np.zeros(2, np.int64))],
                       [O.HostCol(T.T_DEC256, ints_to_limbs([mx, -mx], 256), None, 76, 0)], 2)
    assert oracle_rows(L, h, [T.T_I64], [(T.AGG_SUM, T.T_DEC256, 76, 0, 0)]) == [(0, 0)]
    L.orc_hashagg_destroy(h)
    kt, kn = (C.c_int32 * 1)(T.T_I64), (C.c_uint8 * 1)(0)
    ad = (O.OAgg * 1)()
    ad[0].kind, ad[0].arg_type, ad[0].arg_precision, ad[0].arg_scale, ad[0].arg_nullable = T.AGG_SUM, T.T_DEC256, 76, 0, 0
    L.orc_hashagg_create.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int]
    h = C.c_void_p(L.orc_hashagg_create(kt, kn, 1, ad, 1))
    args = (O.OCol * 1)()
    col = O.HostCol(T.T_DEC256, ints_to_limbs([mx, 1], 256), None, 76, 0)
    args[0] = col.c()
    assert L.orc_hashagg_add_block(h, O.cols([O.HostCol(T.T_I64, np.zeros(2, np.int64))]), args, C.c_int64(2)) == 5     # Overflow
    L.orc_hashagg_destroy(h)


def test_kernel_pass_filter_and_take_goldens():
    """kernel-pass.txt Filter / Take sections: Bitmap -> selection -> take of every column, rendered like the reference."""
    L = O.load()
    cases = [c for c in golden("kernel.json") if c["kind"] in ("filter", "take")]
    assert len(cases) == 2
    for case in cases:
        src = case["source"]
        n = len(src)
        if case["kind"] == "filter":
            bm = np.packbits(np.array(case["arg"], bool), bitorder="little")
            bm = np.concatenate([bm, np.zeros(8, np.uint8)])
            sel = np.zeros(n, np.uint32)
            k = L.orc_filter_select(bm.ctypes.data_as(C.c_void_p), C.c_int64(0), C.c_int64(n), sel.ctypes.data_as(C.c_void_p))
            sel = sel[:k]
        else:
            sel = np.array(case["arg"], np.uint32)
        for c in range(len(case["header"])):
            cells = [r[c] for r in src]
            vals = np.array([0 if x == "NULL" else (int(x) if x.lstrip("-").isdigit() else abs(hash(x)) % 1000) for x in cells], np.int64)
            out = np.zeros(len(sel), np.int64)
            L.orc_take(vals.ctypes.data_as(C.c_void_p), 8, sel.ctypes.data_as(C.c_void_p), C.c_int64(len(sel)), out.ctypes.data_as(C.c_void_p))
            got = [cells[i] for i in sel]          # the rendered cell of a taken row is the source row's cell
            assert out.tolist() == [int(vals[i]) for i in sel]
            assert got == [r[c] for r in case["result"]], (case["kind"], c)


def chunk_pairs(case):
    """(block, row) pairs of a 'Take Block' case: indices as they are, slices (block, start, len) expanded and truncated to
    the limit (take_by_slices_limit_from_blocks, take_chunks.rs:126-190)"""
    if case["kind"] == "chunks":
        return [tuple(t) for t in case["arg"]]
    pairs = [(b, s + i) for b, s, ln in case["arg"] for i in range(ln)]
    return pairs[:case["limit"]] if case["limit"] else pairs


def test_kernel_pass_take_block_goldens():
    """kernel-pass.txt 'Take Block indices' / 'Take Block by slices': rows of several blocks by (block, row) pairs."""
    L = O.load()
    cases = [c for c in golden("kernel.json") if c["kind"] in ("chunks", "slices")]
    assert len(cases) == 3
    for case in cases:
        pairs = np.array(chunk_pairs(case), np.uint32)
        n = len(pairs)
        for c in range(2):  # Column 0 (Int), Column 1 (nullable Int); the Array / Tuple columns are outside the path's types
            blocks = [np.array([0 if r[c] == "NULL" else int(r[c]) for r in b], np.int64) for b in case["blocks"]]
            cells = [[r[c] for r in b] for b in case["blocks"]]
            ptrs = (C.c_void_p * len(blocks))(*[b.ctypes.data for b in blocks])
            out = np.zeros(n, np.int64)
            L.orc_take_chunks(ptrs, 8, pairs.ctypes.data_as(C.c_void_p), C.c_int64(n), out.ctypes.data_as(C.c_void_p))
            assert out.tolist() == [int(blocks[b][r]) for b, r in pairs.tolist()]
            assert [cells[b][r] for b, r in pairs.tolist()] == [r[c] for r in case["result"]], (case["kind"], c)
    # take_ranges / take_compacted_indices have no golden file: closed form
    rng = np.array([[2, 5], [0, 1], [7, 7], [3, 9]], np.uint32)
    out = np.zeros(16, np.uint32)
    L.orc_sel_from_ranges.restype = C.c_int64
    L.orc_sel_from_repeats.restype = C.c_int64
    k = L.orc_sel_from_ranges(rng.ctypes.data_as(C.c_void_p), 4, out.ctypes.data_as(C.c_void_p))
    assert out[:k].tolist() == [2, 3, 4, 0, 3, 4, 5, 6, 7, 8]
    rep = np.array([[5, 2], [1, 0], [9, 3]], np.uint32)
    k = L.orc_sel_from_repeats(rep.ctypes.data_as(C.c_void_p), 3, out.ctypes.data_as(C.c_void_p))
    assert out[:k].tolist() == [5, 5, 9, 9, 9]


def test_sort_goldens_from_sort_rs():
    """tests/it/sort.rs:28-241: DataBlock::sort over (Int64 | Decimal128, String) with asc / desc and LIMIT."""
    L = O.load()
    from databend_amd.device import make_views
    cases = golden("sort.json")
    assert len(cases) == 8
    for case in cases:
        cols = []
        for c in case["source"]:
            if c["kind"] == "String":
                cols.append(O.HostCol(T.T_STRING, make_views([s.encode() for s in c["values"]])))
            elif c["kind"] == "Decimal128":
                cols.append(O.HostCol(T.T_DEC128, O.i128_array(c["values"]), None, 38, 0))
            else:
                cols.append(O.HostCol(T.T_I64, np.array(c["values"], np.int64)))
        n = len(case["source"][0]["values"])
        keys = [cols[d["offset"]] for d in case["sort"]]
        desc = (C.c_uint8 * len(keys))(*[0 if d["asc"] else 1 for d in case["sort"]])
        nf = (C.c_uint8 * len(keys))(*[1 if d["nulls_first"] else 0 for d in case["sort"]])
        m = case["limit"] if 0 < case["limit"] < n else n
        perm = np.zeros(n, np.uint32)
        assert L.orc_sort_perm(O.cols(keys), desc, nf, len(keys), C.c_int64(n), C.c_int64(case["limit"]), perm.ctypes.data_as(C.c_void_p)) == 0
        for c, e in zip(case["source"], case["expected"]):
            assert [c["values"][i] for i in perm[:m]] == e["values"], (case["sort"], case["limit"])


def test_hash_index_cases_of_index_rs():
    """hash_index/index.rs:385-404: incoming (key, hash) pairs with colliding hashes / tags against a table that already
    holds (4, hash, 77): three new groups, every incoming row lands on its own key's group."""
    L = O.load()
    L.orc_hash_index_case.restype = C.c_int
    for inc, pay in (([(1, 123), (2, 456), (3, 123), (4, 44)], [(4, 44, 77)]),
                     ([(1, 11 << 48), (2, 22 << 48), (3, 33 << 48), (4, 44 << 48)], [(4, 44 << 48, 77)])):
        ik = np.array([k for k, _ in inc], np.uint64)
        ih = np.array([h for _, h in inc], np.uint64)
        pk = np.array([k for k, _, _ in pay], np.uint64)
        ph = np.array([h for _, h, _ in pay], np.uint64)
        pv = np.array([v for _, _, v in pay], np.uint64)
        outv = np.zeros(len(inc), np.uint64)
        new = L.orc_hash_index_case(16, ik.ctypes.data_as(C.c_void_p), ih.ctypes.data_as(C.c_void_p), len(inc), pk.ctypes.data_as(C.c_void_p),
                                    ph.ctypes.data_as(C.c_void_p), pv.ctypes.data_as(C.c_void_p), len(pay), outv.ctypes.data_as(C.c_void_p))
        assert new == 3
        assert dict(zip(ik.tolist(), outv.tolist())) == {1: 21, 2: 22, 3: 23, 4: 77}


def test_kernel_pass_scatter_and_concat_goldens():
    """kernel-pass.txt 'Scatter' (:211+) and 'Concat' (:21-53) through the oracle's statement of DataBlock::scatter
    (divide_indices_by_scatter_size + take, scatter.rs:20-66) and DataBlock::concat (concat.rs:62-340): values, validities, strings."""
    from tests import scatter_cases as SC
    L = O.load()
    cases = golden("kernel.json")
    sc = [c for c in cases if c["kind"] == "scatter"]
    cc = [c for c in cases if c["kind"] == "concat"]
    assert len(sc) == 1 and len(cc) == 1
    for case in sc:
        cols = SC.cells_to_columns(case["header"], case["source"])
        views = {}
        spec = []
        for kind, vals, valid in cols:
            if kind == "str":
                from databend_amd.device import make_views
                spec.append({"values": make_views(vals), "bits": [valid]})
            else:
                spec.append({"values": vals, "bits": [valid]})
        S = len(case["results"])
        out, starts, _ = SC.oracle_scatter(L, case["arg"], S, spec)
        assert starts == [0, 2, 4, 5]
        for d in range(S):
            for c, (kind, _, _) in enumerate(cols):
                o = out[d][c]
                vals = [bytes(v[4:4 + int(v[0])]) for v in o["values"]] if kind == "str" else o["values"].tolist()
                assert SC.render(kind, vals, o["bits"][0]) == [r[c] for r in case["results"][d]], (d, c)
    for case in cc:
        ncols = len(case["header"])
        for c in range(ncols):
            blks = [b[c] for b in case["blocks"]]
            if "values" not in blks[0]:
                continue          # Null / Array(Nothing) columns: outside the path's types
            rows = [len(b["values"]) for b in blks]
            valid = SC.oracle_concat_bits(L, [np.array(b["validity"], bool) if "validity" in b else None for b in blks], rows)
            if isinstance(blks[0]["values"][0], str):
                from databend_amd.device import make_views
                v = SC.oracle_concat_fixed(L, [make_views([x.encode() for x in b["values"]]) for b in blks])
                vals = [bytes(r[4:4 + int(r[0])]) for r in v]
                kind = "str"
            else:
                vals = SC.oracle_concat_fixed(L, [np.array(b["values"], np.int32) for b in blks]).tolist()
                kind = "int"
            assert SC.render(kind, vals, valid) == [r[c] for r in case["result"]], c


def test_scatter_then_concat_equals_take_oracle():
    """tests/it/kernel.rs:519-566 (test_scatter): scatter a random block by random indices, concat the pieces == take by the indices
    grouped by destination. Here over the oracle's own statements (the device runs the same property in test_gpu_scatter.py)."""
    from tests import scatter_cases as SC
    L = O.load()
    rng = np.random.default_rng(5)
    for _ in range(12):
        n = int(rng.integers(2, 300))
        S = int(rng.integers(2, 25))
        idx = rng.integers(0, S, n).astype(np.uint32)
        vals = rng.integers(-1000, 1000, n).astype(np.int64)
        valid = rng.random(n) > 0.3
        out, starts, rows = SC.oracle_scatter(L, idx, S, [{"values": vals, "bits": [valid]}])
        take_indices = [j for d in range(S) for j in range(n) if idx[j] == d]
        assert rows.tolist() == take_indices
        cat = SC.oracle_concat_fixed(L, [out[d][0]["values"] for d in range(S)])
        catv = SC.oracle_concat_bits(L, [out[d][0]["bits"][0] for d in range(S)], [starts[d + 1] - starts[d] for d in range(S)])
        assert np.array_equal(cat, vals[take_indices]) and np.array_equal(catv, valid[take_indices])


def test_multi_key_scatter_hash_is_pinned_to_the_golden_siphash():
    """HashFlightScatter::combine_hash_keys (flight_scatter_hash.rs:213-233) writes every key's u64 into ONE std DefaultHasher
    (`write_u64` = the 8 little-endian bytes, SipHash-1-3 with zero keys) and takes finish() % scatter_size. The reference holds no
    expected values for it; it is pinned here three ways:
      (1) the identity DefaultHasher{write_u64(h1); ...; write_u64(hk)}.finish() == siphash64(le(h1) || ... || le(hk)) ties it to the
          byte-string siphash64 that IS pinned on the reference's golden vectors (tests/golden/siphash.json: 0-, 3-, 4-, 8-, 9- and
          11-byte inputs, i.e. the block loop, the tail and the length byte): the oracle's combine equals the oracle's siphash64 of the
          concatenated bytes;
      (2) an independent second statement (tests/siphash_ref.py) agrees on random inputs;
      (3) a hand-worked vector: two keys 1 and 2 -> the 16-byte message 01 00.. 02 00.., worked through the paper's round function
          below with plain Python integers (no shared code with (1) or (2))."""
    from tests import siphash_ref as R
    L = O.load()
    rng = np.random.default_rng(11)
    # (3) hand-worked: SipHash-1-3, k0 = k1 = 0, message = le64(1) || le64(2), length 16
    M64 = (1 << 64) - 1
    rot = lambda x, b: ((x << b) | (x >> (64 - b))) & M64
    v0, v1, v2, v3 = 0x736f6d6570736575, 0x646f72616e646f6d, 0x6c7967656e657261, 0x7465646279746573

    def sipround(v0, v1, v2, v3):
        v0 = (v0 + v1) & M64; v1 = rot(v1, 13) ^ v0; v0 = rot(v0, 32)
        v2 = (v2 + v3) & M64; v3 = rot(v3, 16) ^ v2
        v0 = (v0 + v3) & M64; v3 = rot(v3, 21) ^ v0
        v2 = (v2 + v1) & M64; v1 = rot(v1, 17) ^ v2; v2 = rot(v2, 32)
        return v0, v1, v2, v3
    for m in (1, 2, 16 << 56):          # two message words, then the final word = length 16 in the top byte, no tail bytes
        v3 ^= m
        v0, v1, v2, v3 = sipround(v0, v1, v2, v3)
        v0 ^= m
    v2 ^= 0xFF
    for _ in range(3):
        v0, v1, v2, v3 = sipround(v0, v1, v2, v3)
    hand = v0 ^ v1 ^ v2 ^ v3
    assert hand == 18088007599891946824 == R.siphash13(np.array([1, 2], "<u8").tobytes())      # the committed known answer
    for nkeys in (2, 3, 5):
        n = 257
        hs = [rng.integers(0, 1 << 63, n, dtype=np.uint64) for _ in range(nkeys)]
        if nkeys == 2:
            hs[0][0], hs[1][0] = 1, 2
        cols = [O.HostCol(T.T_U64, h) for h in hs]
        # orc_scatter_indices over u64 KEY columns: every key is hashed (siphash64) and the hashes are combined; checked row by row
        # against both restatements of the combine
        idx = np.zeros(n, np.uint32)
        cnt = np.zeros(7, np.uint64)
        assert L.orc_scatter_indices(O.cols(cols), nkeys, C.c_int64(n), C.c_uint64(7), C.c_uint64(0), idx.ctypes.data_as(C.c_void_p),
                                     cnt.ctypes.data_as(C.c_void_p)) == 0
        for i in range(0, n, 16):
            key_hashes = [R.siphash64("u64", int(h[i])) for h in hs]
            msg = b"".join(int(x).to_bytes(8, "little") for x in key_hashes)
            # (1) the byte-string siphash64 of the oracle over the concatenated bytes
            from databend_amd.device import make_views_general
            views, buf = make_views_general([msg])
            out = np.zeros(1, np.uint64)
            sc = O.HostCol(T.T_STRING, views, None, buffers=[buf]).c()
            assert L.orc_siphash64(C.byref(sc), C.c_int64(1), out.ctypes.data_as(C.c_void_p)) == 0
            assert int(out[0]) % 7 == int(idx[i]), (nkeys, i)
            # (2) the independent statement
            assert R.scatter_index(key_hashes, 7) == int(idx[i])
    # the hand-worked value through the oracle's byte-string siphash64 as well
    views, buf = make_views_general([np.array([1, 2], "<u8").tobytes()])
    out = np.zeros(1, np.uint64)
    sc = O.HostCol(T.T_STRING, views, None, buffers=[buf]).c()
    assert L.orc_siphash64(C.byref(sc), C.c_int64(1), out.ctypes.data_as(C.c_void_p)) == 0
    assert int(out[0]) == hand


def test_arithmetic_decimal_txt_the_q1_shaped_expression_and_the_folded_constant():
    """arithmetic_decimal.txt (tests/golden/make_golden_arith_decimal.py): `l_extendedprice + (1 - l_discount) - l_quantity` over
    Decimal(15, 2) columns — the expression shape of TPC-H Q1's maps, with an integer operand — and the constant
    `1964831797.0000 - 0.0214642400000` (Decimal(14, 4) - Decimal(13, 13) -> Decimal(24, 13): the operands are rescaled by 10^9 and 1
    into a 128-bit result). The oracle evaluates the checked expression node by node; every node's DecimalSize must be the one the
    reference's type checker printed in the parent's signature."""
    from decimal import Decimal
    data = json.load(open(os.path.join(HERE, "golden", "arithmetic_decimal.json")))
    be = OracleBackend()
    seen_sizes = []

    class Recording(OracleBackend):
        def decimal(self, op, a, b, n):
            out = OracleBackend.decimal(self, op, a, b, n)
            seen_sizes.append((out.precision, out.scale))
            return out
    rec = Recording()
    checked, skipped = G.run_cases(data["cases"], rec)
    assert len(checked) == 1 and not skipped, skipped
    # post-order: minus<UInt8, Decimal(15,2)> -> (16,2); plus<Decimal(15,2), Decimal(16,2)> -> (17,2); minus<Decimal(17,2), Decimal(15,2)> -> (18,2)
    assert seen_sizes == [(16, 2), (17, 2), (18, 2)], seen_sizes
    for f in data["folded"]:
        node = G.parse_expr(f["expr"])
        got = G.evaluate(node, {}, be, 1)
        kind = G.parse_type(f["output_type"])
        assert kind[0] == "dec" and (got.precision, got.scale) == (kind[1], kind[2]), (f["ast"], got.precision, got.scale)
        assert got.ints()[:1] == [int(Decimal(f["output"]).scaleb(kind[2]))], (f["ast"], got.ints())
    assert len(data["folded"]) == 1

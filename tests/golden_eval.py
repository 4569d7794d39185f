"""Drives the reference's golden cases (tests/golden/*.json, extracted from the reference's own testdata) through a
backend: the CPU oracle (tests/test_golden_cpu.py) or the C-ABI on the GPU (tests/test_gpu_golden.py).

The `expr` of a case is the reference's checked / optimised expression text, e.g.
    plus<UInt8 NULL, UInt32 NULL>(a2, CAST<UInt32>(c AS UInt32 NULL))
    modulo<Int16, Int16>(plus<Int8, UInt8>(a, 3_u8), b)
    multiply<Decimal(10, 1), Decimal(1, 1)>(e, 0.5_d64(1,1))
It is parsed into calls / casts / literals / column references and evaluated node by node. Function nodes go to the
backend (the code under test); CAST nodes are the planner's and are evaluated here with numpy (lossless widenings,
anything -> Float64, Decimal -> Float64). Cases outside the hot path's function set are reported as skipped by reason.
"""
import re
from decimal import Decimal

import numpy as np

from databend_amd import _lib as T

NUM = {"Int8": (T.T_I8, np.int8), "Int16": (T.T_I16, np.int16), "Int32": (T.T_I32, np.int32), "Int64": (T.T_I64, np.int64),
       "UInt8": (T.T_U8, np.uint8), "UInt16": (T.T_U16, np.uint16), "UInt32": (T.T_U32, np.uint32), "UInt64": (T.T_U64, np.uint64),
       "Float32": (T.T_F32, np.float32), "Float64": (T.T_F64, np.float64)}
NP_OF_CODE = {c: d for c, d in NUM.values()}
SUFFIX = {"u8": "UInt8", "i8": "Int8", "u16": "UInt16", "i16": "Int16", "u32": "UInt32", "i32": "Int32", "u64": "UInt64", "i64": "Int64",
          "f32": "Float32", "f64": "Float64"}
ARITH = {"plus": T.OP_PLUS, "minus": T.OP_MINUS, "multiply": T.OP_MULTIPLY, "divide": T.OP_DIVIDE, "div": T.OP_INTDIV, "modulo": T.OP_MODULO}
CMPS = {"eq": T.CMP_EQ, "noteq": T.CMP_NOTEQ, "lt": T.CMP_LT, "lte": T.CMP_LTE, "gt": T.CMP_GT, "gte": T.CMP_GTE}
INT_PROPS = {T.T_I8: (3, 0), T.T_U8: (3, 0), T.T_I16: (5, 0), T.T_U16: (5, 0), T.T_I32: (10, 0), T.T_U32: (10, 0), T.T_I64: (19, 0), T.T_U64: (20, 0)}


class Skip(Exception):
    """the case is outside what this evaluator / the hot path covers; .args[0] = reason"""


class Val:
    def __init__(self, dtype, arr, validity=None, precision=0, scale=0, is_scalar=False):
        self.dtype, self.arr, self.validity = dtype, arr, validity
        self.precision, self.scale, self.is_scalar = precision, scale, is_scalar

    @property
    def is_decimal(self):
        return self.dtype in (T.T_DEC64, T.T_DEC128, T.T_DEC256)

    def ints(self):
        """python ints of a decimal / integer value"""
        if self.dtype in (T.T_DEC128, T.T_DEC256):
            k = 2 if self.dtype == T.T_DEC128 else 4
            w = np.ascontiguousarray(self.arr).view(np.uint64).reshape(-1, k)
            out = []
            for row in w:
                v = sum(int(x) << (64 * j) for j, x in enumerate(row))
                out.append(v - (1 << (64 * k)) if v >> (64 * k - 1) else v)
            return out
        return [int(x) for x in self.arr]


def parse_type(t):
    t = t.replace(" NULL", "").strip()
    m = re.fullmatch(r"Decimal\((\d+), ?(\d+)\)", t)
    if m:
        return ("dec", int(m.group(1)), int(m.group(2)))
    if t == "Boolean":
        return ("bool",)
    return ("num", t) if t in NUM else ("other", t)


def i128_array(ints):
    out = np.zeros((len(ints), 2), dtype=np.uint64)
    for i, v in enumerate(ints):
        v = int(v) & ((1 << 128) - 1)
        out[i, 0], out[i, 1] = v & 0xFFFFFFFFFFFFFFFF, v >> 64
    return out.reshape(-1)


def i256_array(ints):
    out = np.zeros((len(ints), 4), dtype=np.uint64)
    for i, v in enumerate(ints):
        v = int(v) & ((1 << 256) - 1)
        for j in range(4):
            out[i, j] = (v >> (64 * j)) & 0xFFFFFFFFFFFFFFFF
    return out.reshape(-1)


def dec_val(ints, p, s, validity=None, is_scalar=False):
    """a decimal value in the storage class of its precision (DecimalDataType::from(size), decimal.rs:1713-1722)"""
    if p <= 18:
        return Val(T.T_DEC64, np.array(ints, dtype=np.int64), validity, p, s, is_scalar)
    if p <= 38:
        return Val(T.T_DEC128, i128_array(ints), validity, p, s, is_scalar)
    return Val(T.T_DEC256, i256_array(ints), validity, p, s, is_scalar)


def column_val(entry):
    kind = parse_type(entry["type"])
    validity = np.array(entry["validity"], dtype=bool) if "validity" in entry else None
    if kind[0] == "num":
        code, npd = NUM[kind[1]]
        vals = [float(v) if isinstance(v, str) else v for v in entry["values"]]
        return Val(code, np.array(vals, dtype=npd), validity)
    if kind[0] == "dec":
        p, s = kind[1], kind[2]
        ints = [int(Decimal(str(v)).scaleb(s)) for v in entry["values"]]
        return dec_val(ints, p, s, validity)
    if kind[0] == "bool":
        return Val(T.T_BOOL, np.array(entry["values"], dtype=bool), validity)
    raise Skip(f"column type {entry['type']}")


# ---- parser ---------------------------------------------------------------------------------------------------------
TOKEN = re.compile(r"\s*(CAST<|[A-Za-z_][A-Za-z_0-9]*|-?\d+(?:\.\d+)?(?:e-?\d+)?_[a-z]\d+(?:\(\d+, ?\d+\))?|-?\d+(?:\.\d+)?|[<>(),])")


def split_top(s, sep=","):
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "(<":
            depth += 1
        elif ch in ")>":
            depth -= 1
        if ch == sep and depth == 0:
            out.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


def parse_expr(s):
    s = s.strip()
    m = re.fullmatch(r"CAST<(.*?)>\((.*) AS ([^()]*(?:\(\d+, ?\d+\))?(?: NULL)?)\)", s)
    if m and _balanced(m.group(2)):
        return ("cast", m.group(1).strip(), parse_expr(m.group(2)), m.group(3).strip())
    m = re.fullmatch(r"(\w+)<(.*?)>\((.*)\)", s)
    if m and _balanced(m.group(3)) and _balanced_angle(m.group(2)):
        return ("call", m.group(1), split_top(m.group(2)), [parse_expr(a) for a in split_top(m.group(3))])
    m = re.fullmatch(r"(-?\d+(?:\.\d+)?(?:e-?\d+)?)_([a-z]\d+)(?:\((\d+), ?(\d+)\))?", s)
    if m:
        return ("lit", m.group(1), m.group(2), m.group(3), m.group(4))
    if s in ("true", "false"):
        return ("lit", s, "bool", None, None)
    if re.fullmatch(r"\w+", s):
        return ("col", s)
    raise Skip(f"expression shape: {s[:60]}")


def _balanced(s):
    d = 0
    for ch in s:
        d += ch == "("
        d -= ch == ")"
        if d < 0:
            return False
    return d == 0


def _balanced_angle(s):
    return s.count("(") == s.count(")")


# ---- evaluation -----------------------------------------------------------------------------------------------------
def and_validity(a, b, n):
    va = None if a.validity is None else (np.repeat(a.validity[:1], n) if a.is_scalar else a.validity)
    vb = None if b.validity is None else (np.repeat(b.validity[:1], n) if b.is_scalar else b.validity)
    if va is None:
        return vb
    return va if vb is None else (va & vb)


def cast(v, to_txt, n):
    to = parse_type(to_txt)
    if to[0] == "num":
        code, npd = NUM[to[1]]
        if v.is_decimal:
            if to[1] != "Float64":
                raise Skip("CAST decimal -> " + to[1])
            f = np.array([x / 10 ** v.scale for x in v.ints()], dtype=np.float64)   # decimal_to_float64: value / 10^scale
            return Val(T.T_F64, f, v.validity, is_scalar=v.is_scalar)
        src = v.arr
        if np.issubdtype(npd, np.integer) and not np.array_equal(src.astype(npd).astype(src.dtype), src):
            raise Skip("CAST that changes values")
        return Val(code, src.astype(npd), v.validity, is_scalar=v.is_scalar)
    if to[0] == "dec":
        if v.is_decimal and (v.precision, v.scale) == (to[1], to[2]):
            return v
        raise Skip("CAST to another DecimalSize")
    raise Skip("CAST to " + to_txt)


def literal(node):
    _, txt, suf, p, s = node
    if suf == "bool":
        return Val(T.T_BOOL, np.array([txt == "true"]), is_scalar=True)
    if suf in SUFFIX:
        code, npd = NUM[SUFFIX[suf]]
        return Val(code, np.array([float(txt) if "f" in suf else int(txt)], dtype=npd), is_scalar=True)
    if suf in ("d64", "d128", "d256"):
        p, s = int(p), int(s)
        return dec_val([int(Decimal(txt).scaleb(s))], p, s, is_scalar=True)
    raise Skip("literal suffix " + suf)


def evaluate(node, cols, backend, n):
    kind = node[0]
    if kind == "col":
        if node[1] not in cols:
            raise Skip("unknown column " + node[1])
        return column_val(cols[node[1]])
    if kind == "lit":
        return literal(node)
    if kind == "cast":
        return cast(evaluate(node[2], cols, backend, n), node[3], n)
    _, name, _targs, args = node
    vals = [evaluate(a, cols, backend, n) for a in args]
    if name in ARITH and len(vals) == 2:
        a, b = vals
        if a.dtype == T.T_BOOL or b.dtype == T.T_BOOL:
            raise Skip("arithmetic on Boolean")
        if a.is_decimal or b.is_decimal:
            if (not a.is_decimal and a.dtype in (T.T_F32, T.T_F64)) or (not b.is_decimal and b.dtype in (T.T_F32, T.T_F64)):
                raise Skip("decimal with float operand")
            if name in ("div", "modulo"):
                raise Skip("decimal div / modulo")
            out = backend.decimal(ARITH[name], a, b, n)
        else:
            out = backend.arith(ARITH[name], a, b, n)
        out.validity = and_validity(a, b, n)
        return out
    if name == "minus" and len(vals) == 1:
        a = vals[0]
        if a.is_decimal:   # register_decimal_minus: the same DecimalSize and storage class (arithmetic.rs:514-590)
            out = backend.decimal_neg(a, n)
            out.validity = a.validity
            return out
        # negate(x) = 0 - x in the same result type (Int8 -> Int16, UInt32 -> Int64, Float64 -> Float64)
        zero = Val(a.dtype, np.zeros(1, dtype=NP_OF_CODE[a.dtype]), is_scalar=True)
        out = backend.arith(T.OP_MINUS, zero, a, n)
        out.validity = a.validity
        return out
    if name in CMPS and len(vals) == 2:
        a, b = vals
        bits = backend.cmp(CMPS[name], a, b, n)
        return Val(T.T_BOOL, bits, and_validity(a, b, n))
    raise Skip("function " + name)


def expected_output(case):
    outc = case["columns"]["Output"]
    kind = parse_type(outc["type"])
    validity = np.array(outc["validity"], dtype=bool) if "validity" in outc else None
    return kind, outc, validity


def check_case(case, backend):
    """evaluates one golden case; raises Skip (with the reason) or AssertionError"""
    n = case["n"]
    kind, outc, exp_valid = expected_output(case)
    node = parse_expr(case["expr"])
    if node[0] in ("lit",):
        raise Skip("constant-folded expression")
    got = evaluate(node, case["columns"], backend, n)
    valid = np.ones(n, dtype=bool) if exp_valid is None else exp_valid
    if exp_valid is not None or got.validity is not None:
        gv = np.ones(n, dtype=bool) if got.validity is None else got.validity
        assert np.array_equal(gv, valid), (case["ast"], "validity", gv, valid)
    if kind[0] == "num":
        code, npd = NUM[kind[1]]
        assert got.dtype == code, (case["ast"], "result type", got.dtype, code)
        exp = np.array([float(v) if isinstance(v, str) else v for v in outc["values"]], dtype=npd)
        g = np.asarray(got.arr)[:n]
        if np.issubdtype(npd, np.floating):
            # the golden file prints floats with ~8 significant digits: compare at the printed precision (bit-exactness of
            # the f64 kernels is pinned by the numpy statements of tests/test_oracle_cpu.py)
            assert np.allclose(g[valid], exp[valid], rtol=2e-7, atol=0, equal_nan=True), (case["ast"], g, exp)
        else:
            assert np.array_equal(g[valid], exp[valid]), (case["ast"], g, exp)
    elif kind[0] == "dec":
        assert (got.precision, got.scale) == (kind[1], kind[2]), (case["ast"], "DecimalSize", got.precision, got.scale, kind)
        exp = [int(Decimal(str(v)).scaleb(kind[2])) for v in outc["values"]]
        g = got.ints()[:n]
        assert [x for x, v in zip(g, valid) if v] == [x for x, v in zip(exp, valid) if v], (case["ast"], g, exp)
    elif kind[0] == "bool":
        g = np.asarray(got.arr, dtype=bool)[:n]
        exp = np.array(outc["values"], dtype=bool)
        assert np.array_equal(g[valid], exp[valid]), (case["ast"], g, exp)
    else:
        raise Skip("output type " + outc["type"])


def run_cases(cases, backend):
    """-> (checked asts, {reason: count})"""
    checked, skipped = [], {}
    for case in cases:
        try:
            check_case(case, backend)
            checked.append(case["ast"])
        except Skip as e:
            r = e.args[0].split(":")[0] if e.args[0].startswith(("function", "column type", "CAST")) else e.args[0]
            skipped[r] = skipped.get(r, 0) + 1
    return checked, skipped

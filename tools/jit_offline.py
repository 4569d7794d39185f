#!/usr/bin/env python3
"""Compiles the run-time specialised fused-aggregation kernel of a query shape WITHOUT a GPU (hiprtc needs no device) and
reports what the hardware will run: registers, scratch, and the instruction mix of its loops (tools/isa_loops.py). This is how
the specialised kernels are optimised offline; one GPU run then confirms the time.

    python tools/jit_offline.py [q1|plain4] [--slots 4] [--defs "-DFA_JIT_ROWS=4"] [--keep /tmp/out]

(--defs goes through DBHIP_FAGG_JIT_DEFS, which only an experiments build of the library reads: `make -C databend_amd/csrc clean all EXPERIMENTS=1`;
--slots 260 = 4 slots + 0x100: the FA_MULTI specialisation of a pipelined table's multi-block launches.)
"""
import argparse
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from databend_amd import _lib as L  # noqa: E402
from databend_amd import device as D  # noqa: E402


class FakeCol:
    """types / nullability only: the offline compile never dereferences a column"""

    def __init__(self, dtype, precision=0, scale=0, nullable=False, is_scalar=False):
        self.dtype, self.precision, self.scale, self.nullable, self.is_scalar, self.n = dtype, precision, scale, nullable, is_scalar, 1

    def c(self):
        col = L.Col()
        col.type, col.is_scalar, col.data = self.dtype, int(self.is_scalar), 0x10000
        col.validity = 0x20000 if self.nullable else None
        col.precision, col.scale = self.precision, self.scale
        return col


def q1_shape():
    ship, qty, price, disc, tax = FakeCol(L.T_DATE), FakeCol(L.T_DEC64, 15, 2), FakeCol(L.T_DEC64, 15, 2), FakeCol(L.T_DEC64, 15, 2), FakeCol(L.T_DEC64, 15, 2)
    p = D.ExprProgram([ship, qty, price, disc, tax])
    f = p.cmp(L.EX_LTE, p.load(0), p.const(10471, L.T_DATE))
    q, pr, di, ta = p.load(1), p.load(2), p.load(3), p.load(4)
    one = p.const(1, L.T_U8)
    om = p.arith(L.EX_MINUS, one, di, keep=(one, di))
    dp = p.arith(L.EX_MULTIPLY, pr, om, keep=(pr,))
    op = p.arith(L.EX_PLUS, one, ta)
    ch = p.arith(L.EX_MULTIPLY, dp, op, keep=(dp,))
    aggs = [(L.AGG_SUM, L.T_DEC64, 15, 2, 0), (L.AGG_SUM, L.T_DEC64, 15, 2, 0), (L.AGG_SUM, L.T_DEC128, 31, 4, 0),
            (L.AGG_SUM, L.T_DEC128, 38, 6, 0), (L.AGG_SUM, L.T_DEC64, 15, 2, 0), (L.AGG_COUNT, 0, 0, 0, 0)]
    return [L.T_STRING, L.T_STRING], [0, 0], aggs, [FakeCol(L.T_STRING), FakeCol(L.T_STRING)], p, [q, pr, dp, ch, di, None], f


def q1_rescale_shape():
    """Q1 with scales that force a rescale: extendedprice / discount as Decimal(15,8) — their product is a ROUNDING multiply
    (scale 16 -> 12: divides by 10^4, decimal/src/arithmetic.rs:212-243) — and sum(quantity / price), a decimal divide"""
    ship, qty, price, disc = FakeCol(L.T_DATE), FakeCol(L.T_DEC64, 15, 2), FakeCol(L.T_DEC64, 15, 8), FakeCol(L.T_DEC64, 15, 8)
    p = D.ExprProgram([ship, qty, price, disc])
    f = p.cmp(L.EX_LTE, p.load(0), p.const(10471, L.T_DATE))
    q, pr, di = p.load(1), p.load(2), p.load(3)
    prod = p.arith(L.EX_MULTIPLY, pr, di, keep=(pr,))
    quo = p.arith(L.EX_DIVIDE, q, pr, keep=(q,))
    aggs = [(L.AGG_SUM, L.T_DEC64, 15, 2, 0), (L.AGG_SUM, p.types[prod], p.size[prod][0], p.size[prod][1], 0),
            (L.AGG_SUM, p.types[quo], p.size[quo][0], p.size[quo][1], 0), (L.AGG_COUNT, 0, 0, 0, 0)]
    return [L.T_STRING, L.T_STRING], [0, 0], aggs, [FakeCol(L.T_STRING), FakeCol(L.T_STRING)], p, [q, prod, quo, None], f


def plain4_shape():
    a = FakeCol(L.T_I64)
    p = D.ExprProgram([a])
    aggs = [(L.AGG_SUM, L.T_I64, 0, 0, 0), (L.AGG_COUNT, 0, 0, 0, 0)]
    return [L.T_I64], [0], aggs, [FakeCol(L.T_I64)], p, [("input", 0), None], -1


def general_shape():
    """i64 key; sum(nullable i64), count(*), sum(Decimal64), max(nullable i64) over plain input columns (tests/test_gpu_fused.py)"""
    a, d = FakeCol(L.T_I64, nullable=True), FakeCol(L.T_DEC64, 15, 2)
    p = D.ExprProgram([a, d])
    aggs = [(L.AGG_SUM, L.T_I64, 0, 0, 1), (L.AGG_COUNT, 0, 0, 0, 0), (L.AGG_SUM, L.T_DEC64, 15, 2, 0), (L.AGG_MAX, L.T_I64, 0, 0, 1)]
    return [L.T_I64], [0], aggs, [FakeCol(L.T_I64)], p, [("input", 0), None, ("input", 1), ("input", 0)], -1


def compile_shape(shape, slots, defs):
    key_types, key_nullable, aggs, keys, p, arg_regs, filter_reg = shape
    lib = L.load_library()
    if defs:
        os.environ["DBHIP_FAGG_JIT_DEFS"] = defs
    kt = (C.c_int32 * len(key_types))(*key_types)
    kn = (C.c_uint8 * len(key_types))(*key_nullable)
    ad = (L.AggDesc * len(aggs))()
    for i, (kind, t, pr, sc, nu) in enumerate(aggs):
        ad[i].kind, ad[i].arg_type, ad[i].arg_precision, ad[i].arg_scale, ad[i].arg_nullable = kind, t, pr, sc, nu
    regs = (C.c_int32 * len(aggs))()
    for i, r in enumerate(arg_regs):
        regs[i] = -(2 ** 31) if r is None else (-(1 + r[1]) if isinstance(r, tuple) else r)
    ap = L.AggProgram()
    cprog = p.c_program()
    cin = (L.Col * len(p.inputs))(*[c.c() for c in p.inputs])
    ap.prog, ap.n_ins = C.cast(cprog, C.c_void_p), len(p.ins)
    ap.inputs, ap.n_inputs = C.cast(cin, C.c_void_p), len(p.inputs)
    ap.filter_reg, ap.arg_regs = filter_reg, C.cast(regs, C.c_void_p)
    ck = (L.Col * len(keys))(*[c.c() for c in keys])
    code = C.create_string_buffer(1 << 20)
    log = C.create_string_buffer(1 << 16)
    fn = lib.dbhip_jit_offline
    fn.restype = C.c_int64
    n = fn(kt, kn, len(key_types), ad, len(aggs), ck, C.byref(ap), slots, code, C.c_int64(len(code)), log, C.c_int64(len(log)))
    if n < 0:
        raise SystemExit(f"offline compile failed ({n}): {log.value.decode(errors='replace')[-3000:]} {lib.dbhip_last_error().decode()}")
    return code.raw[:n]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("shape", nargs="?", default="q1")
    ap.add_argument("--slots", type=int, default=4)
    ap.add_argument("--defs", default="")
    ap.add_argument("--keep", default="/tmp/jit_offline")
    a = ap.parse_args()
    code = compile_shape({"q1": q1_shape, "q1rescale": q1_rescale_shape, "plain4": plain4_shape, "general": general_shape}[a.shape](), a.slots, a.defs)
    open(a.keep + ".co", "wb").write(code)
    llvm = "/opt/rocm/lib/llvm/bin"
    asm = subprocess.run([f"{llvm}/llvm-objdump", "-d", a.keep + ".co"], capture_output=True, text=True, check=True).stdout
    open(a.keep + ".s", "w").write(asm)
    notes = subprocess.run([f"{llvm}/llvm-readelf", "--notes", a.keep + ".co"], capture_output=True, text=True).stdout
    for line in notes.splitlines():
        if any(k in line for k in (".vgpr_count", ".sgpr_count", ".private_segment_fixed_size", ".group_segment_fixed_size", ".vgpr_spill")):
            print(line.strip())
    print(f"code object {len(code)} bytes -> {a.keep}.co / .s")
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_loops.py"), a.keep + ".s"])


if __name__ == "__main__":
    main()

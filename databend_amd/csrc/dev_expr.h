// dev_expr.h — the register-program interpreter behind dbhip_expr_eval (k_expr.hip) and the fused
// filter -> map -> partial-aggregate kernel (k_fagg.hip).
//
// Reference: Evaluator::run walks the Expr tree and materialises ONE column per call node
// (src/query/expression/src/evaluator.rs:229-464; BlockOperator::Map, src/query/sql/src/evaluator/block_operator.rs:42-85).
// Here the host flattens the tree(s) post-order into a short register program that one launch interprets with a
// wave-uniform instruction stream (every lane executes the same instruction: the interpreter is scalar-unit work) over a
// register file in LDS ([slot][row slot][lane] u64: conflict-free ds_read/ds_write_b64). 128-bit values (Decimal128)
// take two consecutive slots (lo, hi).
//
// Per-node semantics are those of the per-node kernels: numeric_basic_arithmetic.rs:255-427 / comparison.rs:98-112
// (k_arith.hip, k_cmp_filter.hip) and binary_decimal (decimal/src/arithmetic.rs:190-316, dev_decimal.h). Row errors
// ("divided by zero", "Decimal overflow") are raised only for rows that are live (inside the block, selected by the
// filter that precedes the maps) and whose nullable inputs feeding the node are all valid (function.rs:534-556).
#pragma once
#include "dev_common.h"
#include "dev_decimal.h"
#include "dev_load.h"

constexpr int EX_MAX_INS = 24;
constexpr int EX_MAX_REGS = 16;    // user registers
constexpr int EX_MAX_INPUTS = 8;
constexpr int EX_MAX_DEC = 6;      // decimal call nodes per program
constexpr int EX_MAX_ROOTS = 8;    // results a program may expose (fused aggregate: filter + arguments)
constexpr int EX_MAX_SLOTS = 24;   // LDS slots

struct ExIns {
  int16_t op, dst, a, b, c;  // LDS slots after compilation (c: third operand of IF)
  int16_t type;              // result type (host side checks only)
  int8_t acls, bcls, ocls;   // CLS_SIGNED / CLS_UNSIGNED / CLS_FLOAT of the operands and the result
  int8_t norm_sh;            // result width: shift that sign-/zero-extends from the result's bits (0 for 64-bit)
  int8_t norm_signed, norm_f32;
  int8_t a_wide, b_wide, o_wide;  // 128-bit operands / result
  int8_t a_dec, b_dec;            // decimal node: the operand is a decimal (else an integer, other_to_decimal)
  int8_t dec_idx;                 // index into ExProg::dec, -1 for non-decimal nodes
  uint8_t dep;                    // nullable inputs the node depends on (bit c = input column c)
  uint8_t _pad;
  uint64_t imm, imm_hi;
};

struct ExProg {
  ExIns ins[EX_MAX_INS];
  DecOp dec[EX_MAX_DEC];
  const void* in_data[EX_MAX_INPUTS];
  const uint8_t* in_valid[EX_MAX_INPUTS];
  int64_t in_voff[EX_MAX_INPUTS];
  int32_t in_type[EX_MAX_INPUTS];   // load kind, see ex_load
  int32_t in_scalar[EX_MAX_INPUTS];
  int32_t in_slot[EX_MAX_INPUTS];   // LDS slot of input column c (-1: the program never reads it)
  int32_t in_wide_ord[EX_MAX_INPUTS];  // 128-bit input columns: 0 / 1 = which of the (at most two) hi-word staging registers, else -1
  int32_t in_has_valid[EX_MAX_INPUTS]; // the input column carries a validity Bitmap (part of the program's shape)
  int32_t n_ins, n_inputs, n_slots;
  int32_t n_filter_ins;             // leading instructions that compute the filter (0 = no filter stage)
  int32_t filter_slot;              // slot of the filter's Boolean (-1: none)
  uint32_t filter_dep;              // nullable inputs the filter depends on (a NULL predicate drops the row)
  uint32_t* err_words;              // preset to all ones (may be NULL)
  unsigned long long* err_count;    // may be NULL
};

// load kinds (host: ex_load_kind): the common 8-byte case is the first test
enum { LK_8 = 0, LK_S4 = 1, LK_U4 = 2, LK_F4 = 3, LK_S2 = 4, LK_U2 = 5, LK_S1 = 6, LK_U1 = 7, LK_BOOL = 8, LK_16 = 9 };
// input columns live in global memory; a pointer that reaches the kernel inside a by-value struct is generic to the compiler
// (flat_load: the LDS aperture check and both wait counters). FA_JIT_GLOBAL (run-time specialised kernel): say so.
#if defined(DBHIP_JIT) && defined(FA_JIT_GLOBAL)
#define EX_GPTR(T, p) ((const __attribute__((address_space(1))) T*)(p))
#else
#define EX_GPTR(T, p) ((const T*)(p))
#endif
__device__ __forceinline__ uint64_t ex_load(const void* p, int kind, int64_t i) {
  if (kind == LK_8) return EX_GPTR(uint64_t, p)[i];
  if (kind == LK_S4) return (uint64_t)(int64_t)EX_GPTR(int32_t, p)[i];
  if (kind == LK_U4) return EX_GPTR(uint32_t, p)[i];
  if (kind == LK_F4) return (uint64_t)__double_as_longlong((double)((const float*)p)[i]);
  if (kind == LK_S2) return (uint64_t)(int64_t)((const int16_t*)p)[i];
  if (kind == LK_U2) return ((const uint16_t*)p)[i];
  if (kind == LK_S1) return (uint64_t)(int64_t)((const int8_t*)p)[i];
  if (kind == LK_U1) return ((const uint8_t*)p)[i];
  if (kind == LK_16) return EX_GPTR(uint64_t, p)[2 * i];
  return bit_get((const uint8_t*)p, i);
}

__device__ __forceinline__ double ex_to_f64(uint64_t w, int cls) {
  if (cls == CLS_FLOAT) return __longlong_as_double((long long)w);
  if (cls == CLS_SIGNED) return (double)(int64_t)w;
  return (double)w;
}

// widened register image of `w` at the node's result type, from the host-decoded width (no type switch)
__device__ __forceinline__ uint64_t ex_norm(uint64_t w, const ExIns& I) {
  if (I.norm_f32) return (uint64_t)__double_as_longlong((double)(float)__longlong_as_double((long long)w));
  if (I.norm_sh == 0) return w;
  return I.norm_signed ? (uint64_t)(((int64_t)(w << I.norm_sh)) >> I.norm_sh) : ((w << I.norm_sh) >> I.norm_sh);
}

__device__ __forceinline__ int ex_cmp3(uint64_t a, uint64_t b, int cls) {
  if (cls == CLS_SIGNED) return ((int64_t)a > (int64_t)b) - ((int64_t)a < (int64_t)b);
  if (cls == CLS_UNSIGNED) return (a > b) - (a < b);
  const double x = __longlong_as_double((long long)a), y = __longlong_as_double((long long)b);
  const bool xn = x != x, yn = y != y;
  if (xn || yn) return (int)xn - (int)yn;  // OrderedFloat: NaN largest, NaN == NaN
  return (x > y) - (x < y);
}

enum {
  EX_LOAD = 0, EX_CONST = 1, EX_PLUS = 2, EX_MINUS = 3, EX_MULTIPLY = 4, EX_DIVIDE = 5,
  EX_EQ = 6, EX_NOTEQ = 7, EX_LT = 8, EX_LTE = 9, EX_GT = 10, EX_GTE = 11,
  EX_AND = 12, EX_OR = 13, EX_NOT = 14, EX_CAST = 15, EX_IF = 16, EX_IS_TRUE = 17,
  EX_DEC = 32  // internal: PLUS / MINUS / MULTIPLY / DIVIDE on decimals (the DecOp says which)
};

// Interprets instructions [pc0, pc1) for the ROWS row slots of this lane. `ex_regs` = the workgroup's LDS register file
// ([slot][ROWS][256] u64), `live[k]`: the row may raise, `vmask[k]`: bit c = input column c is valid (or not nullable) there.
// `P`: the program's metadata, `PA`: its pointers (the same object ahead of time; under DBHIP_JIT — the run-time compiled
// specialisation, fagg_device.h — P is a constexpr, the pc loop is unrolled, every switch below folds, and ex_regs is a
// per-lane array in VGPRs indexed by constants).
// DIV: the instantiation carries the long divisions of the rounding decimal multiply / divide (only programs with such a node launch
// it: ~16 more VGPRs); the run-time specialised kernels always do (their DecOps are constants: nothing is paid by programs without one).
template <int ROWS, bool DIV = false>
__device__ __forceinline__ void ex_interpret(const ExProg& P, const ExProg& PA, uint64_t* ex_regs, int tid, int pc0, int pc1,
                                             const int64_t (&row)[ROWS], const bool (&live)[ROWS],
                                             const uint32_t (&vmask)[ROWS]) {
#undef EX_REG
#ifdef DBHIP_JIT
#define EX_REG(r, k) ex_regs[(r) * ROWS + (k)]
  _Pragma("unroll")
  for (int pc = pc0; pc < pc1; ++pc) {
    const ExIns I = P.ins[pc];
#else
#define EX_REG(r, k) ex_regs[((r) * ROWS + (k)) * 256 + tid]
  ExIns nxt = P.ins[pc0 < EX_MAX_INS ? pc0 : 0];
  for (int pc = pc0; pc < pc1; ++pc) {
    const ExIns I = nxt;
    nxt = P.ins[pc + 1 < EX_MAX_INS ? pc + 1 : 0];   // the scalar load of the next instruction overlaps this one's work
#endif
    const int acls = I.acls, bcls = I.bcls, ocls = I.ocls;
#define EX_ROWS_DO(EXPR)                                  \
  _Pragma("unroll") for (int k = 0; k < ROWS; ++k) {      \
    const uint64_t x = EX_REG(I.a, k);                    \
    const uint64_t y = EX_REG(I.b, k);                    \
    (void)x; (void)y;                                     \
    EX_REG(I.dst, k) = (EXPR);                            \
  }
#define EX_ROWS_DO1(EXPR)                                 \
  _Pragma("unroll") for (int k = 0; k < ROWS; ++k) {      \
    const uint64_t x = EX_REG(I.a, k);                    \
    (void)x;                                              \
    EX_REG(I.dst, k) = (EXPR);                            \
  }
// 128-bit view of an operand: wide = two slots, else the 64-bit image extended by its class
#define EX_RD128(slot, wide, cls, k)                                                                      \
  ((i128)(((u128)((wide) ? EX_REG((slot) + 1, k) : (((cls) == CLS_SIGNED && (EX_REG(slot, k) >> 63)) ? ~0ULL : 0ULL)) << 64) | \
          (u128)EX_REG(slot, k)))
#define EX_RAISE(k)                                                                                        \
  do {                                                                                                     \
    if (live[k] && ((vmask[k] & I.dep) == I.dep)) { /* NULL / filtered / padding rows never raise (function.rs:536-543) */ \
      if (PA.err_words) atomicAnd(&PA.err_words[row[k] >> 5], ~(1u << (row[k] & 31)));                     \
      if (PA.err_count) atomicAdd(PA.err_count, 1ULL);                                                     \
    }                                                                                                      \
  } while (0)
    switch (I.op) {
      case EX_CONST:
#pragma unroll
        for (int k = 0; k < ROWS; ++k) {
          EX_REG(I.dst, k) = I.imm;
          if (I.o_wide) EX_REG(I.dst + 1, k) = I.imm_hi;
        }
        break;
      case EX_PLUS:
        if (ocls == CLS_FLOAT) EX_ROWS_DO(ex_norm((uint64_t)__double_as_longlong(ex_to_f64(x, acls) + ex_to_f64(y, bcls)), I))
        else EX_ROWS_DO(ex_norm(x + y, I))
        break;
      case EX_MINUS:
        if (ocls == CLS_FLOAT) EX_ROWS_DO(ex_norm((uint64_t)__double_as_longlong(ex_to_f64(x, acls) - ex_to_f64(y, bcls)), I))
        else EX_ROWS_DO(ex_norm(x - y, I))
        break;
      case EX_MULTIPLY:
        if (ocls == CLS_FLOAT) EX_ROWS_DO(ex_norm((uint64_t)__double_as_longlong(ex_to_f64(x, acls) * ex_to_f64(y, bcls)), I))
        else EX_ROWS_DO(ex_norm(x * y, I))
        break;
      case EX_DIVIDE:
#pragma unroll
        for (int k = 0; k < ROWS; ++k) {
          const double a = ex_to_f64(EX_REG(I.a, k), acls), bb = ex_to_f64(EX_REG(I.b, k), bcls);
          uint64_t r = 0;
          if (bb == 0.0) EX_RAISE(k);
          else r = (uint64_t)__double_as_longlong(a / bb);
          EX_REG(I.dst, k) = r;
        }
        break;
      case EX_DEC: {
        const DecOp* D = &P.dec[I.dec_idx];
        const bool t128 = D->t_is_128 != 0;
        if (D->trivial) {  // operands already at their bound sizes, a plain wrapping op in T (e.g. Q1's products)
          const int dop = D->op;
#pragma unroll
          for (int k = 0; k < ROWS; ++k) {
            const i128 av = wrap_T(EX_RD128(I.a, I.a_wide, acls, k), t128), bv = wrap_T(EX_RD128(I.b, I.b_wide, bcls, k), t128);
            const u128 ua = (u128)av, ub = (u128)bv;
            const i128 r = wrap_T((i128)(dop == DBHIP_OP_PLUS ? ua + ub : (dop == DBHIP_OP_MINUS ? ua - ub : ua * ub)), t128);
            EX_REG(I.dst, k) = (uint64_t)(u128)r;
            if (I.o_wide) EX_REG(I.dst + 1, k) = (uint64_t)((u128)r >> 64);
          }
#ifdef DBHIP_JIT
#define EX_DIV_OK true
#else
#define EX_DIV_OK DIV
#endif
        } else if (EX_DIV_OK && dec_op_needs_division(*D)) {
          // rounding multiply / divide (do_round_mul / do_round_div, types/decimal.rs:759-797,1024-1064): the long divisions of
          // dec_row. Interpreter: the register file is LDS, so the slots can be walked by a ROLLED loop — one copy of the
          // division code, not ROWS; specialised kernel: registers indexed by constants, unrolled, the divisor 10^k a constant.
          uint32_t bad = 0;
#ifdef DBHIP_JIT
#pragma unroll
#else
#pragma unroll 1
#endif
          for (int k = 0; k < ROWS; ++k) {
            const i128 av = EX_RD128(I.a, I.a_wide, acls, k), bv = EX_RD128(I.b, I.b_wide, bcls, k);
            i128 r;
            if (!dec_row(*D, av, bv, I.a_dec != 0, I.b_dec != 0, t128, &r)) {
              bad |= 1u << k;
              r = 1;  // error rows hold T::one(), like the reference builders
            }
            EX_REG(I.dst, k) = (uint64_t)(u128)r;
            if (I.o_wide) EX_REG(I.dst + 1, k) = (uint64_t)((u128)r >> 64);
          }
#pragma unroll
          for (int k = 0; k < ROWS; ++k)
            if ((bad >> k) & 1u) EX_RAISE(k);
        } else {
#pragma unroll
          for (int k = 0; k < ROWS; ++k) {
            const i128 av = EX_RD128(I.a, I.a_wide, acls, k), bv = EX_RD128(I.b, I.b_wide, bcls, k);
            i128 r;
            if (!dec_row_nodiv(*D, av, bv, I.a_dec != 0, I.b_dec != 0, t128, &r)) {
              EX_RAISE(k);
              r = 1;  // error rows hold T::one(), like the reference builders
            }
            EX_REG(I.dst, k) = (uint64_t)(u128)r;
            if (I.o_wide) EX_REG(I.dst + 1, k) = (uint64_t)((u128)r >> 64);
          }
        }
      } break;
      case EX_EQ: case EX_NOTEQ: case EX_LT: case EX_LTE: case EX_GT: case EX_GTE:
#pragma unroll
        for (int k = 0; k < ROWS; ++k) {
          int c3;
          if (I.a_wide) {  // Decimal128 (same scale on both sides: the planner casts)
            const i128 av = EX_RD128(I.a, 1, CLS_SIGNED, k), bv = EX_RD128(I.b, 1, CLS_SIGNED, k);
            c3 = (av > bv) - (av < bv);
          } else {
            c3 = ex_cmp3(EX_REG(I.a, k), EX_REG(I.b, k), acls);
          }
          bool r;
          switch (I.op) {
            case EX_EQ: r = c3 == 0; break;
            case EX_NOTEQ: r = c3 != 0; break;
            case EX_LT: r = c3 < 0; break;
            case EX_LTE: r = c3 <= 0; break;
            case EX_GT: r = c3 > 0; break;
            default: r = c3 >= 0; break;
          }
          EX_REG(I.dst, k) = (uint64_t)r;
        }
        break;
      case EX_AND: EX_ROWS_DO(x & y & 1) break;
      case EX_OR: EX_ROWS_DO((x | y) & 1) break;
      case EX_NOT: EX_ROWS_DO1((x ^ 1) & 1) break;
      case EX_IS_TRUE:  // decode_predicate: TRUE only where the operand is valid and set (imm = the nullable inputs it depends on)
#pragma unroll
        for (int k = 0; k < ROWS; ++k)
          EX_REG(I.dst, k) = (EX_REG(I.a, k) & 1) & (uint64_t)((vmask[k] & (uint32_t)I.imm) == (uint32_t)I.imm);
        break;
      case EX_IF:  // if(cond, then, else) over values that cannot raise (checked on the host): a select
#pragma unroll
        for (int k = 0; k < ROWS; ++k) {
          const bool t = EX_REG(I.a, k) & 1;
          const uint64_t lo = t ? EX_REG(I.b, k) : EX_REG(I.c, k);
          uint64_t hi = 0;
          if (I.o_wide) hi = t ? EX_REG(I.b + 1, k) : EX_REG(I.c + 1, k);
          EX_REG(I.dst, k) = lo;
          if (I.o_wide) EX_REG(I.dst + 1, k) = hi;
        }
        break;
      default:  // EX_CAST (lossless widenings only, checked on the host)
        if (I.o_wide) {
#pragma unroll
          for (int k = 0; k < ROWS; ++k) {
            const i128 v = EX_RD128(I.a, I.a_wide, acls, k);
            EX_REG(I.dst, k) = (uint64_t)(u128)v;
            EX_REG(I.dst + 1, k) = (uint64_t)((u128)v >> 64);
          }
        } else if (ocls == CLS_FLOAT) EX_ROWS_DO1(ex_norm((uint64_t)__double_as_longlong(ex_to_f64(x, acls)), I))
        else EX_ROWS_DO1(x)
        break;
    }
#undef EX_ROWS_DO
#undef EX_ROWS_DO1
  }
#undef EX_RD128
#undef EX_RAISE
#undef EX_REG
}

// ---- host side (k_expr.hip) -------------------------------------------------------------------------------------
struct ExRoot {          // one result the caller wants to read after the program ran
  int32_t reg;           // in: user register (>= 0), or -(1 + input column) to expose an input column as it is
  int32_t slot;          // out: LDS slot (lo; hi = slot + 1 for 128-bit values)
  int32_t type, precision, scale, wide;
  uint32_t dep;          // out: nullable inputs it depends on
};
// Checks and compiles a dbhip_expr_ins program: types every node (ResultTypeOfBinary / decimal result sizes), compiles
// LOADs away, allocates LDS slots by liveness (the roots stay live to the end). `n_filter_ins` leading instructions
// compute the filter root `filter_root` (index into roots, -1 = none). Returns a dbhip status.
int32_t dbhip_expr_compile_internal(const dbhip_expr_ins* prog_host, int32_t n_ins, const dbhip_col* inputs_host, int32_t n_inputs,
                                    ExRoot* roots, int32_t n_roots, int32_t filter_root, ExProg* out, bool* out_may_raise,
                                    bool* out_any_nullable);

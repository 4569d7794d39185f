// k_expr.hip — fused expression evaluation (SURVEY §8 a1 + §8f rank 2).
//
// Reference: Evaluator::run walks the Expr tree and materialises ONE column per call node
// (src/query/expression/src/evaluator.rs:229-464; CompoundBlockOperator / BlockOperator::Map,
// src/query/sql/src/evaluator/block_operator.rs:42-85): sum(a + b*c) is three kernels moving 56 B/row where
// 24 B/row are needed. Here a whole expression tree is ONE launch: the host flattens the tree (post-order) into
// a short register program, the kernel interprets it with a wave-uniform instruction stream — every lane of
// every wave executes the same instruction, so the interpreter costs scalar-unit work only — over a register
// file that lives in LDS ([register][row slot][lane] u64: conflict-free ds_read/ds_write_b64), two row slots
// per lane (rows base + lane and base + 64 + lane: every column access is a fully coalesced wave load and a
// Boolean result is exactly one ballot word). Intermediate columns never exist in HBM.
//
// Semantics per node are the per-node kernels' (k_arith.hip / k_cmp_filter.hip), which restate
// numeric_basic_arithmetic.rs:255-427 and comparison.rs:98-112: operands are cast `as` the node's result type
// (arithmetics_type.rs), integer arithmetic wraps at the result width, `/` is f64 with the "divided by zero" row
// error (only for rows whose inputs are all valid, function.rs:534-556), floats compare as OrderedFloat. NULLs:
// passthrough_nullable (register_vectorize.rs:447-471) — the payload is computed for all rows, the result's validity
// is the AND of the validity of every nullable input column the program loads.
#include "dev_common.h"
#include "dev_load.h"
#include "runtime.h"

#include <string.h>

using namespace dbhip;

namespace {

constexpr int EX_MAX_INS = 32;
constexpr int EX_MAX_REGS = 8;
constexpr int EX_MAX_INPUTS = 8;
constexpr int EX_ROWS = 2;  // row slots per lane

// Everything the interpreter needs is decoded on the host: the scalar unit is shared by the CU's four SIMDs, and a
// type switch per operand per instruction made the first version SALU bound (483 scalar instructions per 128 rows).
struct ExIns {
  int16_t op, dst, a, b;   // a = input index for LOAD
  int16_t type;            // result type (host side checks only)
  int8_t acls, bcls, ocls; // CLS_SIGNED / CLS_UNSIGNED / CLS_FLOAT of the operands and the result
  int8_t norm_sh;          // result width: shift that sign-/zero-extends from the result's bits (0 for 64-bit)
  int8_t norm_signed, norm_f32;
  uint64_t imm;
};

struct ExProg {
  ExIns ins[EX_MAX_INS];
  const void* in_data[EX_MAX_INPUTS];
  const uint8_t* in_valid[EX_MAX_INPUTS];
  int64_t in_voff[EX_MAX_INPUTS];
  int32_t in_type[EX_MAX_INPUTS];   // load kind, see ex_load
  int32_t in_scalar[EX_MAX_INPUTS];
  int32_t n_ins, n_inputs, out_reg, out_type, n_slots;
  int32_t in_slot[EX_MAX_INPUTS];   // LDS register of input column c (-1: the program never reads it)
  int32_t out_kind, out_cls;        // store width in bytes (0 = bitmap, -4 = f32), class of the result
  int64_t n;
  void* out_values;              // numeric: elements of out_type; BOOL: bitmap words
  uint64_t* out_validity;        // bitmap words (may be NULL)
  uint32_t* err_words;           // preset to all ones (may be NULL)
  unsigned long long* err_count; // may be NULL
  unsigned long long* sum_out;   // may be NULL: accumulate the sum of the valid rows of out_reg
};

// load kinds (host: ex_load_kind): the common 8-byte case is the first test
enum { LK_8 = 0, LK_S4 = 1, LK_U4 = 2, LK_F4 = 3, LK_S2 = 4, LK_U2 = 5, LK_S1 = 6, LK_U1 = 7, LK_BOOL = 8 };
__device__ __forceinline__ uint64_t ex_load(const void* p, int kind, int64_t i) {
  if (kind == LK_8) return ((const uint64_t*)p)[i];
  if (kind == LK_S4) return (uint64_t)(int64_t)((const int32_t*)p)[i];
  if (kind == LK_U4) return ((const uint32_t*)p)[i];
  if (kind == LK_F4) return (uint64_t)__double_as_longlong((double)((const float*)p)[i]);
  if (kind == LK_S2) return (uint64_t)(int64_t)((const int16_t*)p)[i];
  if (kind == LK_U2) return ((const uint16_t*)p)[i];
  if (kind == LK_S1) return (uint64_t)(int64_t)((const int8_t*)p)[i];
  if (kind == LK_U1) return ((const uint8_t*)p)[i];
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
  EX_AND = 12, EX_OR = 13, EX_NOT = 14, EX_CAST = 15
};

// Register numbering inside the kernel: 0 .. n_temps-1 are temporaries, n_temps + c is input column c. The host
// compiles LOAD instructions away (an operand that names a loaded register is rewritten to the input register), so
// that ALL column loads of a chunk are issued back to back before the first instruction is interpreted — with the
// loads inside the interpreter loop every LOAD would cost a full HBM round trip of its own.
// NIN: compile-time bound of the input count (unused slots cost neither code nor VGPRs); ALL8: every input is a
// plain 8-byte non-scalar column (i64 / u64 / f64 / timestamp / decimal64) — straight-line loads, no kind tests.
template <int NIN, bool ALL8, int ROWS>
__global__ __launch_bounds__(256) void expr_kernel(ExProg P) {
  extern __shared__ uint64_t ex_regs[];  // [n_slots][ROWS][256]
  const int tid = threadIdx.x, lane = tid & 63;
  const int64_t rows_per_wave = 64 * ROWS;
  const int64_t nchunks = (P.n + rows_per_wave - 1) / rows_per_wave;
  const int64_t wave_global = ((int64_t)blockIdx.x * blockDim.x + tid) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  const int out_cls = P.out_cls;
  uint64_t acc_i = 0;
  double acc_f = 0.0;
#define EX_REG(r, k) ex_regs[((r) * ROWS + (k)) * 256 + tid]

  for (int64_t c = wave_global; c < nchunks; c += nwaves) {
    const int64_t base = c * rows_per_wave;
    int64_t row[ROWS];
    bool in_range[ROWS], valid[ROWS];
#pragma unroll
    for (int k = 0; k < ROWS; ++k) {
      row[k] = base + 64 * k + lane;
      in_range[k] = row[k] < P.n;
      valid[k] = in_range[k];
      if (!in_range[k]) row[k] = P.n - 1;  // clamp: loads stay in bounds, results are masked
    }
    // ---- all input loads of the chunk, back to back ----
    uint64_t in[NIN][ROWS];
#pragma unroll
    for (int ci = 0; ci < NIN; ++ci) {
      if (ci < P.n_inputs) {
#pragma unroll
        for (int k = 0; k < ROWS; ++k) {
          if (ALL8) {
            in[ci][k] = ((const uint64_t*)P.in_data[ci])[row[k]];
            if (P.in_valid[ci]) valid[k] = valid[k] && bit_get(P.in_valid[ci], P.in_voff[ci] + row[k]);
          } else {
            const int64_t j = P.in_scalar[ci] ? 0 : row[k];
            in[ci][k] = ex_load(P.in_data[ci], P.in_type[ci], j);
            if (P.in_valid[ci]) valid[k] = valid[k] && bit_get(P.in_valid[ci], P.in_voff[ci] + j);
          }
        }
      }
    }
#pragma unroll
    for (int ci = 0; ci < NIN; ++ci) {
      if (ci < P.n_inputs) {
#pragma unroll
        for (int k = 0; k < ROWS; ++k)
          if (P.in_slot[ci] >= 0) EX_REG(P.in_slot[ci], k) = in[ci][k];
      }
    }
    // ---- interpret (wave-uniform instruction stream): ONE dispatch per instruction, the row slots loop inside the case ----
    for (int pc = 0; pc < P.n_ins; ++pc) {
      const ExIns I = P.ins[pc];
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
      switch (I.op) {
        case EX_CONST:
#pragma unroll
          for (int k = 0; k < ROWS; ++k) EX_REG(I.dst, k) = I.imm;
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
            if (bb == 0.0) {
              if (valid[k]) {  // NULL rows never raise (function.rs:536-543); padding rows are not valid
                if (P.err_words) atomicAnd(&P.err_words[row[k] >> 5], ~(1u << (row[k] & 31)));
                if (P.err_count) atomicAdd(P.err_count, 1ULL);
              }
            } else {
              r = (uint64_t)__double_as_longlong(a / bb);
            }
            EX_REG(I.dst, k) = r;
          }
          break;
        case EX_EQ: EX_ROWS_DO((uint64_t)(ex_cmp3(x, y, acls) == 0)) break;
        case EX_NOTEQ: EX_ROWS_DO((uint64_t)(ex_cmp3(x, y, acls) != 0)) break;
        case EX_LT: EX_ROWS_DO((uint64_t)(ex_cmp3(x, y, acls) < 0)) break;
        case EX_LTE: EX_ROWS_DO((uint64_t)(ex_cmp3(x, y, acls) <= 0)) break;
        case EX_GT: EX_ROWS_DO((uint64_t)(ex_cmp3(x, y, acls) > 0)) break;
        case EX_GTE: EX_ROWS_DO((uint64_t)(ex_cmp3(x, y, acls) >= 0)) break;
        case EX_AND: EX_ROWS_DO(x & y & 1) break;
        case EX_OR: EX_ROWS_DO((x | y) & 1) break;
        case EX_NOT: EX_ROWS_DO1((x ^ 1) & 1) break;
        default:  // EX_CAST (lossless widenings only, checked on the host)
          if (ocls == CLS_FLOAT) EX_ROWS_DO1(ex_norm((uint64_t)__double_as_longlong(ex_to_f64(x, acls)), I))
          else EX_ROWS_DO1(x)
          break;
      }
#undef EX_ROWS_DO
#undef EX_ROWS_DO1
    }
    // ---- result ----
#pragma unroll
    for (int k = 0; k < ROWS; ++k) {
      const uint64_t r = EX_REG(P.out_reg, k);
      const int64_t word = (base >> 6) + k;  // 64-row word of this slot
      if (P.out_values) {
        if (P.out_kind == 0) {
          const uint64_t m = __ballot(in_range[k] && (r & 1));
          if (lane == 0 && base + 64 * k < P.n) ((uint64_t*)P.out_values)[word] = m;
        } else if (in_range[k]) {
          if (P.out_kind == 8) ((uint64_t*)P.out_values)[row[k]] = r;
          else if (P.out_kind == 4) ((uint32_t*)P.out_values)[row[k]] = (uint32_t)r;
          else if (P.out_kind == -4) ((float*)P.out_values)[row[k]] = (float)__longlong_as_double((long long)r);
          else if (P.out_kind == 2) ((uint16_t*)P.out_values)[row[k]] = (uint16_t)r;
          else ((uint8_t*)P.out_values)[row[k]] = (uint8_t)r;
        }
      }
      if (P.out_validity) {
        const uint64_t m = __ballot(valid[k]);
        if (lane == 0 && base + 64 * k < P.n) P.out_validity[word] = m;
      }
      if (P.sum_out && valid[k]) {
        if (out_cls == CLS_FLOAT) acc_f += __longlong_as_double((long long)r);
        else acc_i += r;
      }
    }
  }
#undef EX_REG
  if (P.sum_out) {
    if (out_cls == CLS_FLOAT) {
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) acc_f += __shfl_xor(acc_f, off, 64);
      if (lane == 0 && acc_f != 0.0) atomicAdd((double*)P.sum_out, acc_f);
    } else {
      acc_i = wave_sum_u64(acc_i);
      if (lane == 0 && acc_i) atomicAdd(P.sum_out, (unsigned long long)acc_i);
    }
  }
}

bool ex_numeric(int t) { return type_class(t) >= 0; }

int ex_cls(int t) { return t == DBHIP_T_BOOL ? CLS_UNSIGNED : type_class(t); }

int ex_load_kind(int t) {
  switch (t) {
    case DBHIP_T_BOOL: return LK_BOOL;
    case DBHIP_T_I8: return LK_S1;
    case DBHIP_T_U8: return LK_U1;
    case DBHIP_T_I16: return LK_S2;
    case DBHIP_T_U16: return LK_U2;
    case DBHIP_T_I32: case DBHIP_T_DATE: return LK_S4;
    case DBHIP_T_U32: return LK_U4;
    case DBHIP_T_F32: return LK_F4;
    default: return LK_8;
  }
}

void ex_decode(ExIns& d, int ta, int tb) {
  d.acls = (int8_t)(ta >= 0 ? ex_cls(ta) : 0);
  d.bcls = (int8_t)(tb >= 0 ? ex_cls(tb) : 0);
  d.ocls = (int8_t)ex_cls(d.type);
  d.norm_f32 = d.type == DBHIP_T_F32;
  d.norm_signed = d.ocls == CLS_SIGNED;
  const int bits = d.type == DBHIP_T_BOOL ? 64 : type_bits(d.type);
  d.norm_sh = (int8_t)((d.ocls == CLS_FLOAT) ? 0 : 64 - bits);
}

bool ex_lossless_cast(int from, int to) {
  if (from == to) return true;
  const int fc = type_class(from), tc = type_class(to);
  const int fb = type_bits(from), tb = type_bits(to);
  if (fc == CLS_FLOAT) return tc == CLS_FLOAT && tb >= fb;
  if (tc == CLS_FLOAT) return to == DBHIP_T_F64 ? fb <= 32 : fb <= 16;  // exactly representable
  if (fc == CLS_UNSIGNED) return tb > fb || (tc == CLS_UNSIGNED && tb == fb);
  return tc == CLS_SIGNED && tb >= fb;
}

}  // namespace

extern "C" {

int32_t dbhip_expr_eval(const dbhip_expr_ins* prog_host, int32_t n_ins, const dbhip_col* inputs_host, int32_t n_inputs,
                        int64_t n, int32_t out_reg, void* out_values, uint8_t* out_validity, uint8_t* err_bitmap,
                        uint64_t* err_count_dev, void* sum_out_dev, void* stream) {
  DBHIP_REQUIRE(prog_host && n_ins >= 1 && n_ins <= EX_MAX_INS, "dbhip_expr_eval: 1..32 instructions");
  DBHIP_REQUIRE(n_inputs >= 0 && n_inputs <= EX_MAX_INPUTS && (inputs_host || n_inputs == 0), "dbhip_expr_eval: 0..8 input columns");
  DBHIP_REQUIRE(out_reg >= 0 && out_reg < EX_MAX_REGS && n >= 0, "dbhip_expr_eval: bad out register / n");
  ExProg P;
  memset(&P, 0, sizeof(P));
  int reg_type[EX_MAX_REGS];
  int reg_loc[EX_MAX_REGS];  // where the register's current value lives: temp r, or EX_MAX_REGS + input column
  int n_out = 0;             // instructions kept (LOADs are compiled away)
  for (int r = 0; r < EX_MAX_REGS; ++r) { reg_type[r] = -1; reg_loc[r] = r; }
  bool any_nullable = false, may_raise = false;
  for (int c = 0; c < n_inputs; ++c) {
    const dbhip_col& col = inputs_host[c];
    if (!(ex_numeric(col.type) || col.type == DBHIP_T_BOOL)) {
      set_error("dbhip_expr_eval: input %d has type %d (numeric, date, timestamp, decimal64-as-i64 and boolean columns only)", c, col.type);
      return DBHIP_ERR_UNSUPPORTED;
    }
    DBHIP_REQUIRE(col.data || n == 0, "dbhip_expr_eval: NULL input column");
    P.in_data[c] = col.data; P.in_valid[c] = col.validity; P.in_voff[c] = col.validity_offset;
    P.in_type[c] = ex_load_kind(col.type); P.in_scalar[c] = col.is_scalar;
  }
  for (int i = 0; i < n_ins; ++i) {
    const dbhip_expr_ins& s = prog_host[i];
    ExIns& d = P.ins[n_out];
    if (s.dst < 0 || s.dst >= EX_MAX_REGS) { set_error("dbhip_expr_eval: instruction %d: register %d out of range (0..7)", i, s.dst); return DBHIP_ERR_INVALID; }
    d.op = (int16_t)s.op; d.dst = (int16_t)s.dst; d.type = (int16_t)s.type; d.imm = s.imm;
    d.a = (int16_t)((s.a >= 0 && s.a < EX_MAX_REGS) ? reg_loc[s.a] : 0);
    d.b = (int16_t)((s.b >= 0 && s.b < EX_MAX_REGS) ? reg_loc[s.b] : 0);
    auto src = [&](int r) -> int { return (r >= 0 && r < EX_MAX_REGS) ? reg_type[r] : -1; };
    switch (s.op) {
      case DBHIP_EX_LOAD:
        if (s.a < 0 || s.a >= n_inputs || inputs_host[s.a].type != s.type) { set_error("dbhip_expr_eval: instruction %d: LOAD of input %d as type %d", i, s.a, s.type); return DBHIP_ERR_INVALID; }
        any_nullable |= inputs_host[s.a].validity != nullptr;
        reg_type[s.dst] = s.type;
        reg_loc[s.dst] = EX_MAX_REGS + s.a;
        continue;  // no instruction: the value is read straight from the input register
      case DBHIP_EX_CONST:
        if (!(ex_numeric(s.type) || s.type == DBHIP_T_BOOL)) { set_error("dbhip_expr_eval: instruction %d: CONST of type %d", i, s.type); return DBHIP_ERR_INVALID; }
        ex_decode(d, -1, -1);
        break;
      case DBHIP_EX_PLUS: case DBHIP_EX_MINUS: case DBHIP_EX_MULTIPLY: case DBHIP_EX_DIVIDE: {
        const int ta = src(s.a), tb = src(s.b);
        const int aop = s.op == DBHIP_EX_PLUS ? DBHIP_OP_PLUS : s.op == DBHIP_EX_MINUS ? DBHIP_OP_MINUS : s.op == DBHIP_EX_MULTIPLY ? DBHIP_OP_MULTIPLY : DBHIP_OP_DIVIDE;
        if (ta < 0 || tb < 0 || !ex_numeric(ta) || !ex_numeric(tb) || dbhip_arith_result_type(aop, ta, tb) != s.type) {
          set_error("dbhip_expr_eval: instruction %d: op %d on types (%d,%d) does not yield type %d (arithmetics_type.rs)", i, s.op, ta, tb, s.type);
          return DBHIP_ERR_INVALID;
        }
        ex_decode(d, ta, tb);
        may_raise |= s.op == DBHIP_EX_DIVIDE;
      } break;
      case DBHIP_EX_EQ: case DBHIP_EX_NOTEQ: case DBHIP_EX_LT: case DBHIP_EX_LTE: case DBHIP_EX_GT: case DBHIP_EX_GTE: {
        const int ta = src(s.a), tb = src(s.b);
        if (ta < 0 || ta != tb || s.type != DBHIP_T_BOOL) { set_error("dbhip_expr_eval: instruction %d: comparison needs equal operand types (%d,%d) and a Boolean result", i, ta, tb); return DBHIP_ERR_INVALID; }
        ex_decode(d, ta, tb);
      } break;
      case DBHIP_EX_AND: case DBHIP_EX_OR: case DBHIP_EX_NOT: {
        const int ta = src(s.a), tb = s.op == DBHIP_EX_NOT ? DBHIP_T_BOOL : src(s.b);
        if (ta != DBHIP_T_BOOL || tb != DBHIP_T_BOOL || s.type != DBHIP_T_BOOL) { set_error("dbhip_expr_eval: instruction %d: Boolean operator on non-Boolean registers", i); return DBHIP_ERR_INVALID; }
        ex_decode(d, DBHIP_T_BOOL, DBHIP_T_BOOL);
      } break;
      case DBHIP_EX_CAST: {
        const int ta = src(s.a);
        if (ta < 0 || !ex_numeric(ta) || !ex_numeric(s.type)) { set_error("dbhip_expr_eval: instruction %d: CAST %d -> %d", i, ta, s.type); return DBHIP_ERR_INVALID; }
        if (!ex_lossless_cast(ta, s.type)) { set_error("dbhip_expr_eval: CAST %d -> %d can overflow: keep the checked CPU cast", ta, s.type); return DBHIP_ERR_UNSUPPORTED; }
        ex_decode(d, ta, -1);
      } break;
      default:
        set_error("dbhip_expr_eval: instruction %d: unknown op %d", i, s.op);
        return DBHIP_ERR_INVALID;
    }
    reg_type[s.dst] = s.type;
    reg_loc[s.dst] = s.dst;
    ++n_out;
  }
  if (reg_type[out_reg] < 0) { set_error("dbhip_expr_eval: out register %d is never written", out_reg); return DBHIP_ERR_INVALID; }
  DBHIP_REQUIRE(out_values || sum_out_dev, "dbhip_expr_eval: neither an output column nor a sum was asked for");
  DBHIP_REQUIRE(!sum_out_dev || ex_numeric(reg_type[out_reg]), "dbhip_expr_eval: sum needs a numeric result");
  DBHIP_REQUIRE(!any_nullable || out_validity || sum_out_dev, "dbhip_expr_eval: nullable inputs need out_validity");
  hipStream_t s = resolve_stream(stream);
  if (err_bitmap) DBHIP_CHECK(hipMemsetAsync(err_bitmap, 0xFF, (size_t)ceil_div(n, 32) * 4, s));
  if (n == 0) return DBHIP_OK;
  P.n_ins = n_out; P.n_inputs = n_inputs; P.out_type = reg_type[out_reg]; P.n = n;
  P.out_values = out_values; P.out_validity = (uint64_t*)out_validity;
  P.err_words = may_raise ? (uint32_t*)err_bitmap : nullptr;
  P.err_count = may_raise ? (unsigned long long*)err_count_dev : nullptr;
  P.sum_out = (unsigned long long*)sum_out_dev;
  static const bool force2 = getenv("DBHIP_EXPR_ROWS2") != nullptr;
  kernel_timer_start(s);
  // LDS register allocation. A LOCATION is a user temporary (0..7) or an input column (EX_MAX_REGS + c); its value is live
  // from its definition to its last read before the next definition (the out register's last value lives to the end).
  // Slots are handed out in one forward walk; an operand that dies in an instruction frees its slot BEFORE the
  // destination is placed, so results overwrite dead operands in place (every thread reads its own cells of a and b before
  // it writes dst). a + b * c needs 3 slots instead of 5: the LDS footprint per wave is what bounds this kernel's
  // occupancy, and with it how many loads are in flight.
  {
    constexpr int NLOC = EX_MAX_REGS + EX_MAX_INPUTS;
    const int out_loc = reg_loc[out_reg];
    int slot_of[NLOC];
    bool used[NLOC];
    for (int l = 0; l < NLOC; ++l) { slot_of[l] = -1; used[l] = false; }
    int n_slots = 0;
    auto take = [&]() { for (int q = 0; q < NLOC; ++q) if (!used[q]) { used[q] = true; if (q + 1 > n_slots) n_slots = q + 1; return q; } return -1; };
    auto reads = [&](const ExIns& I, int loc) {
      if (I.op == DBHIP_EX_CONST) return false;
      if (I.a == loc) return true;
      return I.op != DBHIP_EX_NOT && I.op != DBHIP_EX_CAST && I.b == loc;
    };
    // is the value that location `loc` holds right after instruction i still read later?
    auto live_after = [&](int i, int loc) {
      for (int j = i + 1; j < n_out; ++j) {
        if (reads(P.ins[j], loc)) return true;
        if (P.ins[j].dst == loc) return false;
      }
      return loc == out_loc;
    };
    // input columns the program (or the result) reads get their slots first: the kernel fills them at the top of every chunk
    for (int c = 0; c < n_inputs; ++c) {
      const int loc = EX_MAX_REGS + c;
      P.in_slot[c] = -1;
      bool any = loc == out_loc;
      for (int j = 0; j < n_out && !any; ++j) any = reads(P.ins[j], loc);
      if (any) { slot_of[loc] = take(); P.in_slot[c] = slot_of[loc]; }
    }
    for (int i = 0; i < n_out; ++i) {
      ExIns& I = P.ins[i];
      const int la = I.a, lb = I.b, ld = I.dst;
      const bool ra = reads(I, la), rb = reads(I, lb) && lb != la;
      const int sa = ra ? slot_of[la] : 0, sb = (reads(I, lb)) ? slot_of[lb] : 0;
      if (ra && !live_after(i, la) && la != ld) { used[slot_of[la]] = false; slot_of[la] = -1; }
      if (rb && !live_after(i, lb) && lb != ld) { used[slot_of[lb]] = false; slot_of[lb] = -1; }
      if (slot_of[ld] >= 0) { used[slot_of[ld]] = false; slot_of[ld] = -1; }  // the old value of dst ends here (read above if it was an operand)
      slot_of[ld] = take();
      I.a = (int16_t)(sa < 0 ? 0 : sa); I.b = (int16_t)(sb < 0 ? 0 : sb); I.dst = (int16_t)slot_of[ld];
    }
    P.out_reg = slot_of[out_loc];
    P.n_slots = n_slots;
  }
  P.out_cls = ex_cls(P.out_type);
  switch (P.out_type) {
    case DBHIP_T_BOOL: P.out_kind = 0; break;
    case DBHIP_T_F32: P.out_kind = -4; break;
    default: P.out_kind = type_bits(P.out_type) / 8; break;
  }
  // row slots per lane: 4 (32 B per operand per lane in flight, half the per-row interpreter overhead) while the LDS register
  // file allows it, else 2
  int rows_per_lane = EX_ROWS;
  if (!force2 && (size_t)P.n_slots * 4 * 256 * 8 <= 64 * 1024) rows_per_lane = 4;
  const size_t lds = (size_t)(P.n_slots > 0 ? P.n_slots : 1) * rows_per_lane * 256 * 8;
  if (lds > 64 * 1024) {
    set_error("dbhip_expr_eval: %d live registers exceed the LDS register file; split the expression", P.n_slots);
    return DBHIP_ERR_UNSUPPORTED;
  }
  // workgroups: a power of two (r01ze sweep over 128 M rows of a + b * c: 1024 0.75 ms, 2048 0.76, 3072 0.78 — but 1280 1.01 and
  // 1536 0.91: the waves' common stride through the three input arrays wants to be a power of two); 8 rows per lane: 0.97 ms
  const int64_t chunks = ceil_div(n, 64 * rows_per_lane);
  const int grid = grid_for(ceil_div(chunks, 4) * 256, 256, 1024);
  bool all8 = true;
  for (int c = 0; c < n_inputs; ++c) all8 &= P.in_type[c] == LK_8 && !P.in_scalar[c];
  const dim3 g(grid), b(256);
#define EX_LAUNCH(NIN_, R_)                                                                   \
  do {                                                                                        \
    if (all8) hipLaunchKernelGGL((expr_kernel<NIN_, true, R_>), g, b, lds, s, P);              \
    else hipLaunchKernelGGL((expr_kernel<NIN_, false, R_>), g, b, lds, s, P);                  \
  } while (0)
  if (rows_per_lane == 4) {
    if (n_inputs <= 2) EX_LAUNCH(2, 4); else if (n_inputs <= 4) EX_LAUNCH(4, 4); else EX_LAUNCH(8, 4);
  } else {
    if (n_inputs <= 2) EX_LAUNCH(2, 2); else if (n_inputs <= 4) EX_LAUNCH(4, 2); else EX_LAUNCH(8, 2);
  }
#undef EX_LAUNCH
  kernel_timer_stop(s);
  DBHIP_LAUNCH_CHECK();
  return DBHIP_OK;
}

}  // extern "C"

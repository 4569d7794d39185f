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
#include "dev_expr.h"
#include "runtime.h"

#include <string.h>

using namespace dbhip;

namespace {

struct ExOut {
  int32_t out_slot, out_kind, out_cls;  // store width in bytes (0 = bitmap, -4 = f32, 16 = i128), class of the result
  uint32_t out_dep;                     // nullable inputs the result depends on (validity = all of them valid)
  int64_t n;
  void* out_values;              // numeric: elements of out_type; BOOL: bitmap words
  uint64_t* out_validity;        // bitmap words (may be NULL)
  unsigned long long* sum_out;   // may be NULL: accumulate the sum of the valid rows of the result
};

// Register numbering inside the kernel: LDS slots. The host compiles LOAD instructions away (an operand that names a
// loaded register is rewritten to the input column's slot), so that ALL column loads of a chunk are issued back to back
// before the first instruction is interpreted — with the loads inside the interpreter loop every LOAD would cost a full
// HBM round trip of its own.
// NIN: compile-time bound of the input count (unused slots cost neither code nor VGPRs); ALL8: every input is a
// plain 8-byte non-scalar column (i64 / u64 / f64 / timestamp / decimal64) — straight-line loads, no kind tests.
template <int NIN, bool ALL8, int ROWS, bool DIV = false>
__global__ __launch_bounds__(256) void expr_kernel(ExProg P, ExOut O) {
  extern __shared__ uint64_t ex_regs[];  // [n_slots][ROWS][256]
  const int tid = threadIdx.x, lane = tid & 63;
  const int64_t rows_per_wave = 64 * ROWS;
  const int64_t nchunks = (O.n + rows_per_wave - 1) / rows_per_wave;
  const int64_t wave_global = ((int64_t)blockIdx.x * blockDim.x + tid) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  const int out_cls = O.out_cls;
  uint64_t acc_i = 0;
  double acc_f = 0.0;
#define EX_REG(r, k) ex_regs[((r) * ROWS + (k)) * 256 + tid]

  for (int64_t c = wave_global; c < nchunks; c += nwaves) {
    const int64_t base = c * rows_per_wave;
    int64_t row[ROWS];
    bool in_range[ROWS];
    uint32_t vmask[ROWS];
#pragma unroll
    for (int k = 0; k < ROWS; ++k) {
      row[k] = base + 64 * k + lane;
      in_range[k] = row[k] < O.n;
      vmask[k] = 0xFFu;
      if (!in_range[k]) row[k] = O.n - 1;  // clamp: loads stay in bounds, results are masked
    }
    // ---- all input loads of the chunk, back to back ----
    uint64_t in[NIN][ROWS], hi0[ROWS], hi1[ROWS];  // hi words of the (at most two) 128-bit input columns
#pragma unroll
    for (int k = 0; k < ROWS; ++k) { hi0[k] = 0; hi1[k] = 0; }
#pragma unroll
    for (int ci = 0; ci < NIN; ++ci) {
      if (ci < P.n_inputs) {
#pragma unroll
        for (int k = 0; k < ROWS; ++k) {
          if (ALL8) {
            in[ci][k] = ((const uint64_t*)P.in_data[ci])[row[k]];
            if (P.in_valid[ci] && !bit_get(P.in_valid[ci], P.in_voff[ci] + row[k])) vmask[k] &= ~(1u << ci);
          } else {
            const int64_t j = P.in_scalar[ci] ? 0 : row[k];
            in[ci][k] = ex_load(P.in_data[ci], P.in_type[ci], j);
            if (P.in_wide_ord[ci] == 0) hi0[k] = ((const uint64_t*)P.in_data[ci])[2 * j + 1];
            else if (P.in_wide_ord[ci] == 1) hi1[k] = ((const uint64_t*)P.in_data[ci])[2 * j + 1];
            if (P.in_valid[ci] && !bit_get(P.in_valid[ci], P.in_voff[ci] + j)) vmask[k] &= ~(1u << ci);
          }
        }
      }
    }
#pragma unroll
    for (int ci = 0; ci < NIN; ++ci) {
      if (ci < P.n_inputs) {
#pragma unroll
        for (int k = 0; k < ROWS; ++k)
          if (P.in_slot[ci] >= 0) {
            EX_REG(P.in_slot[ci], k) = in[ci][k];
            if (!ALL8 && P.in_wide_ord[ci] >= 0) EX_REG(P.in_slot[ci] + 1, k) = P.in_wide_ord[ci] == 0 ? hi0[k] : hi1[k];
          }
      }
    }
    // ---- interpret (wave-uniform instruction stream): ONE dispatch per instruction, the row slots loop inside the case ----
    ex_interpret<ROWS, DIV>(P, P, ex_regs, tid, 0, P.n_ins, row, in_range, vmask);
    // ---- result ----
#pragma unroll
    for (int k = 0; k < ROWS; ++k) {
      const uint64_t r = EX_REG(O.out_slot, k);
      const bool valid = in_range[k] && ((vmask[k] & O.out_dep) == O.out_dep);
      const int64_t word = (base >> 6) + k;  // 64-row word of this slot
      if (O.out_values) {
        if (O.out_kind == 0) {
          const uint64_t m = __ballot(in_range[k] && (r & 1));
          if (lane == 0 && base + 64 * k < O.n) ((uint64_t*)O.out_values)[word] = m;
        } else if (in_range[k]) {
          if (O.out_kind == 8) ((uint64_t*)O.out_values)[row[k]] = r;
          else if (O.out_kind == 16) { ((uint64_t*)O.out_values)[2 * row[k]] = r; ((uint64_t*)O.out_values)[2 * row[k] + 1] = EX_REG(O.out_slot + 1, k); }
          else if (O.out_kind == 4) ((uint32_t*)O.out_values)[row[k]] = (uint32_t)r;
          else if (O.out_kind == -4) ((float*)O.out_values)[row[k]] = (float)__longlong_as_double((long long)r);
          else if (O.out_kind == 2) ((uint16_t*)O.out_values)[row[k]] = (uint16_t)r;
          else ((uint8_t*)O.out_values)[row[k]] = (uint8_t)r;
        }
      }
      if (O.out_validity) {
        const uint64_t m = __ballot(valid);
        if (lane == 0 && base + 64 * k < O.n) O.out_validity[word] = m;
      }
      if (O.sum_out && valid) {
        if (out_cls == CLS_FLOAT) acc_f += __longlong_as_double((long long)r);
        else acc_i += r;
      }
    }
  }
#undef EX_REG
  if (O.sum_out) {
    if (out_cls == CLS_FLOAT) {
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) acc_f += __shfl_xor(acc_f, off, 64);
      if (lane == 0 && acc_f != 0.0) atomicAdd((double*)O.sum_out, acc_f);
    } else {
      acc_i = wave_sum_u64(acc_i);
      if (lane == 0 && acc_i) atomicAdd(O.sum_out, (unsigned long long)acc_i);
    }
  }
}

bool ex_numeric(int t) { return type_class(t) >= 0; }
bool ex_decimal(int t) { return t == DBHIP_T_DEC64 || t == DBHIP_T_DEC128; }

int ex_cls(int t) { return (t == DBHIP_T_BOOL) ? CLS_UNSIGNED : (t == DBHIP_T_DEC128 ? CLS_SIGNED : type_class(t)); }

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
    case DBHIP_T_DEC128: return LK_16;
    default: return LK_8;
  }
}

void ex_decode(ExIns& d, int ta, int tb) {
  d.acls = (int8_t)(ta >= 0 ? ex_cls(ta) : 0);
  d.bcls = (int8_t)(tb >= 0 ? ex_cls(tb) : 0);
  d.ocls = (int8_t)ex_cls(d.type);
  d.norm_f32 = d.type == DBHIP_T_F32;
  d.norm_signed = d.ocls == CLS_SIGNED;
  const int bits = (d.type == DBHIP_T_BOOL || d.type == DBHIP_T_DEC128) ? 64 : type_bits(d.type);
  d.norm_sh = (int8_t)((d.ocls == CLS_FLOAT) ? 0 : 64 - bits);
  d.a_wide = ta == DBHIP_T_DEC128;
  d.b_wide = tb == DBHIP_T_DEC128;
  d.o_wide = d.type == DBHIP_T_DEC128;
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

struct RegInfo {
  int type = -1, precision = 0, scale = 0;
  int loc = 0;        // where the register's current value lives: temp r, or EX_MAX_REGS + input column
  uint32_t dep = 0;   // nullable inputs the value depends on
  bool may_raise = false;  // some node below it can raise a row error
  bool is_const = false;   // the register holds a CONST (its 64-bit image in cimm)
  uint64_t cimm = 0;
  // The register is an AND over a nullable operand, evaluated STRICTLY (NULL as soon as one operand is NULL). The reference's
  // `and` is three-valued (FALSE AND NULL = FALSE); strict evaluation agrees with it exactly where NULL and FALSE are not told
  // apart: as the filter, or as an operand of another AND that ends in the filter. Anywhere else the program is refused.
  bool strict_and = false;
};

}  // namespace

int32_t dbhip_expr_compile_internal(const dbhip_expr_ins* prog_host, int32_t n_ins, const dbhip_col* inputs_host, int32_t n_inputs,
                                    ExRoot* roots, int32_t n_roots, int32_t filter_root, ExProg* out, bool* out_may_raise,
                                    bool* out_any_nullable) {
  DBHIP_REQUIRE(n_ins >= 0 && n_ins <= 2 * EX_MAX_INS && (prog_host || n_ins == 0), "expression program: too many instructions");
  DBHIP_REQUIRE(n_inputs >= 0 && n_inputs <= EX_MAX_INPUTS && (inputs_host || n_inputs == 0), "expression program: 0..8 input columns");
  DBHIP_REQUIRE(n_roots >= 1 && n_roots <= EX_MAX_ROOTS && roots, "expression program: 1..8 results");
  ExProg& P = *out;
  memset(&P, 0, sizeof(P));
  RegInfo reg[EX_MAX_REGS];
  for (int r = 0; r < EX_MAX_REGS; ++r) reg[r].loc = r;
  int n_out = 0, n_dec = 0, n_wide = 0, n_hidden = 0;
  constexpr int EX_HIDDEN = 8;   // hidden locations: decimal constants converted on the host
  bool any_nullable = false, may_raise = false;
  for (int c = 0; c < n_inputs; ++c) {
    const dbhip_col& col = inputs_host[c];
    if (!(ex_numeric(col.type) || col.type == DBHIP_T_BOOL || col.type == DBHIP_T_DEC128)) {
      set_error("expression program: input %d has type %d (numeric, date, timestamp, decimal and boolean columns only)", c, col.type);
      return DBHIP_ERR_UNSUPPORTED;
    }
    DBHIP_REQUIRE(col.data, "expression program: NULL input column");
    P.in_data[c] = col.data; P.in_valid[c] = col.validity; P.in_voff[c] = col.validity_offset;
    P.in_type[c] = ex_load_kind(col.type); P.in_scalar[c] = col.is_scalar;
    P.in_wide_ord[c] = -1;
    P.in_has_valid[c] = col.validity != nullptr;
    if (col.type == DBHIP_T_DEC128) {
      if (n_wide >= 2) { set_error("expression program: more than two Decimal128 input columns"); return DBHIP_ERR_UNSUPPORTED; }
      P.in_wide_ord[c] = n_wide++;
    }
  }
  int filter_reg = filter_root >= 0 ? roots[filter_root].reg : -1;
  int n_filter_ins_src = -1;  // source instruction after which the filter register holds its final value
  if (filter_reg >= 0) {
    for (int i = 0; i < n_ins; ++i) if (prog_host[i].dst == filter_reg) n_filter_ins_src = i + 1;
  }
  for (int i = 0; i < n_ins; ++i) {
    const dbhip_expr_ins& s = prog_host[i];
    if (n_out >= EX_MAX_INS) { set_error("expression program: more than %d instructions after LOAD elimination", EX_MAX_INS); return DBHIP_ERR_UNSUPPORTED; }
    ExIns& d = P.ins[n_out];
    memset(&d, 0, sizeof(d));
    if (s.dst < 0 || s.dst >= EX_MAX_REGS) { set_error("expression program: instruction %d: register %d out of range (0..%d)", i, s.dst, EX_MAX_REGS - 1); return DBHIP_ERR_INVALID; }
    const uint8_t s_prec = s.precision, s_scale = s.scale;
    d.op = (int16_t)s.op; d.dst = (int16_t)s.dst; d.type = (int16_t)s.type; d.imm = s.imm; d.dec_idx = -1;
    auto ok_reg = [&](int r) { return r >= 0 && r < EX_MAX_REGS && reg[r].type >= 0; };
    d.a = (int16_t)(ok_reg(s.a) ? reg[s.a].loc : 0);
    d.b = (int16_t)(ok_reg(s.b) ? reg[s.b].loc : 0);
    RegInfo res;
    res.type = s.type; res.loc = s.dst;
    if (s.op != DBHIP_EX_AND && s.op != DBHIP_EX_IS_TRUE && s.op != DBHIP_EX_LOAD && s.op != DBHIP_EX_CONST) {
      const int rc3 = s.op == DBHIP_EX_IF ? (int)(s.imm & 0xFF) : -1;
      const bool unary1 = s.op == DBHIP_EX_NOT || s.op == DBHIP_EX_CAST;
      if ((ok_reg(s.a) && reg[s.a].strict_and) || (!unary1 && ok_reg(s.b) && reg[s.b].strict_and) || (rc3 >= 0 && ok_reg(rc3) && reg[rc3].strict_and)) {
        set_error("expression program: instruction %d consumes an AND over a nullable operand; only the filter (or another AND) may — "
                  "the reference's and / or are three-valued (evaluator.rs:284-305)", i);
        return DBHIP_ERR_UNSUPPORTED;
      }
    }
    switch (s.op) {
      case DBHIP_EX_LOAD: {
        if (s.a < 0 || s.a >= n_inputs || inputs_host[s.a].type != s.type) { set_error("expression program: instruction %d: LOAD of input %d as type %d", i, s.a, s.type); return DBHIP_ERR_INVALID; }
        RegInfo& r = reg[s.dst];
        r.type = s.type; r.precision = inputs_host[s.a].precision; r.scale = inputs_host[s.a].scale;
        r.loc = EX_MAX_REGS + s.a;
        r.dep = inputs_host[s.a].validity ? (1u << s.a) : 0;
        r.may_raise = false;
        any_nullable |= inputs_host[s.a].validity != nullptr;
        continue;  // no instruction: the value is read straight from the input's slot
      }
      case DBHIP_EX_CONST:
        if (!(ex_numeric(s.type) || s.type == DBHIP_T_BOOL)) { set_error("expression program: instruction %d: CONST of type %d (Decimal128 constants: widen a Decimal64 one with CAST)", i, s.type); return DBHIP_ERR_INVALID; }
        ex_decode(d, -1, -1);
        res.precision = s_prec; res.scale = s_scale;
        res.is_const = true; res.cimm = s.imm;
        break;
      case DBHIP_EX_PLUS: case DBHIP_EX_MINUS: case DBHIP_EX_MULTIPLY: case DBHIP_EX_DIVIDE: {
        if (!ok_reg(s.a) || !ok_reg(s.b)) { set_error("expression program: instruction %d reads an unset register", i); return DBHIP_ERR_INVALID; }
        const RegInfo &ra = reg[s.a], &rb = reg[s.b];
        const int aop = s.op == DBHIP_EX_PLUS ? DBHIP_OP_PLUS : s.op == DBHIP_EX_MINUS ? DBHIP_OP_MINUS : s.op == DBHIP_EX_MULTIPLY ? DBHIP_OP_MULTIPLY : DBHIP_OP_DIVIDE;
        if (ex_decimal(ra.type) || ex_decimal(rb.type)) {
          // binary_decimal (decimal/src/arithmetic.rs:190-316); an integer operand is converted like other_to_decimal
          if (n_dec >= EX_MAX_DEC) { set_error("expression program: more than %d decimal nodes", EX_MAX_DEC); return DBHIP_ERR_UNSUPPORTED; }
          if ((!ex_decimal(ra.type) && (type_class(ra.type) == CLS_FLOAT || type_class(ra.type) < 0)) ||
              (!ex_decimal(rb.type) && (type_class(rb.type) == CLS_FLOAT || type_class(rb.type) < 0))) {
            set_error("expression program: instruction %d: decimal arithmetic with a non-integer operand (%d,%d)", i, ra.type, rb.type);
            return DBHIP_ERR_UNSUPPORTED;
          }
          int ot, op_, os_;
          int32_t rc = dbhip_decimal_decode_internal(aop, ra.type, ra.precision, ra.scale, rb.type, rb.precision, rb.scale, &P.dec[n_dec], &ot, &op_, &os_);
          if (rc) return rc;
          if (ot != s.type || (s_prec && (s_prec != op_ || s_scale != os_))) {
            set_error("expression program: instruction %d: decimal op %d yields type %d Decimal(%d,%d), the node says type %d Decimal(%d,%d)", i, s.op, ot, op_, os_, s.type, s_prec, s_scale);
            return DBHIP_ERR_INVALID;
          }
          DecOp& D = P.dec[n_dec];
          ex_decode(d, ra.type, rb.type);
          d.op = EX_DEC; d.dec_idx = (int8_t)n_dec++;
          d.a_dec = ex_decimal(ra.type); d.b_dec = ex_decimal(rb.type);
          // A CONSTANT operand is brought to its bound size here, once (binary_decimal converts a Value::Scalar argument
          // once per block, not per row): `1 - l_discount` becomes `100 - l_discount` over two values of the bound type.
          const bool t128 = D.t_is_128 != 0;
          for (int side = 0; side < 2; ++side) {
            const RegInfo& ro = side == 0 ? ra : rb;
            const bool o_dec = ex_decimal(ro.type);
            const bool pass = side == 0 ? (o_dec ? !D.a_check : D.a_to_scale == 0) : (o_dec ? !D.b_check : D.b_to_scale == 0);
            if (!ro.is_const || pass || ro.type == DBHIP_T_DEC128 || n_hidden >= EX_HIDDEN || n_out + 1 >= EX_MAX_INS) continue;
            const int cls = ex_cls(ro.type);
            const i128 x = cls == CLS_SIGNED ? (i128)(int64_t)ro.cimm : (i128)(u128)ro.cimm;
            i128 conv;
            const bool okc = side == 0 ? convert_operand(x, o_dec, D.a_from_scale, D.a_to_scale, D.a_to_precision, D.a_check, t128, &conv)
                                       : convert_operand(x, o_dec, D.b_from_scale, D.b_to_scale, D.b_to_precision, D.b_check, t128, &conv);
            if (!okc) continue;  // the conversion overflows: leave it to the kernel, which raises for the live rows
            // a CONST into a hidden location in front of the node (this slot of P.ins is the node's: shift it by one)
            const ExIns node = P.ins[n_out];
            ExIns& cst = P.ins[n_out];
            memset(&cst, 0, sizeof(cst));
            const int hloc = EX_MAX_REGS + EX_MAX_INPUTS + n_hidden++;
            cst.op = EX_CONST; cst.dst = (int16_t)hloc; cst.type = (int16_t)(t128 ? DBHIP_T_DEC128 : DBHIP_T_DEC64); cst.dec_idx = -1;
            cst.imm = (uint64_t)(u128)conv; cst.imm_hi = (uint64_t)((u128)conv >> 64);
            ex_decode(cst, -1, -1);
            ++n_out;
            ExIns& nd = P.ins[n_out];
            nd = node;
            if (side == 0) { nd.a = (int16_t)hloc; nd.a_dec = 1; nd.a_wide = t128; nd.acls = CLS_SIGNED; D.a_check = 0; D.a_from_scale = D.a_to_scale; }
            else { nd.b = (int16_t)hloc; nd.b_dec = 1; nd.b_wide = t128; nd.bcls = CLS_SIGNED; D.b_check = 0; D.b_from_scale = D.b_to_scale; }
          }
          {
            ExIns& nd = P.ins[n_out];
            const bool pass_a = nd.a_dec ? !D.a_check : D.a_to_scale == 0, pass_b = nd.b_dec ? !D.b_check : D.b_to_scale == 0;
            D.trivial = pass_a && pass_b && !dec_op_needs_division(D) && (((D.op == DBHIP_OP_PLUS || D.op == DBHIP_OP_MINUS) && !D.overflow) || D.op == DBHIP_OP_MULTIPLY);
            res.may_raise = !D.trivial;
          }
          res.precision = op_; res.scale = os_;
        } else {
          if (!ex_numeric(ra.type) || !ex_numeric(rb.type) || dbhip_arith_result_type(aop, ra.type, rb.type) != s.type) {
            set_error("expression program: instruction %d: op %d on types (%d,%d) does not yield type %d (arithmetics_type.rs)", i, s.op, ra.type, rb.type, s.type);
            return DBHIP_ERR_INVALID;
          }
          ex_decode(d, ra.type, rb.type);
          res.may_raise = s.op == DBHIP_EX_DIVIDE;
        }
        res.dep = ra.dep | rb.dep;
        res.may_raise |= ra.may_raise | rb.may_raise;
      } break;
      case DBHIP_EX_EQ: case DBHIP_EX_NOTEQ: case DBHIP_EX_LT: case DBHIP_EX_LTE: case DBHIP_EX_GT: case DBHIP_EX_GTE: {
        if (!ok_reg(s.a) || !ok_reg(s.b)) { set_error("expression program: instruction %d reads an unset register", i); return DBHIP_ERR_INVALID; }
        const RegInfo &ra = reg[s.a], &rb = reg[s.b];
        if (ra.type != rb.type || s.type != DBHIP_T_BOOL || (ex_decimal(ra.type) && ra.scale != rb.scale)) {
          set_error("expression program: instruction %d: comparison needs equal operand types (%d,%d; decimals: equal scales) and a Boolean result", i, ra.type, rb.type);
          return DBHIP_ERR_INVALID;
        }
        ex_decode(d, ra.type, rb.type);
        res.dep = ra.dep | rb.dep; res.may_raise = ra.may_raise | rb.may_raise;
      } break;
      case DBHIP_EX_AND: case DBHIP_EX_OR: case DBHIP_EX_NOT: {
        const bool unary = s.op == DBHIP_EX_NOT;
        if (!ok_reg(s.a) || (!unary && !ok_reg(s.b)) || reg[s.a].type != DBHIP_T_BOOL || (!unary && reg[s.b].type != DBHIP_T_BOOL) || s.type != DBHIP_T_BOOL) {
          set_error("expression program: instruction %d: Boolean operator on non-Boolean registers", i);
          return DBHIP_ERR_INVALID;
        }
        ex_decode(d, DBHIP_T_BOOL, DBHIP_T_BOOL);
        res.dep = reg[s.a].dep | (unary ? 0 : reg[s.b].dep);
        res.may_raise = reg[s.a].may_raise | (unary ? false : reg[s.b].may_raise);
        if (s.op == DBHIP_EX_OR && res.dep) {  // TRUE OR NULL = TRUE: a strict OR would drop / null such rows
          set_error("expression program: instruction %d: OR over a nullable operand is three-valued in the reference (or_filters, evaluator.rs:301-303); not fused", i);
          return DBHIP_ERR_UNSUPPORTED;
        }
        if (s.op == DBHIP_EX_AND) res.strict_and = res.dep != 0;
      } break;
      case DBHIP_EX_IS_TRUE: {
        // decode_predicate (NULL -> FALSE): the operand's validity is folded into the value, the result depends on no nullable
        // input any more. A strict AND below it is exact here (FALSE AND NULL and TRUE AND NULL both decode to FALSE).
        if (!ok_reg(s.a) || reg[s.a].type != DBHIP_T_BOOL || s.type != DBHIP_T_BOOL) {
          set_error("expression program: instruction %d: IS_TRUE needs a Boolean register and a Boolean result", i);
          return DBHIP_ERR_INVALID;
        }
        ex_decode(d, DBHIP_T_BOOL, -1);
        d.imm = reg[s.a].dep;
        res.dep = 0; res.may_raise = reg[s.a].may_raise;
      } break;
      case DBHIP_EX_CAST: {
        if (!ok_reg(s.a)) { set_error("expression program: instruction %d reads an unset register", i); return DBHIP_ERR_INVALID; }
        const RegInfo& ra = reg[s.a];
        if (s.type == DBHIP_T_DEC128 || ra.type == DBHIP_T_DEC128 || (ex_decimal(s.type) != ex_decimal(ra.type))) {
          // decimal widening keeps the scale (decimal_expand_cast faster path, cast.rs:901-979): Decimal64 -> Decimal128 only
          if (!(ra.type == DBHIP_T_DEC64 && s.type == DBHIP_T_DEC128 && (!s_prec || (s_scale == ra.scale && s_prec >= ra.precision)))) {
            set_error("expression program: CAST %d -> %d: only Decimal64 -> Decimal128 at the same scale is fused; keep the checked CPU cast", ra.type, s.type);
            return DBHIP_ERR_UNSUPPORTED;
          }
          res.precision = s_prec ? s_prec : 38; res.scale = ra.scale;
        } else {
          if (!ex_numeric(ra.type) || !ex_numeric(s.type)) { set_error("expression program: instruction %d: CAST %d -> %d", i, ra.type, s.type); return DBHIP_ERR_INVALID; }
          if (!ex_lossless_cast(ra.type, s.type)) { set_error("expression program: CAST %d -> %d can overflow: keep the checked CPU cast", ra.type, s.type); return DBHIP_ERR_UNSUPPORTED; }
          res.precision = ra.precision; res.scale = ra.scale;
        }
        ex_decode(d, ra.type, -1);
        res.dep = ra.dep; res.may_raise = ra.may_raise;
      } break;
      case DBHIP_EX_IF: {
        // if(cond, then, else) (evaluator.rs:284-305 evaluates the branches under the condition's validity, so an error in
        // the branch a row does not take is never raised): fused only when neither branch can raise — then it is a select
        const int rc_ = (int)(s.imm & 0xFF);
        if (!ok_reg(s.a) || !ok_reg(s.b) || !ok_reg(rc_) || reg[s.a].type != DBHIP_T_BOOL) { set_error("expression program: instruction %d: if(cond, then, else) needs a Boolean condition and two set registers", i); return DBHIP_ERR_INVALID; }
        const RegInfo &rt = reg[s.b], &re = reg[rc_];
        if (rt.type != re.type || rt.type != s.type || (ex_decimal(rt.type) && (rt.scale != re.scale))) { set_error("expression program: instruction %d: the branches of if() must have the node's type (%d,%d -> %d)", i, rt.type, re.type, s.type); return DBHIP_ERR_INVALID; }
        if (rt.may_raise || re.may_raise) { set_error("expression program: a branch of if() can raise a row error; keep the CPU evaluator's lazy branches"); return DBHIP_ERR_UNSUPPORTED; }
        if (reg[s.a].dep) { set_error("expression program: if() on a nullable condition is not fused"); return DBHIP_ERR_UNSUPPORTED; }
        ex_decode(d, rt.type, re.type);
        d.a = (int16_t)reg[s.a].loc; d.b = (int16_t)rt.loc; d.c = (int16_t)re.loc;
        d.imm = 0;
        res.precision = rt.precision > re.precision ? rt.precision : re.precision; res.scale = rt.scale;
        res.dep = rt.dep | re.dep;
      } break;
      default:
        set_error("expression program: instruction %d: unknown op %d", i, s.op);
        return DBHIP_ERR_INVALID;
    }
    P.ins[n_out].dep = (uint8_t)res.dep;   // (a folded constant may have shifted the node by one slot)
    may_raise |= res.may_raise;
    reg[s.dst] = res;
    ++n_out;
    if (i + 1 == n_filter_ins_src) P.n_filter_ins = n_out;
  }
  // roots
  constexpr int NLOC = EX_MAX_REGS + EX_MAX_INPUTS + EX_HIDDEN;
  int root_loc[EX_MAX_ROOTS];
  for (int r = 0; r < n_roots; ++r) {
    ExRoot& R = roots[r];
    if (R.reg >= 0) {
      if (R.reg >= EX_MAX_REGS || reg[R.reg].type < 0) { set_error("expression program: result register %d is never written", R.reg); return DBHIP_ERR_INVALID; }
      const RegInfo& ri = reg[R.reg];
      if (ri.strict_and && r != filter_root) {
        set_error("expression program: result register %d is an AND over a nullable operand (FALSE AND NULL = FALSE in the reference); only the filter may be", R.reg);
        return DBHIP_ERR_UNSUPPORTED;
      }
      root_loc[r] = ri.loc; R.type = ri.type; R.precision = ri.precision; R.scale = ri.scale; R.dep = ri.dep;
    } else {
      const int c = -R.reg - 1;
      if (c >= n_inputs) { set_error("expression program: result names input column %d of %d", c, n_inputs); return DBHIP_ERR_INVALID; }
      root_loc[r] = EX_MAX_REGS + c; R.type = inputs_host[c].type; R.precision = inputs_host[c].precision; R.scale = inputs_host[c].scale;
      R.dep = inputs_host[c].validity ? (1u << c) : 0;
      any_nullable |= inputs_host[c].validity != nullptr;
    }
    R.wide = R.type == DBHIP_T_DEC128;
  }
  if (filter_root >= 0 && roots[filter_root].type != DBHIP_T_BOOL) { set_error("expression program: the filter must be Boolean"); return DBHIP_ERR_INVALID; }
  // LDS slot allocation. A LOCATION is a user temporary (0..EX_MAX_REGS-1) or an input column (EX_MAX_REGS + c); its value is
  // live from its definition to its last read before the next definition (a root's last value lives to the end). Slots
  // are handed out in one forward walk; an operand that dies in an instruction frees its slot(s) BEFORE the destination is
  // placed, so results overwrite dead operands in place (every thread reads its own cells of a, b, c before it writes
  // dst). 128-bit locations take two adjacent slots. The LDS footprint per wave is what bounds the kernels' occupancy.
  {
    int slot_of[NLOC], width[NLOC];
    bool used[EX_MAX_SLOTS + 2];
    for (int l = 0; l < NLOC; ++l) { slot_of[l] = -1; width[l] = 1; }
    for (int q = 0; q < EX_MAX_SLOTS + 2; ++q) used[q] = false;
    int n_slots = 0;
    auto take = [&](int w) {
      for (int q = 0; q + w <= EX_MAX_SLOTS; ++q) {
        bool free_ = !used[q] && (w == 1 || !used[q + 1]);
        if (free_) { used[q] = true; if (w == 2) used[q + 1] = true; if (q + w > n_slots) n_slots = q + w; return q; }
      }
      return -1;
    };
    auto release = [&](int loc) { if (slot_of[loc] >= 0) { used[slot_of[loc]] = false; if (width[loc] == 2) used[slot_of[loc] + 1] = false; slot_of[loc] = -1; } };
    auto reads = [&](const ExIns& I, int loc) {
      if (I.op == EX_CONST) return false;
      if (I.a == loc) return true;
      if (I.op == EX_IF && I.c == loc) return true;
      return I.op != EX_NOT && I.op != EX_CAST && I.b == loc;
    };
    auto is_root = [&](int loc) { for (int r = 0; r < n_roots; ++r) if (root_loc[r] == loc) return true; return false; };
    auto live_after = [&](int i, int loc) {
      for (int j = i + 1; j < n_out; ++j) {
        if (reads(P.ins[j], loc)) return true;
        if (P.ins[j].dst == loc) return false;
      }
      return is_root(loc);
    };
    bool fail = false;
    for (int c = 0; c < n_inputs; ++c) {
      const int loc = EX_MAX_REGS + c;
      width[loc] = inputs_host[c].type == DBHIP_T_DEC128 ? 2 : 1;
      P.in_slot[c] = -1;
      bool any = is_root(loc);
      for (int j = 0; j < n_out && !any; ++j) any = reads(P.ins[j], loc);
      if (any) { slot_of[loc] = take(width[loc]); P.in_slot[c] = slot_of[loc]; fail |= slot_of[loc] < 0; }
    }
    for (int i = 0; i < n_out && !fail; ++i) {
      ExIns& I = P.ins[i];
      const int la = I.a, lb = I.b, lc = I.c, ld = I.dst;
      const bool ra = reads(I, la), rb = I.op != EX_CONST && I.op != EX_NOT && I.op != EX_CAST && reads(I, lb), rc3 = I.op == EX_IF;
      const int sa = ra ? slot_of[la] : 0, sb = rb ? slot_of[lb] : 0, sc = rc3 ? slot_of[lc] : 0;
      if (ra && !live_after(i, la) && la != ld) release(la);
      if (rb && lb != la && !live_after(i, lb) && lb != ld) release(lb);
      if (rc3 && lc != la && lc != lb && !live_after(i, lc) && lc != ld) release(lc);
      release(ld);  // the old value of dst ends here (read above if it was an operand)
      width[ld] = I.o_wide ? 2 : 1;
      slot_of[ld] = take(width[ld]);
      fail |= slot_of[ld] < 0;
      I.a = (int16_t)(sa < 0 ? 0 : sa); I.b = (int16_t)(sb < 0 ? 0 : sb); I.c = (int16_t)(sc < 0 ? 0 : sc); I.dst = (int16_t)slot_of[ld];
    }
    if (fail) { set_error("expression program: more than %d live LDS slots; split the expression", EX_MAX_SLOTS); return DBHIP_ERR_UNSUPPORTED; }
    for (int r = 0; r < n_roots; ++r) roots[r].slot = slot_of[root_loc[r]];
    P.n_slots = n_slots;
  }
  P.n_ins = n_out; P.n_inputs = n_inputs;
  P.filter_slot = filter_root >= 0 ? roots[filter_root].slot : -1;
  P.filter_dep = filter_root >= 0 ? roots[filter_root].dep : 0;
  if (filter_root < 0) P.n_filter_ins = 0;
  if (out_may_raise) *out_may_raise = may_raise;
  if (out_any_nullable) *out_any_nullable = any_nullable;
  return DBHIP_OK;
}

extern "C" {

int32_t dbhip_expr_eval(const dbhip_expr_ins* prog_host, int32_t n_ins, const dbhip_col* inputs_host, int32_t n_inputs,
                        int64_t n, int32_t out_reg, void* out_values, uint8_t* out_validity, uint8_t* err_bitmap,
                        uint64_t* err_count_dev, void* sum_out_dev, void* stream) {
  DBHIP_REQUIRE(prog_host && n_ins >= 1, "dbhip_expr_eval: empty program");
  DBHIP_REQUIRE(out_reg >= 0 && out_reg < EX_MAX_REGS && n >= 0, "dbhip_expr_eval: bad out register / n");
  if (n == 0) {
    hipStream_t s0 = resolve_stream(stream);
    (void)s0;
  }
  for (int c = 0; c < n_inputs; ++c) DBHIP_REQUIRE(inputs_host[c].data || n == 0, "dbhip_expr_eval: NULL input column");
  ExProg P;
  ExRoot root;
  memset(&root, 0, sizeof(root));
  root.reg = out_reg;
  bool may_raise = false, any_nullable = false;
  // (n == 0: inputs may carry NULL data pointers; give the checker a harmless address)
  dbhip_col tmp_in[EX_MAX_INPUTS];
  for (int c = 0; c < n_inputs && c < EX_MAX_INPUTS; ++c) { tmp_in[c] = inputs_host[c]; if (!tmp_in[c].data) tmp_in[c].data = &tmp_in[c]; }
  int32_t rc = dbhip_expr_compile_internal(prog_host, n_ins, n_inputs ? tmp_in : nullptr, n_inputs, &root, 1, -1, &P, &may_raise, &any_nullable);
  if (rc) return rc;
  DBHIP_REQUIRE(out_values || sum_out_dev, "dbhip_expr_eval: neither an output column nor a sum was asked for");
  DBHIP_REQUIRE(!sum_out_dev || (ex_numeric(root.type) && root.type != DBHIP_T_DEC128), "dbhip_expr_eval: sum needs a numeric (<= 64-bit) result");
  DBHIP_REQUIRE(!any_nullable || out_validity || sum_out_dev, "dbhip_expr_eval: nullable inputs need out_validity");
  hipStream_t s = resolve_stream(stream);
  if (err_bitmap) DBHIP_CHECK(hipMemsetAsync(err_bitmap, 0xFF, (size_t)ceil_div(n, 32) * 4, s));
  if (n == 0) return DBHIP_OK;
  P.err_words = may_raise ? (uint32_t*)err_bitmap : nullptr;
  P.err_count = may_raise ? (unsigned long long*)err_count_dev : nullptr;
  ExOut O;
  memset(&O, 0, sizeof(O));
  O.n = n; O.out_values = out_values; O.out_validity = (uint64_t*)out_validity; O.sum_out = (unsigned long long*)sum_out_dev;
  O.out_slot = root.slot; O.out_dep = root.dep; O.out_cls = ex_cls(root.type);
  switch (root.type) {
    case DBHIP_T_BOOL: O.out_kind = 0; break;
    case DBHIP_T_F32: O.out_kind = -4; break;
    case DBHIP_T_DEC128: O.out_kind = 16; break;
    default: O.out_kind = type_bits(root.type) / 8; break;
  }
  static const bool force2 = exp_env("DBHIP_EXPR_ROWS2") != nullptr;
  kernel_timer_start(s);
  // row slots per lane: 4 (32 B per operand per lane in flight, half the per-row interpreter overhead) while the LDS register
  // file allows it, else 2
  // a program with a rounding decimal multiply / divide runs the one instantiation that carries the long divisions
  bool has_div = false;
  for (int i = 0; i < P.n_ins; ++i) has_div |= P.ins[i].op == EX_DEC && dec_op_needs_division(P.dec[P.ins[i].dec_idx]);
  int rows_per_lane = 2;
  if (!force2 && !has_div && (size_t)P.n_slots * 4 * 256 * 8 <= 64 * 1024) rows_per_lane = 4;
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
    if (all8) hipLaunchKernelGGL((expr_kernel<NIN_, true, R_>), g, b, lds, s, P, O);           \
    else hipLaunchKernelGGL((expr_kernel<NIN_, false, R_>), g, b, lds, s, P, O);               \
  } while (0)
  if (has_div) {
    hipLaunchKernelGGL((expr_kernel<8, false, 2, true>), g, b, lds, s, P, O);
  } else if (rows_per_lane == 4) {
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

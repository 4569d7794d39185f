// dev_decimal.h — one decimal +,-,*,/ on one row, shared by the column kernel (k_decimal.hip) and the fused expression
// interpreter (dev_expr.h). Reference semantics:
//   result size / operand sizes   decimal/src/arithmetic.rs:80-139 (ArithmeticOp::result_size)
//   operand conversion            :141-153 convert_to_decimal -> cast.rs:701-753 (integer_to_decimal),
//                                 cast.rs:901-979 (decimal_expand_cast), :1036-1049
//   compute                       :190-316 binary_decimal in T = storage class of the result precision
//   rounding                      types/decimal.rs:759-797 (i64), :1024-1060 (i128 via i256)
#pragma once
#include "dev_common.h"

// everything binary_decimal needs besides the two values, decoded on the host (wave-uniform in the kernels)
struct DecOp {
  int op;             // DBHIP_OP_PLUS / MINUS / MULTIPLY / DIVIDE
  // conversion of each operand into its bound size
  int a_from_scale, a_to_scale, a_to_precision, a_check;   // check: range-check after rescale
  int b_from_scale, b_to_scale, b_to_precision, b_check;
  int t_is_128;       // compute type T: 0 = i64, 1 = i128
  int ret_precision, ret_scale;
  int overflow;       // return precision == T::MAX_PRECISION
  int scale_mul;      // multiply: sa+sb-sr ; divide: sb+sr-sa
  int trivial;        // (fused interpreter) both operands arrive at their bound sizes and the op is a plain wrapping
                      // +, - or * in T with nothing to check: computed inline, no call
};

__host__ __device__ __forceinline__ i128 wrap_T(i128 v, bool t128) { return t128 ? v : (i128)(int64_t)v; }

__host__ __device__ __forceinline__ i128 max_for_precision(int p) { return pow10_i128(p) - 1; }

// checked multiply in T (i64 or i128)
__host__ __device__ __forceinline__ bool checked_mul_T(i128 x, i128 f, bool t128, i128* out) {
  if (!t128) {
    i128 r = x * f;  // both fit in i64 -> exact in i128
    if (r > (i128)INT64_MAX || r < (i128)INT64_MIN) return false;
    *out = r;
    return true;
  }
  bool neg = (x < 0) != (f < 0);
  u128 ax = x < 0 ? (u128)0 - (u128)x : (u128)x;
  u128 af = f < 0 ? (u128)0 - (u128)f : (u128)f;
  u256 pr = u256_mul_128(ax, af);
  u128 lim = neg ? ((u128)1 << 127) : (((u128)1 << 127) - 1);
  if (pr.hi != 0 || pr.lo > lim) return false;
  *out = neg ? (i128)((u128)0 - pr.lo) : (i128)pr.lo;
  return true;
}

// Brings an operand to its bound (precision, scale) in T. Returns false on "Decimal overflow".
__host__ __device__ __forceinline__ bool convert_operand(i128 x, bool is_decimal, int from_scale, int to_scale,
                                                int to_precision, int check, bool t128, i128* out) {
  if (!is_decimal) {
    // integer_to_decimal (cast.rs:701-753): scale 0 never checks
    if (to_scale == 0) {
      *out = wrap_T(x, t128);
      return true;
    }
    i128 xt = x;
    if (!t128 && (x > (i128)INT64_MAX || x < (i128)INT64_MIN)) return false;  // T::from_i128
    i128 r;
    if (!checked_mul_T(xt, pow10_i128(to_scale), t128, &r)) return false;
    i128 mx = max_for_precision(to_precision);
    if (r > mx || r < -mx) return false;
    *out = r;
    return true;
  }
  if (!check) {  // same scale: passthrough (decimal_expand_cast faster path), then `as T`
    *out = wrap_T(x, t128);
    return true;
  }
  i128 r;
  if (!checked_mul_T(wrap_T(x, t128), pow10_i128(to_scale - from_scale), t128, &r)) return false;
  i128 mx = max_for_precision(to_precision);
  if (r > mx || r < -mx) return false;
  *out = r;
  return true;
}

// num / d truncated toward zero like the `/` of i128 (what the reference's `/` on i64 / i128 does); divisors that fit 64 bits — every
// power of ten up to 10^19, every Decimal64 divisor — take the two-digit division of dev_common.h instead of the 128-step loop the
// compiler expands a 128-bit `/` into
__host__ __device__ __forceinline__ i128 sdiv128(i128 num, i128 d) {
  const bool neg = (num < 0) != (d < 0);
  const u128 un = num < 0 ? (u128)0 - (u128)num : (u128)num;
  const u128 ud = d < 0 ? (u128)0 - (u128)d : (u128)d;
  u128 q;
  if ((uint64_t)(ud >> 64) == 0) q = udiv128_by_64(un, (uint64_t)ud, nullptr);
  else q = un / ud;
  return neg ? (i128)((u128)0 - q) : (i128)q;
}

// i128 path of do_round_mul with overflow (decimal.rs:1040-1054): 256-bit intermediate
__device__ bool round_mul_128_overflow(i128 a, i128 b, int shift, i128* out) {
  bool neg = (a < 0) != (b < 0);
  u128 A = a < 0 ? (u128)0 - (u128)a : (u128)a;
  u128 B = b < 0 ? (u128)0 - (u128)b : (u128)b;
  u128 div = (u128)pow10_i128(shift);
  u256 pr = u256_add_128(u256_mul_128(A, B), div / 2);
  u256 q = u256_div_128(pr, div, nullptr);
  u128 lim = neg ? ((u128)1 << 127) : (((u128)1 << 127) - 1);
  if (q.hi != 0 || q.lo > lim) return false;
  *out = neg ? (i128)((u128)0 - q.lo) : (i128)q.lo;
  return true;
}

// i128 do_round_div (decimal.rs:1056-1064): low 128 bits of the i256 quotient
__device__ i128 round_div_128(i128 a, i128 b, int mul_scale) {
  bool neg = (a < 0) != (b < 0);
  u128 A = a < 0 ? (u128)0 - (u128)a : (u128)a;
  u128 B = b < 0 ? (u128)0 - (u128)b : (u128)b;
  // 10^mul_scale may exceed u128 for very large scales; the supported range is <= 38
  u256 num = u256_add_128(u256_mul_128(A, (u128)pow10_i128(mul_scale)), B / 2);
  u256 q = u256_div_128(num, B, nullptr);
  return neg ? (i128)((u128)0 - q.lo) : (i128)q.lo;
}

// one row: operands -> bound sizes -> op in T; false = the row raises (value 1 is stored, like the reference builders)
__device__ __forceinline__ bool dec_row(const DecOp& p, i128 av, i128 bv, bool a_dec, bool b_dec, bool t128, i128* out) {
  i128 a, b, r = 1;
    bool ok = convert_operand(av, a_dec, p.a_from_scale,
                              p.a_to_scale, p.a_to_precision, p.a_check, t128, &a);
    ok = convert_operand(bv, b_dec, p.b_from_scale,
                         p.b_to_scale, p.b_to_precision, p.b_check, t128, &b) && ok;
    if (ok) {
      switch (p.op) {
        case DBHIP_OP_PLUS:
        case DBHIP_OP_MINUS: {
          i128 t = p.op == DBHIP_OP_PLUS ? (i128)((u128)a + (u128)b) : (i128)((u128)a - (u128)b);
          t = wrap_T(t, t128);
          if (p.overflow) {
            i128 mx = max_for_precision(p.ret_precision);
            if (t < -mx || t > mx) ok = false;
          }
          r = t;
        } break;
        case DBHIP_OP_MULTIPLY: {
          if (p.scale_mul == 0) {
            r = wrap_T((i128)((u128)a * (u128)b), t128);
          } else if (!t128) {
            i128 div = pow10_i128(p.scale_mul);
            if (!p.overflow) {
              // (self*rhs +- div/2)/div in wrapping i64
              int64_t prod = (int64_t)((uint64_t)(int64_t)a * (uint64_t)(int64_t)b);
              int64_t d = (int64_t)div;
              int64_t num = ((a < 0) == (b < 0)) ? (int64_t)((uint64_t)prod + (uint64_t)(d / 2))
                                                 : (int64_t)((uint64_t)prod - (uint64_t)(d / 2));
              r = (i128)(num / d);
            } else {
              i128 num = ((a < 0) == (b < 0)) ? a * b + div / 2 : a * b - div / 2;
              i128 res = sdiv128(num, div);
              i128 mx = max_for_precision(18);  // i64::DECIMAL_MAX
              if (res < -mx || res > mx) ok = false;
              r = res;
            }
          } else {
            if (!p.overflow) {
              i128 div = pow10_i128(p.scale_mul);
              i128 prod = (i128)((u128)a * (u128)b);
              i128 num = ((a < 0) == (b < 0)) ? (i128)((u128)prod + (u128)(div / 2))
                                              : (i128)((u128)prod - (u128)(div / 2));
              r = sdiv128(num, div);
            } else {
              ok = round_mul_128_overflow(a, b, p.scale_mul, &r);
            }
          }
        } break;
        default: {  // DIVIDE
          if (b == 0) {
            ok = false;
          } else if (!t128) {
            i128 mul = pow10_i128(p.scale_mul);
            i128 am = (i128)((u128)a * (u128)mul);
            i128 num = ((a < 0) == (b < 0)) ? (i128)((u128)am + (u128)(b / 2))
                                            : (i128)((u128)am - (u128)(b / 2));
            r = (i128)(int64_t)sdiv128(num, b);
          } else {
            r = round_div_128(a, b, p.scale_mul);
          }
        } break;
      }
    }
    *out = r;
    return ok;
}


// The division-free subset of dec_row, inlined by the fused interpreter (dev_expr.h) for every node that does not divide: operand
// conversions (multiplies by powers of ten with their range checks), plus / minus with the precision-38/18 check, multiply at
// scale_mul == 0. The rounding multiply (scale_mul > 0) and divide need 128- / 256-bit divisions that the compiler expands in place
// (~600 instructions each); nodes that need them (dec_op_needs_division) go through the full dec_row — in the interpreter ONE
// copy inside a rolled loop over the lane's row slots, in the run-time specialised kernels per slot with the DecOp folded to
// constants (decimal/src/arithmetic.rs:212-243).
__device__ __forceinline__ bool dec_row_nodiv(const DecOp& p, i128 av, i128 bv, bool a_dec, bool b_dec, bool t128, i128* out) {
  i128 a, b, r = 1;
  bool ok = convert_operand(av, a_dec, p.a_from_scale, p.a_to_scale, p.a_to_precision, p.a_check, t128, &a);
  ok = convert_operand(bv, b_dec, p.b_from_scale, p.b_to_scale, p.b_to_precision, p.b_check, t128, &b) && ok;
  if (ok) {
    if (p.op == DBHIP_OP_MULTIPLY) {
      r = wrap_T((i128)((u128)a * (u128)b), t128);   // scale_mul == 0 (checked on the host)
    } else {
      i128 t = p.op == DBHIP_OP_PLUS ? (i128)((u128)a + (u128)b) : (i128)((u128)a - (u128)b);
      t = wrap_T(t, t128);
      if (p.overflow) {
        const i128 mx = max_for_precision(p.ret_precision);
        if (t < -mx || t > mx) ok = false;
      }
      r = t;
    }
  }
  *out = r;
  return ok;
}
__host__ __device__ inline bool dec_op_needs_division(const DecOp& p) { return p.op == DBHIP_OP_DIVIDE || (p.op == DBHIP_OP_MULTIPLY && p.scale_mul != 0); }

// host: decode one decimal call node (k_decimal.hip)
int32_t dbhip_decimal_decode_internal(int op, int a_type, int a_prec, int a_scale, int b_type, int b_prec, int b_scale,
                                      DecOp* out, int* out_type, int* out_precision, int* out_scale);

// k_decimal.hip — decimal +,-,*,/ column kernel.
// Reference semantics:
//   result size / operand sizes   decimal/src/arithmetic.rs:80-139 (ArithmeticOp::result_size)
//   operand conversion            :141-153 convert_to_decimal -> cast.rs:701-753 (integer_to_decimal),
//                                 cast.rs:901-979 (decimal_expand_cast), :1036-1049
//   compute                       :190-316 binary_decimal in T = storage class of the
//                                 result precision (types/decimal.rs:1713-1722)
//   rounding                      types/decimal.rs:759-797 (i64), :1024-1060 (i128 via i256)
// Error rows get the value 1 (`T::one()`), exactly like the reference builders.
#include "dev_common.h"
#include "dev_load.h"
#include "runtime.h"

using namespace dbhip;

namespace {

struct DecParams {
  const void* a;
  const void* b;
  void* out;
  const uint8_t* a_validity;
  const uint8_t* b_validity;
  int64_t a_voff, b_voff;
  int64_t n;
  uint32_t* err_words;
  unsigned long long* err_count;
  int a_type, b_type, out_type;
  int a_scalar, b_scalar;
  int op;
  // conversion of each operand into its bound size
  int a_from_scale, a_to_scale, a_to_precision, a_check;   // check: range-check after rescale
  int b_from_scale, b_to_scale, b_to_precision, b_check;
  int t_is_128;       // compute type T: 0 = i64, 1 = i128
  int ret_precision, ret_scale;
  int overflow;       // return precision == T::MAX_PRECISION
  int scale_mul;      // multiply: sa+sb-sr ; divide: sb+sr-sa
};

__device__ __forceinline__ i128 load_operand(const void* p, int type, bool scalar, int64_t i) {
  int64_t j = scalar ? 0 : i;
  switch (type) {
    case DBHIP_T_DEC128: return ((const i128*)p)[j];
    case DBHIP_T_DEC64: case DBHIP_T_I64: return (i128)((const int64_t*)p)[j];
    case DBHIP_T_I8: return (i128)((const int8_t*)p)[j];
    case DBHIP_T_I16: return (i128)((const int16_t*)p)[j];
    case DBHIP_T_I32: return (i128)((const int32_t*)p)[j];
    case DBHIP_T_U8: return (i128)((const uint8_t*)p)[j];
    case DBHIP_T_U16: return (i128)((const uint16_t*)p)[j];
    case DBHIP_T_U32: return (i128)((const uint32_t*)p)[j];
    default: return (i128)((const uint64_t*)p)[j];  // U64
  }
}

__device__ __forceinline__ i128 wrap_T(i128 v, bool t128) { return t128 ? v : (i128)(int64_t)v; }

__device__ __forceinline__ i128 max_for_precision(int p) { return pow10_i128(p) - 1; }

// checked multiply in T (i64 or i128)
__device__ __forceinline__ bool checked_mul_T(i128 x, i128 f, bool t128, i128* out) {
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
__device__ __forceinline__ bool convert_operand(i128 x, bool is_decimal, int from_scale, int to_scale,
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

__device__ __forceinline__ void dec_raise(const DecParams& p, int64_t row) {
  if (p.a_validity && !bit_get(p.a_validity, p.a_voff + row)) return;
  if (p.b_validity && !bit_get(p.b_validity, p.b_voff + row)) return;
  if (p.err_words) atomicAnd(&p.err_words[row >> 5], ~(1u << (row & 31)));
  if (p.err_count) atomicAdd(p.err_count, 1ULL);
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
__device__ __forceinline__ bool dec_row(const DecParams& p, i128 av, i128 bv, bool a_dec, bool b_dec, bool t128, i128* out) {
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
              i128 res = num / div;
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
              r = num / div;
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
            r = (i128)(int64_t)(num / b);
          } else {
            r = round_div_128(a, b, p.scale_mul);
          }
        } break;
      }
    }
    *out = r;
    return ok;
}

// the N loads of one operand with the type switch outside the row loop (independent loads in one basic block)
template <int N>
__device__ __forceinline__ void load_operand_n(const void* p, int type, bool scalar, const int64_t (&i)[N], i128 (&out)[N]) {
  int64_t j[N];
#pragma unroll
  for (int u = 0; u < N; ++u) j[u] = scalar ? 0 : i[u];
  switch (type) {
    case DBHIP_T_DEC128:
#pragma unroll
      for (int u = 0; u < N; ++u) out[u] = ((const i128*)p)[j[u]];
      break;
    case DBHIP_T_DEC64: case DBHIP_T_I64:
#pragma unroll
      for (int u = 0; u < N; ++u) out[u] = (i128)((const int64_t*)p)[j[u]];
      break;
    case DBHIP_T_I8:
#pragma unroll
      for (int u = 0; u < N; ++u) out[u] = (i128)((const int8_t*)p)[j[u]];
      break;
    case DBHIP_T_I16:
#pragma unroll
      for (int u = 0; u < N; ++u) out[u] = (i128)((const int16_t*)p)[j[u]];
      break;
    case DBHIP_T_I32:
#pragma unroll
      for (int u = 0; u < N; ++u) out[u] = (i128)((const int32_t*)p)[j[u]];
      break;
    case DBHIP_T_U8:
#pragma unroll
      for (int u = 0; u < N; ++u) out[u] = (i128)((const uint8_t*)p)[j[u]];
      break;
    case DBHIP_T_U16:
#pragma unroll
      for (int u = 0; u < N; ++u) out[u] = (i128)((const uint16_t*)p)[j[u]];
      break;
    case DBHIP_T_U32:
#pragma unroll
      for (int u = 0; u < N; ++u) out[u] = (i128)((const uint32_t*)p)[j[u]];
      break;
    default:
#pragma unroll
      for (int u = 0; u < N; ++u) out[u] = (i128)((const uint64_t*)p)[j[u]];  // U64
      break;
  }
}

// Four rows per lane (rows base + u T + t): with one row per lane only 16 bytes per lane are in flight and the kernel
// is latency bound (0.44 of the HBM rate on dec64 x dec64 -> dec128).
__global__ __launch_bounds__(256) void decimal_kernel(DecParams p) {
  const bool t128 = p.t_is_128;
  const bool a_dec = p.a_type == DBHIP_T_DEC64 || p.a_type == DBHIP_T_DEC128;
  const bool b_dec = p.b_type == DBHIP_T_DEC64 || p.b_type == DBHIP_T_DEC128;
  constexpr int U = 4;
  const int64_t T = (int64_t)gridDim.x * blockDim.x;
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (int64_t base = 0; base < p.n; base += U * T) {
    int64_t row[U];
    bool in[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      row[u] = base + u * T + t;
      in[u] = row[u] < p.n;
      if (!in[u]) row[u] = p.n - 1;
    }
    i128 av[U], bv[U];
    load_operand_n<U>(p.a, p.a_type, p.a_scalar, row, av);
    load_operand_n<U>(p.b, p.b_type, p.b_scalar, row, bv);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (!in[u]) continue;
      const int64_t i = row[u];
      i128 r;
      if (!dec_row(p, av[u], bv[u], a_dec, b_dec, t128, &r)) {
        dec_raise(p, i);
        r = 1;
      }
      if (p.out_type == DBHIP_T_DEC128) ((i128*)p.out)[i] = r;
      else ((int64_t*)p.out)[i] = (int64_t)r;
    }
  }
}

struct DSize {
  int p, s;
};

bool decimal_props(int type, int prec, int scale, DSize* out) {
  switch (type) {
    case DBHIP_T_DEC64: case DBHIP_T_DEC128: *out = {prec, scale}; return prec >= 1 && prec <= 38 && scale <= prec;
    case DBHIP_T_I8: case DBHIP_T_U8: *out = {3, 0}; return true;     // number.rs:452-465
    case DBHIP_T_I16: case DBHIP_T_U16: *out = {5, 0}; return true;
    case DBHIP_T_I32: case DBHIP_T_U32: *out = {10, 0}; return true;
    case DBHIP_T_I64: *out = {19, 0}; return true;
    case DBHIP_T_U64: *out = {20, 0}; return true;
  }
  return false;
}

inline int imin(int a, int b) { return a < b ? a : b; }
inline int imax(int a, int b) { return a > b ? a : b; }

// ArithmeticOp::result_size (arithmetic.rs:80-139) restricted to <=38 digits operands
bool result_size(int op, DSize a, DSize b, DSize* left, DSize* right, DSize* ret) {
  int precision, scale;
  int la = a.p - a.s, lb = b.p - b.s;
  switch (op) {
    case DBHIP_OP_MULTIPLY:
      scale = imin(a.s + b.s, imax(imax(a.s, b.s), 12));
      precision = la + lb + scale;
      break;
    case DBHIP_OP_DIVIDE:
      scale = imax(a.s, imin(a.s + 6, 12));
      precision = la + b.s + scale;
      break;
    case DBHIP_OP_PLUS:
    case DBHIP_OP_MINUS:
      scale = imax(a.s, b.s);
      precision = imax(la, lb) + scale + 1;
      break;
    default:
      return false;
  }
  precision = imin(precision, 38);  // both operands are <= Decimal128
  if (precision < 1 || scale > precision) return false;  // DecimalSize::new
  *ret = {precision, scale};
  switch (op) {
    case DBHIP_OP_MULTIPLY:
      *left = {precision, a.s};
      *right = {precision, b.s};
      break;
    case DBHIP_OP_DIVIDE: {
      int pp = imax(precision, imax(a.p, b.p));
      *left = {pp, a.s};
      *right = {pp, b.s};
    } break;
    default:
      *left = *ret;
      *right = *ret;
  }
  return left->s <= left->p && right->s <= right->p;
}

}  // namespace

extern "C" {

int32_t dbhip_decimal_result_size(int32_t op, uint8_t lp, uint8_t ls, uint8_t rp, uint8_t rs,
                                  uint8_t* out_precision_host, uint8_t* out_scale_host) {
  DSize l, r, ret;
  if (!result_size(op, {lp, ls}, {rp, rs}, &l, &r, &ret)) {
    set_error("dbhip_decimal_result_size: no decimal result for op %d on (%d,%d),(%d,%d)", op, lp, ls, rp, rs);
    return DBHIP_ERR_INVALID;
  }
  if (out_precision_host) *out_precision_host = (uint8_t)ret.p;
  if (out_scale_host) *out_scale_host = (uint8_t)ret.s;
  return DBHIP_OK;
}

int32_t dbhip_decimal_arith(int32_t op, const dbhip_col* lhs, const dbhip_col* rhs, int64_t n,
                            int32_t out_type, uint8_t out_precision, uint8_t out_scale, void* out,
                            uint8_t* err_bitmap, uint64_t* err_count_dev, void* stream) {
  DBHIP_REQUIRE(lhs && rhs && (out || n == 0), "dbhip_decimal_arith: NULL argument");
  DSize a, b, left, right, ret;
  if (!decimal_props(lhs->type, lhs->precision, lhs->scale, &a) ||
      !decimal_props(rhs->type, rhs->precision, rhs->scale, &b)) {
    set_error("dbhip_decimal_arith: operand types (%d,%d) have no decimal properties", lhs->type, rhs->type);
    return DBHIP_ERR_INVALID;
  }
  bool a_dec = lhs->type == DBHIP_T_DEC64 || lhs->type == DBHIP_T_DEC128;
  bool b_dec = rhs->type == DBHIP_T_DEC64 || rhs->type == DBHIP_T_DEC128;
  DBHIP_REQUIRE(a_dec || b_dec, "dbhip_decimal_arith: at least one side must be decimal");
  if (!result_size(op, a, b, &left, &right, &ret)) {
    set_error("dbhip_decimal_arith: unsupported op %d", op);
    return DBHIP_ERR_INVALID;
  }
  int want_type = ret.p <= 18 ? DBHIP_T_DEC64 : DBHIP_T_DEC128;
  if (ret.p != out_precision || ret.s != out_scale || want_type != out_type) {
    set_error("dbhip_decimal_arith: result is type %d Decimal(%d,%d); caller passed type %d Decimal(%d,%d)",
              want_type, ret.p, ret.s, out_type, out_precision, out_scale);
    return DBHIP_ERR_INVALID;
  }
  hipStream_t s = resolve_stream(stream);
  if (err_bitmap) DBHIP_CHECK(hipMemsetAsync(err_bitmap, 0xFF, (size_t)ceil_div(n, 32) * 4, s));
  if (n == 0) return DBHIP_OK;
  DecParams p;
  p.a = lhs->data; p.b = rhs->data; p.out = out;
  p.a_validity = lhs->validity; p.b_validity = rhs->validity;
  p.a_voff = lhs->validity_offset; p.b_voff = rhs->validity_offset;
  p.n = n;
  p.err_words = (uint32_t*)err_bitmap;
  p.err_count = (unsigned long long*)err_count_dev;
  p.a_type = lhs->type; p.b_type = rhs->type; p.out_type = out_type;
  p.a_scalar = lhs->is_scalar; p.b_scalar = rhs->is_scalar;
  p.op = op;
  p.a_from_scale = a.s; p.a_to_scale = left.s; p.a_to_precision = left.p;
  p.a_check = a_dec ? (a.s != left.s) : 1;
  p.b_from_scale = b.s; p.b_to_scale = right.s; p.b_to_precision = right.p;
  p.b_check = b_dec ? (b.s != right.s) : 1;
  p.t_is_128 = ret.p > 18;
  p.ret_precision = ret.p; p.ret_scale = ret.s;
  p.overflow = ret.p == (p.t_is_128 ? 38 : 18);
  p.scale_mul = op == DBHIP_OP_MULTIPLY ? a.s + b.s - ret.s
                                        : (op == DBHIP_OP_DIVIDE ? b.s + ret.s - a.s : 0);
  if (p.scale_mul < 0 || p.scale_mul > 38) {
    set_error("dbhip_decimal_arith: scale shift %d outside the supported range", p.scale_mul);
    return DBHIP_ERR_UNSUPPORTED;
  }
  hipLaunchKernelGGL(decimal_kernel, dim3(grid_for(ceil_div(n, 4), 256, 1024)), dim3(256), 0, s, p);
  DBHIP_LAUNCH_CHECK();
  return DBHIP_OK;
}

}  // extern "C"

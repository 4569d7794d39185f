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
#include "dev_decimal.h"
#include "dev_load.h"
#include "runtime.h"

using namespace dbhip;

namespace {

struct DecParams : DecOp {
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
};

__device__ __forceinline__ i128 load_operand(const void* p, int type, bool scalar, int64_t i) {
  int64_t j = scalar ? 0 : i;
  switch (type) {
    case DBHIP_T_DEC128: return ((const i128*)p)[j];
    case DBHIP_T_DEC256: return ((const i128*)p)[2 * j];  // as_decimal::<T>() into a narrower T: the low bits
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

__device__ __forceinline__ void dec_raise(const DecParams& p, int64_t row) {
  if (p.a_validity && !bit_get(p.a_validity, p.a_voff + row)) return;
  if (p.b_validity && !bit_get(p.b_validity, p.b_voff + row)) return;
  if (p.err_words) atomicAnd(&p.err_words[row >> 5], ~(1u << (row & 31)));
  if (p.err_count) atomicAdd(p.err_count, 1ULL);
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
    case DBHIP_T_DEC256:  // a Decimal256 operand under a result of at most 38 digits (T = i64 / i128): as_decimal::<T>()
#pragma unroll
      for (int u = 0; u < N; ++u) out[u] = ((const i128*)p)[2 * j[u]];
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
  const bool a_dec = p.a_type == DBHIP_T_DEC64 || p.a_type == DBHIP_T_DEC128 || p.a_type == DBHIP_T_DEC256;
  const bool b_dec = p.b_type == DBHIP_T_DEC64 || p.b_type == DBHIP_T_DEC128 || p.b_type == DBHIP_T_DEC256;
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
    case DBHIP_T_DEC64: case DBHIP_T_DEC128: case DBHIP_T_DEC256: *out = {prec, scale}; return prec >= 1 && prec <= 76 && scale <= prec;
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

// ArithmeticOp::result_size (arithmetic.rs:80-139)
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
  precision = imin(precision, (a.p <= 38 && b.p <= 38) ? 38 : 76);  // :115-121: both at most Decimal128 -> clamp to 38
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

namespace dbhip {
int32_t decimal256_arith(int32_t op, const dbhip_col* lhs, const dbhip_col* rhs, int64_t n, int32_t out_type, uint8_t out_precision,
                         uint8_t out_scale, void* out, uint8_t* err_bitmap, uint64_t* err_count_dev, void* stream);  // k_decimal256.hip
}

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
  if (out_precision > 38)  // T = i256 (an operand beyond 38 digits): the Decimal256 class, k_decimal256.hip
    return decimal256_arith(op, lhs, rhs, n, out_type, out_precision, out_scale, out, err_bitmap, err_count_dev, stream);
  DecParams p;
  int want_type, rp, rs;
  int32_t rc = dbhip_decimal_decode_internal(op, lhs->type, lhs->precision, lhs->scale, rhs->type, rhs->precision, rhs->scale, &p,
                                             &want_type, &rp, &rs);
  if (rc) return rc;
  if (rp != out_precision || rs != out_scale || want_type != out_type) {
    set_error("dbhip_decimal_arith: result is type %d Decimal(%d,%d); caller passed type %d Decimal(%d,%d)",
              want_type, rp, rs, out_type, out_precision, out_scale);
    return DBHIP_ERR_INVALID;
  }
  hipStream_t s = resolve_stream(stream);
  if (err_bitmap) DBHIP_CHECK(hipMemsetAsync(err_bitmap, 0xFF, (size_t)ceil_div(n, 32) * 4, s));
  if (n == 0) return DBHIP_OK;
  p.a = lhs->data; p.b = rhs->data; p.out = out;
  p.a_validity = lhs->validity; p.b_validity = rhs->validity;
  p.a_voff = lhs->validity_offset; p.b_voff = rhs->validity_offset;
  p.n = n;
  p.err_words = (uint32_t*)err_bitmap;
  p.err_count = (unsigned long long*)err_count_dev;
  p.a_type = lhs->type; p.b_type = rhs->type; p.out_type = out_type;
  p.a_scalar = lhs->is_scalar; p.b_scalar = rhs->is_scalar;
  hipLaunchKernelGGL(decimal_kernel, dim3(grid_for(ceil_div(n, 4), 256, 1024)), dim3(256), 0, s, p);
  DBHIP_LAUNCH_CHECK();
  return DBHIP_OK;
}

}  // extern "C"

// Host-side decode of one decimal call node (shared with the fused expression compiler, k_expr.hip): operand types ->
// result storage class / DecimalSize (ArithmeticOp::result_size, arithmetic.rs:80-139) and everything dec_row needs.
int32_t dbhip_decimal_decode_internal(int op, int a_type, int a_prec, int a_scale, int b_type, int b_prec, int b_scale,
                                      DecOp* out, int* out_type, int* out_precision, int* out_scale) {
  DSize a, b, left, right, ret;
  if (!decimal_props(a_type, a_prec, a_scale, &a) || !decimal_props(b_type, b_prec, b_scale, &b)) {
    set_error("decimal arithmetic: operand types (%d,%d) have no decimal properties", a_type, b_type);
    return DBHIP_ERR_INVALID;
  }
  const bool a_dec = a_type == DBHIP_T_DEC64 || a_type == DBHIP_T_DEC128 || a_type == DBHIP_T_DEC256;
  const bool b_dec = b_type == DBHIP_T_DEC64 || b_type == DBHIP_T_DEC128 || b_type == DBHIP_T_DEC256;
  DBHIP_REQUIRE(a_dec || b_dec, "decimal arithmetic: at least one side must be decimal");
  if (!result_size(op, a, b, &left, &right, &ret)) {
    set_error("decimal arithmetic: unsupported op %d", op);
    return DBHIP_ERR_INVALID;
  }
  if (ret.p > 38) {  // T = i256: dbhip_decimal_arith routes such nodes to k_decimal256.hip before it gets here; fused programs keep them out
    set_error("decimal arithmetic: Decimal(%d,%d) results (T = i256) are evaluated by dbhip_decimal_arith, not inside fused programs", ret.p, ret.s);
    return DBHIP_ERR_UNSUPPORTED;
  }
  DecOp& p = *out;
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
    set_error("decimal arithmetic: scale shift %d outside the supported range", p.scale_mul);
    return DBHIP_ERR_UNSUPPORTED;
  }
  {
    const bool pass_a = a_dec ? !p.a_check : p.a_to_scale == 0, pass_b = b_dec ? !p.b_check : p.b_to_scale == 0;
    p.trivial = pass_a && pass_b && (((op == DBHIP_OP_PLUS || op == DBHIP_OP_MINUS) && !p.overflow) || (op == DBHIP_OP_MULTIPLY && p.scale_mul == 0));
  }
  *out_type = ret.p <= 18 ? DBHIP_T_DEC64 : DBHIP_T_DEC128;
  *out_precision = ret.p;
  *out_scale = ret.s;
  return DBHIP_OK;
}

// k_decimal256.hip — the Decimal256 (i256) class (types/decimal.rs:1282-1500) and the decimal functions that span
// storage classes: binary arithmetic with T = i256, unary minus, comparisons with a Decimal256 side, to_decimal / try_to_decimal
// for decimal and integer sources. Row semantics: dev_i256.h (which also cites the reference lines).
// These columns are rare next to Decimal64 / 128 and their rows cost hundreds of instructions (32-bit-limb long division), so
// the kernels are one row per lane, grid-stride, with plain 16-byte loads — the arithmetic, not HBM, bounds them.
#include "dev_common.h"
#include "dev_i256.h"
#include "runtime.h"

using namespace dbhip;

namespace {

__device__ __forceinline__ bool is_dec_type(int t) { return t == DBHIP_T_DEC64 || t == DBHIP_T_DEC128 || t == DBHIP_T_DEC256; }
inline bool is_dec_type_h(int t) { return t == DBHIP_T_DEC64 || t == DBHIP_T_DEC128 || t == DBHIP_T_DEC256; }
inline int dec_bits_of(int t) { return t == DBHIP_T_DEC64 ? 64 : (t == DBHIP_T_DEC128 ? 128 : (t == DBHIP_T_DEC256 ? 256 : 0)); }
inline int dec_type_of_bits(int bits) { return bits == 64 ? DBHIP_T_DEC64 : (bits == 128 ? DBHIP_T_DEC128 : DBHIP_T_DEC256); }

// any decimal / integer column value, sign- (or zero-) extended into the carrier
__device__ __forceinline__ I256 load_i256(const void* p, int type, bool scalar, int64_t i) {
  const int64_t j = scalar ? 0 : i;
  switch (type) {
    case DBHIP_T_DEC256: {
      const uint4* q = (const uint4*)p + 2 * j;
      const uint4 lo = q[0], hi = q[1];
      I256 r;
      r.w[0] = (uint64_t)lo.x | ((uint64_t)lo.y << 32); r.w[1] = (uint64_t)lo.z | ((uint64_t)lo.w << 32);
      r.w[2] = (uint64_t)hi.x | ((uint64_t)hi.y << 32); r.w[3] = (uint64_t)hi.z | ((uint64_t)hi.w << 32);
      return r;
    }
    case DBHIP_T_DEC128: {
      const uint4 v = ((const uint4*)p)[j];
      return i256_from_i128_words((uint64_t)v.x | ((uint64_t)v.y << 32), (uint64_t)v.z | ((uint64_t)v.w << 32));
    }
    case DBHIP_T_DEC64: case DBHIP_T_I64: return i256_from_i64(((const int64_t*)p)[j]);
    case DBHIP_T_I8: return i256_from_i64(((const int8_t*)p)[j]);
    case DBHIP_T_I16: return i256_from_i64(((const int16_t*)p)[j]);
    case DBHIP_T_I32: return i256_from_i64(((const int32_t*)p)[j]);
    case DBHIP_T_U8: return i256_from_u64(((const uint8_t*)p)[j]);
    case DBHIP_T_U16: return i256_from_u64(((const uint16_t*)p)[j]);
    case DBHIP_T_U32: return i256_from_u64(((const uint32_t*)p)[j]);
    default: return i256_from_u64(((const uint64_t*)p)[j]);  // U64
  }
}
__device__ __forceinline__ void store_bits(void* out, int bits, int64_t i, const I256& v) {
  if (bits == 64) ((uint64_t*)out)[i] = v.w[0];
  else if (bits == 128) ((uint4*)out)[i] = make_uint4((uint32_t)v.w[0], (uint32_t)(v.w[0] >> 32), (uint32_t)v.w[1], (uint32_t)(v.w[1] >> 32));
  else {
    uint4* q = (uint4*)out + 2 * i;
    q[0] = make_uint4((uint32_t)v.w[0], (uint32_t)(v.w[0] >> 32), (uint32_t)v.w[1], (uint32_t)(v.w[1] >> 32));
    q[1] = make_uint4((uint32_t)v.w[2], (uint32_t)(v.w[2] >> 32), (uint32_t)v.w[3], (uint32_t)(v.w[3] >> 32));
  }
}

struct Arith256Params {
  Dec256Op op;
  const void* a;
  const void* b;
  void* out;
  const uint8_t* a_validity;
  const uint8_t* b_validity;
  int64_t a_voff, b_voff, n;
  uint32_t* err_words;
  unsigned long long* err_count;
  int a_type, b_type, a_scalar, b_scalar;
};

__global__ __launch_bounds__(256) void decimal256_arith_kernel(Arith256Params p) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < p.n; i += (int64_t)gridDim.x * blockDim.x) {
    const I256 x = load_i256(p.a, p.a_type, p.a_scalar, i), y = load_i256(p.b, p.b_type, p.b_scalar, i);
    I256 r;
    if (!dec256_row(p.op, x, y, &r)) {
      r = i256_from_u64(1);  // T::one()
      const bool a_null = p.a_validity && !bit_get(p.a_validity, p.a_voff + (p.a_scalar ? 0 : i));
      const bool b_null = p.b_validity && !bit_get(p.b_validity, p.b_voff + (p.b_scalar ? 0 : i));
      if (!a_null && !b_null) {  // NULL rows never raise (function.rs:536-543)
        if (p.err_words) atomicAnd(&p.err_words[i >> 5], ~(1u << (i & 31)));
        if (p.err_count) atomicAdd(p.err_count, 1ULL);
      }
    }
    store_bits(p.out, 256, i, r);
  }
}

__global__ __launch_bounds__(256) void decimal_neg_kernel(const void* src, int type, int bits, int64_t n, void* out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    store_bits(out, bits, i, i256_neg(load_i256(src, type, false, i)));  // the low `bits` of the 256-bit negation = wrapping -t in T
}

struct Cmp256Params {
  const void* a;
  const void* b;
  uint8_t* out;
  int64_t n, out_bytes;
  int a_type, b_type, a_scalar, b_scalar, op, bits;
  int fa_one, fb_one, same_f;
  I256 fa, fb;
};
__device__ __forceinline__ bool apply_cmp_op(int op, int c) {
  switch (op) {
    case DBHIP_CMP_EQ: return c == 0;
    case DBHIP_CMP_NOTEQ: return c != 0;
    case DBHIP_CMP_LT: return c < 0;
    case DBHIP_CMP_LTE: return c <= 0;
    case DBHIP_CMP_GT: return c > 0;
    default: return c >= 0;
  }
}
__global__ __launch_bounds__(256) void cmp_decimal256_kernel(Cmp256Params p) {
  const int64_t n_pad = (p.n + 63) & ~63LL;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_pad; i += (int64_t)gridDim.x * blockDim.x) {
    bool r = false;
    if (i < p.n) {
      const I256 a = load_i256(p.a, p.a_type, p.a_scalar, i), b = load_i256(p.b, p.b_type, p.b_scalar, i);
      r = apply_cmp_op(p.op, dec256_cmp3(a, b, p.fa, p.fb, p.fa_one, p.fb_one, p.same_f, p.bits));
    }
    const uint64_t m = __ballot(r);
    const int lane = lane_id();
    if ((lane & 7) == 0) {
      const int64_t byte = i >> 3;
      if (byte < p.out_bytes) p.out[byte] = (uint8_t)(m >> lane);
    }
  }
}

struct Cast256Params {
  Dec256Cast c;
  const void* src;
  const uint8_t* validity;
  int64_t voff, n;
  void* out;
  uint64_t* bitmap;  // whole 64-bit words, preset to ones
  unsigned long long* err_count;
  int src_type, is_try;
};
__global__ __launch_bounds__(256) void decimal_cast_kernel(Cast256Params p) {
  const int64_t n_pad = (p.n + 63) & ~63LL;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_pad; i += (int64_t)gridDim.x * blockDim.x) {
    bool keep = true;  // the row's bit in `bitmap`
    if (i < p.n) {
      const bool valid = !p.validity || bit_get(p.validity, p.voff + i);
      I256 y;
      const bool ok = dec256_cast_row(p.c, load_i256(p.src, p.src_type, false, i), &y);
      if (!ok) y = i256_from_u64(1);
      store_bits(p.out, p.c.dbits, i, y);
      if (p.is_try) keep = ok && valid;          // try_to_decimal: result validity
      else if (!ok && valid) {                    // to_decimal: "Decimal overflow" row error
        keep = false;
        if (p.err_count) atomicAdd(p.err_count, 1ULL);
      }
    }
    const uint64_t m = __ballot(keep);
    if (p.bitmap && lane_id() == 0 && i < p.n) p.bitmap[i >> 6] = m;
  }
}

}  // namespace

// ---- entry points reached through the dispatchers of k_decimal.hip / k_cmp_filter.hip -------------------------------------------
namespace dbhip {

// binary_decimal whose result has more than 38 digits (T = i256). Operands: decimals of any storage class or integers.
int32_t decimal256_arith(int32_t op, const dbhip_col* lhs, const dbhip_col* rhs, int64_t n, int32_t out_type, uint8_t out_precision,
                         uint8_t out_scale, void* out, uint8_t* err_bitmap, uint64_t* err_count_dev, void* stream) {
  auto props = [](const dbhip_col* c, DecSize* o) {
    switch (c->type) {
      case DBHIP_T_DEC64: case DBHIP_T_DEC128: case DBHIP_T_DEC256: *o = {c->precision, c->scale}; return c->precision >= 1 && c->precision <= 76 && c->scale <= c->precision;
      case DBHIP_T_I8: case DBHIP_T_U8: *o = {3, 0}; return true;
      case DBHIP_T_I16: case DBHIP_T_U16: *o = {5, 0}; return true;
      case DBHIP_T_I32: case DBHIP_T_U32: *o = {10, 0}; return true;
      case DBHIP_T_I64: *o = {19, 0}; return true;
      case DBHIP_T_U64: *o = {20, 0}; return true;
    }
    return false;
  };
  DecSize a, b, ret;
  if (!props(lhs, &a) || !props(rhs, &b)) {
    set_error("decimal arithmetic: operand types (%d,%d) have no decimal properties", lhs->type, rhs->type);
    return DBHIP_ERR_INVALID;
  }
  Arith256Params p;
  if (!dec256_make_op(op, is_dec_type_h(lhs->type), a, is_dec_type_h(rhs->type), b, &p.op, &ret)) {
    set_error("decimal arithmetic: no Decimal256 result for op %d on (%d,%d),(%d,%d)", op, a.p, a.s, b.p, b.s);
    return DBHIP_ERR_INVALID;
  }
  if (ret.p != out_precision || ret.s != out_scale || out_type != DBHIP_T_DEC256) {
    set_error("dbhip_decimal_arith: result is type %d Decimal(%d,%d); caller passed type %d Decimal(%d,%d)", DBHIP_T_DEC256, ret.p, ret.s,
              out_type, out_precision, out_scale);
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
  p.a_type = lhs->type; p.b_type = rhs->type; p.a_scalar = lhs->is_scalar; p.b_scalar = rhs->is_scalar;
  hipLaunchKernelGGL(decimal256_arith_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, p);
  DBHIP_LAUNCH_CHECK();
  return DBHIP_OK;
}

// DecimalCmp with a Decimal256-stored side (any DecimalSizes)
int32_t cmp_decimal256(int32_t op, const dbhip_col* lhs, const dbhip_col* rhs, int64_t n, uint8_t* out_bitmap, void* stream) {
  if (n == 0) return DBHIP_OK;
  DBHIP_REQUIRE(lhs->precision >= 1 && lhs->precision <= 76 && rhs->precision >= 1 && rhs->precision <= 76 &&
                    lhs->scale <= lhs->precision && rhs->scale <= rhs->precision, "dbhip_cmp: bad DecimalSize");
  const int scale = lhs->scale > rhs->scale ? lhs->scale : rhs->scale;
  const int la = lhs->precision - lhs->scale, lb = rhs->precision - rhs->scale;
  int precision = (la > lb ? la : lb) + scale;  // calc_size (comparison.rs:369-384)
  const int cap = (lhs->precision <= 38 && rhs->precision <= 38) ? 38 : 76;
  if (precision > cap) precision = cap;
  Cmp256Params q;
  q.a = lhs->data; q.b = rhs->data; q.out = out_bitmap; q.n = n; q.out_bytes = ceil_div(n, 8);
  q.a_type = lhs->type; q.b_type = rhs->type; q.a_scalar = lhs->is_scalar; q.b_scalar = rhs->is_scalar;
  q.op = op; q.bits = dec_storage_bits(precision);
  q.fa = i256_pow10(scale - lhs->scale); q.fb = i256_pow10(scale - rhs->scale);
  q.fa_one = scale == lhs->scale; q.fb_one = scale == rhs->scale; q.same_f = lhs->scale == rhs->scale;
  hipLaunchKernelGGL(cmp_decimal256_kernel, dim3(grid_for(n, 256)), dim3(256), 0, resolve_stream(stream), q);
  DBHIP_LAUNCH_CHECK();
  return DBHIP_OK;
}

}  // namespace dbhip

extern "C" {

int32_t dbhip_decimal_neg(const dbhip_col* src, int64_t n, void* out, void* stream) {
  DBHIP_REQUIRE(src && (out || n == 0), "dbhip_decimal_neg: NULL argument");
  const int bits = dec_bits_of(src->type);
  DBHIP_REQUIRE(bits != 0 && !src->is_scalar, "dbhip_decimal_neg: the argument must be a decimal column");
  if (n == 0) return DBHIP_OK;
  hipLaunchKernelGGL(decimal_neg_kernel, dim3(grid_for(n, 256)), dim3(256), 0, resolve_stream(stream), src->data, src->type, bits, n, out);
  DBHIP_LAUNCH_CHECK();
  return DBHIP_OK;
}

int32_t dbhip_decimal_cast(const dbhip_col* src, int32_t dst_type, uint8_t dst_precision, uint8_t dst_scale, int32_t is_try,
                           int32_t rounding_mode, int64_t n, void* out, uint8_t* bitmap, uint64_t* err_count_dev, void* stream) {
  DBHIP_REQUIRE(src && (out || n == 0), "dbhip_decimal_cast: NULL argument");
  DBHIP_REQUIRE(!src->is_scalar, "dbhip_decimal_cast: scalar sources are constant-folded by the planner");
  DBHIP_REQUIRE(bitmap || !is_try, "dbhip_decimal_cast: TRY_CAST needs the validity bitmap");
  DBHIP_REQUIRE((((uintptr_t)bitmap) & 7) == 0, "dbhip_decimal_cast: bitmap must be 8-byte aligned");
  const int sbits = dec_bits_of(src->type);
  if (!sbits) {
    switch (src->type) {
      case DBHIP_T_I8: case DBHIP_T_I16: case DBHIP_T_I32: case DBHIP_T_I64:
      case DBHIP_T_U8: case DBHIP_T_U16: case DBHIP_T_U32: case DBHIP_T_U64: break;
      default:
        set_error("dbhip_decimal_cast: source type %d (float / string / variant sources keep the CPU function)", src->type);
        return DBHIP_ERR_UNSUPPORTED;
    }
  }
  Cast256Params p;
  if (!dec256_make_cast(sbits, {src->precision, src->scale}, {dst_precision, dst_scale}, rounding_mode != 0, &p.c)) {
    set_error("dbhip_decimal_cast: bad DecimalSize (%d,%d) -> (%d,%d)", src->precision, src->scale, dst_precision, dst_scale);
    return DBHIP_ERR_INVALID;
  }
  if (dst_type != dec_type_of_bits(p.c.dbits)) {
    set_error("dbhip_decimal_cast: Decimal(%d,%d) is stored as type %d; caller passed %d", dst_precision, dst_scale,
              dec_type_of_bits(p.c.dbits), dst_type);
    return DBHIP_ERR_INVALID;
  }
  hipStream_t s = resolve_stream(stream);
  if (bitmap) DBHIP_CHECK(hipMemsetAsync(bitmap, 0xFF, (size_t)ceil_div(n, 64) * 8, s));
  if (n == 0) return DBHIP_OK;
  p.src = src->data; p.validity = src->validity; p.voff = src->validity_offset; p.n = n; p.out = out;
  p.bitmap = (uint64_t*)bitmap; p.err_count = (unsigned long long*)err_count_dev;
  p.src_type = src->type; p.is_try = is_try;
  hipLaunchKernelGGL(decimal_cast_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, p);
  DBHIP_LAUNCH_CHECK();
  return DBHIP_OK;
}

}  // extern "C"

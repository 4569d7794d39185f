// k_arith.hip — numeric arithmetic column kernels and the
// fused config-1 reduction. Reference semantics:
//   plus/minus/multiply  numeric_basic_arithmetic.rs:255-412 (cast both sides `as`
//                        the result type, then wrapping op; floats in f64)
//   divide               :410-427 (f64, "divided by zero")
//   div (intdiv)         :459-490 (f64 quotient, saturating cast like Rust `as`)
//   modulo               arithmetic_modulo.rs:28-184 (computed in LeastSuper, MIN % -1 = 0,
//                        "Division by zero")
//   result types         src/query/codegen/src/writes/arithmetics_type.rs:222-250
//
// One kernel serves every (lhs, rhs) type pair: operands are loaded with 16-byte
// (or the widest natural) vector loads chosen by a wave-uniform switch, widened
// losslessly to 64 bits, combined, and truncated to the result width on store —
// this equals "cast to the result type, then wrapping op" because the cast chain
// is value preserving modulo 2^64. HBM-bound: 4 rows per lane, grid-stride.
#include "dev_common.h"
#include "dev_load.h"
#include "runtime.h"

#include <math.h>

using namespace dbhip;

namespace {

__device__ __forceinline__ double wide_to_f64(uint64_t w, int cls) {
  if (cls == CLS_FLOAT) return __longlong_as_double((long long)w);
  if (cls == CLS_SIGNED) return (double)(int64_t)w;
  return (double)w;
}

// Rust `f64 as iN/uN`: truncate toward zero, saturate, NaN -> 0.
__device__ __forceinline__ uint64_t f64_to_int_sat(double x, int out_type) {
  int bits = type_bits(out_type);
  bool is_signed = type_class(out_type) == CLS_SIGNED;
  if (x != x) return 0;
  if (is_signed) {
    double lo = -ldexp(1.0, bits - 1);
    double hi = ldexp(1.0, bits - 1);  // exclusive
    if (x <= lo) return (uint64_t)(int64_t)((bits == 64) ? INT64_MIN : -(1LL << (bits - 1)));
    if (x >= hi) return (uint64_t)((bits == 64) ? INT64_MAX : ((1LL << (bits - 1)) - 1));
    return (uint64_t)(int64_t)x;
  } else {
    double hi = ldexp(1.0, bits);
    if (x <= 0.0) return 0;
    if (x >= hi) return (bits == 64) ? UINT64_MAX : ((1ULL << bits) - 1);
    return (uint64_t)x;
  }
}

// wrap a widened integer to `bits` with the given signedness and re-widen
__device__ __forceinline__ uint64_t wrap_to(uint64_t w, int bits, bool is_signed) {
  if (bits == 64) return w;
  if (is_signed) {
    int sh = 64 - bits;
    return (uint64_t)(((int64_t)(w << sh)) >> sh);
  }
  return w & ((1ULL << bits) - 1);
}

struct ArithParams {
  const void* a;
  const void* b;
  void* out;
  const uint8_t* a_validity;
  const uint8_t* b_validity;
  int64_t a_voff, b_voff;
  int64_t n;
  uint32_t* err_words;  // bitmap as 32-bit words, preset to all ones
  unsigned long long* err_count;
  int a_type, b_type, out_type;
  int a_scalar, b_scalar;
  int op;
  int m_type;  // LeastSuper type for modulo
};

__device__ __forceinline__ void raise_row(const ArithParams& p, int64_t row) {
  if (p.a_validity && !bit_get(p.a_validity, p.a_voff + row)) return;
  if (p.b_validity && !bit_get(p.b_validity, p.b_voff + row)) return;
  if (p.err_words) atomicAnd(&p.err_words[row >> 5], ~(1u << (row & 31)));
  if (p.err_count) atomicAdd(p.err_count, 1ULL);
}

__global__ __launch_bounds__(256) void arith_kernel(ArithParams p) {
  const int acls = type_class(p.a_type), bcls = type_class(p.b_type);
  const int ocls = type_class(p.out_type);
  const int64_t nquads = (p.n + 3) >> 2;
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < nquads;
       q += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i0 = q << 2;
    uint64_t a[4], b[4], r[4];
    load4_wide(p.a, p.a_type, p.a_scalar, i0, p.n, a);
    load4_wide(p.b, p.b_type, p.b_scalar, i0, p.n, b);
    switch (p.op) {
      case DBHIP_OP_PLUS:
      case DBHIP_OP_MINUS:
      case DBHIP_OP_MULTIPLY:
        if (ocls == CLS_FLOAT) {
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            double x = wide_to_f64(a[k], acls), y = wide_to_f64(b[k], bcls);
            double z = p.op == DBHIP_OP_PLUS ? x + y : (p.op == DBHIP_OP_MINUS ? x - y : x * y);
            r[k] = (uint64_t)__double_as_longlong(z);
          }
        } else {
#pragma unroll
          for (int k = 0; k < 4; ++k)
            r[k] = p.op == DBHIP_OP_PLUS ? a[k] + b[k]
                                         : (p.op == DBHIP_OP_MINUS ? a[k] - b[k] : a[k] * b[k]);
        }
        break;
      case DBHIP_OP_DIVIDE:
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          double x = wide_to_f64(a[k], acls), y = wide_to_f64(b[k], bcls);
          if (y == 0.0) {
            if (i0 + k < p.n) raise_row(p, i0 + k);
            r[k] = 0;  // F64::default()
          } else {
            r[k] = (uint64_t)__double_as_longlong(x / y);
          }
        }
        break;
      case DBHIP_OP_DIV0:     // div0_function (numeric_basic_arithmetic.rs:441-448): x / 0 = 0, no error
      case DBHIP_OP_DIVNULL:  // divnull_function (:450-457): x / 0 = NULL — the row's bit of `err_bitmap` is cleared
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          double x = wide_to_f64(a[k], acls), y = wide_to_f64(b[k], bcls);
          if (y == 0.0) {
            if (p.op == DBHIP_OP_DIVNULL && i0 + k < p.n) raise_row(p, i0 + k);
            r[k] = 0;  // F64::default()
          } else {
            r[k] = (uint64_t)__double_as_longlong(x / y);
          }
        }
        break;
      case DBHIP_OP_INTDIV:
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          double x = wide_to_f64(a[k], acls), y = wide_to_f64(b[k], bcls);
          if (y == 0.0) {
            if (i0 + k < p.n) raise_row(p, i0 + k);
            r[k] = 0;
          } else {
            r[k] = f64_to_int_sat(x / y, p.out_type);
          }
        }
        break;
      default: {  // MODULO
        const int mcls = type_class(p.m_type);
        const int mbits = type_bits(p.m_type);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          bool rhs_zero = (bcls == CLS_FLOAT) ? (wide_to_f64(b[k], bcls) == 0.0) : (b[k] == 0);
          if (rhs_zero) {
            if (i0 + k < p.n) raise_row(p, i0 + k);
            r[k] = 0;
            continue;
          }
          if (mcls == CLS_FLOAT) {
            double x = wide_to_f64(a[k], acls), y = wide_to_f64(b[k], bcls);
            double z;
            if (mbits == 32) z = (double)fmodf((float)x, (float)y);
            else z = fmod(x, y);
            r[k] = (uint64_t)__double_as_longlong(z);  // Modulo result is F64 when float
          } else {
            // cast both sides `as M`
            uint64_t x, y;
            if (acls == CLS_FLOAT) x = f64_to_int_sat(wide_to_f64(a[k], acls), p.m_type);
            else x = wrap_to(a[k], mbits, mcls == CLS_SIGNED);
            if (bcls == CLS_FLOAT) y = f64_to_int_sat(wide_to_f64(b[k], bcls), p.m_type);
            else y = wrap_to(b[k], mbits, mcls == CLS_SIGNED);
            if (mcls == CLS_SIGNED) {
              int64_t sx = (int64_t)x, sy = (int64_t)y;
              int64_t mn = (mbits == 64) ? INT64_MIN : -(1LL << (mbits - 1));
              if (sy == 0) { r[k] = 0; if (i0 + k < p.n) raise_row(p, i0 + k); }
              else if (sx == mn && sy == -1) r[k] = 0;
              else r[k] = (uint64_t)(sx % sy);
            } else {
              if (y == 0) { r[k] = 0; if (i0 + k < p.n) raise_row(p, i0 + k); }
              else r[k] = x % y;
            }
          }
        }
      } break;
    }
    store4_wide(p.out, p.out_type, i0, p.n, r);
  }
}

// ---- fused config-1: sum(a + b*c) over Int64 (wrapping) ---------------------
// 24 B/row read, nothing written but one atomic per block.
__global__ __launch_bounds__(256) void sum_abc_kernel(const int64_t* __restrict__ a,
                                                      const int64_t* __restrict__ b,
                                                      const int64_t* __restrict__ c, int64_t n,
                                                      unsigned long long* out) {
  uint64_t acc = 0;
  const int64_t npairs = n >> 1;
  typedef VecT<int64_t, 2> V2;
  const V2* a2 = (const V2*)a;
  const V2* b2 = (const V2*)b;
  const V2* c2 = (const V2*)c;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  // 4 independent 16-B load triples in flight per lane
  for (; i + 3 * stride < npairs; i += 4 * stride) {
    V2 x0 = a2[i], y0 = b2[i], z0 = c2[i];
    V2 x1 = a2[i + stride], y1 = b2[i + stride], z1 = c2[i + stride];
    V2 x2 = a2[i + 2 * stride], y2 = b2[i + 2 * stride], z2 = c2[i + 2 * stride];
    V2 x3 = a2[i + 3 * stride], y3 = b2[i + 3 * stride], z3 = c2[i + 3 * stride];
    acc += (uint64_t)x0.v[0] + (uint64_t)y0.v[0] * (uint64_t)z0.v[0];
    acc += (uint64_t)x0.v[1] + (uint64_t)y0.v[1] * (uint64_t)z0.v[1];
    acc += (uint64_t)x1.v[0] + (uint64_t)y1.v[0] * (uint64_t)z1.v[0];
    acc += (uint64_t)x1.v[1] + (uint64_t)y1.v[1] * (uint64_t)z1.v[1];
    acc += (uint64_t)x2.v[0] + (uint64_t)y2.v[0] * (uint64_t)z2.v[0];
    acc += (uint64_t)x2.v[1] + (uint64_t)y2.v[1] * (uint64_t)z2.v[1];
    acc += (uint64_t)x3.v[0] + (uint64_t)y3.v[0] * (uint64_t)z3.v[0];
    acc += (uint64_t)x3.v[1] + (uint64_t)y3.v[1] * (uint64_t)z3.v[1];
  }
  for (; i < npairs; i += stride) {
    V2 x = a2[i], y = b2[i], z = c2[i];
    acc += (uint64_t)x.v[0] + (uint64_t)y.v[0] * (uint64_t)z.v[0];
    acc += (uint64_t)x.v[1] + (uint64_t)y.v[1] * (uint64_t)z.v[1];
  }
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0)
    acc += (uint64_t)a[n - 1] + (uint64_t)b[n - 1] * (uint64_t)c[n - 1];
  acc = wave_sum_u64(acc);
  __shared__ uint64_t part[4];
  if (lane_id() == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out, (unsigned long long)(part[0] + part[1] + part[2] + part[3]));
}

// ---- sum(column) -----------------------------------------------------------
struct SumParams {
  const void* data;
  const uint8_t* validity;
  int64_t voff;
  int64_t n;
  void* out;
  int type;
};

// Integer sums wrap (NumberSumState, aggregate_sum.rs:113-129 with Sum = i64/u64).
__global__ __launch_bounds__(256) void sum_int_kernel(SumParams p) {
  uint64_t acc = 0;
  const int64_t nquads = (p.n + 3) >> 2;
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < nquads;
       q += (int64_t)gridDim.x * blockDim.x) {
    uint64_t v[4];
    load4_wide(p.data, p.type, false, q << 2, p.n, v);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      int64_t row = (q << 2) + k;
      bool ok = row < p.n && (!p.validity || bit_get(p.validity, p.voff + row));
      acc += ok ? v[k] : 0;
    }
  }
  acc = wave_sum_u64(acc);
  __shared__ uint64_t part[4];
  if (lane_id() == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0)
    atomicAdd((unsigned long long*)p.out, (unsigned long long)(part[0] + part[1] + part[2] + part[3]));
}

// Float sums: per-thread partials in f64, then a deterministic two-level tree
// (fixed grid => run-to-run reproducible). The reference adds left to right
// (sum_batch, aggregate_sum.rs:71-97); parity for f64 sums is therefore a
// tolerance test, stated in DESIGN.md.
__global__ __launch_bounds__(256) void sum_f64_partial_kernel(SumParams p, double* partials) {
  double acc = 0.0;
  const int64_t nquads = (p.n + 3) >> 2;
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < nquads;
       q += (int64_t)gridDim.x * blockDim.x) {
    uint64_t v[4];
    load4_wide(p.data, p.type, false, q << 2, p.n, v);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      int64_t row = (q << 2) + k;
      bool ok = row < p.n && (!p.validity || bit_get(p.validity, p.voff + row));
      acc += ok ? __longlong_as_double((long long)v[k]) : 0.0;
    }
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, 64);
  __shared__ double part[4];
  if (lane_id() == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partials[blockIdx.x] = (part[0] + part[1]) + (part[2] + part[3]);
}

__global__ void sum_f64_final_kernel(const double* partials, int nparts, double* out) {
  double acc = 0.0;
  for (int i = threadIdx.x; i < nparts; i += 64) acc += partials[i];
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, 64);
  if (threadIdx.x == 0) *out += acc;
}

int next_bits(int w) { return w < 64 ? w * 2 : 64; }

int make_num(int bits, bool is_signed, bool is_float) {
  if (is_float) return bits == 32 ? DBHIP_T_F32 : (bits == 64 ? DBHIP_T_F64 : -1);
  switch (bits) {
    case 8: return is_signed ? DBHIP_T_I8 : DBHIP_T_U8;
    case 16: return is_signed ? DBHIP_T_I16 : DBHIP_T_U16;
    case 32: return is_signed ? DBHIP_T_I32 : DBHIP_T_U32;
    case 64: return is_signed ? DBHIP_T_I64 : DBHIP_T_U64;
  }
  return -1;
}

bool is_plain_number(int t) { return t >= DBHIP_T_I8 && t <= DBHIP_T_F64; }

// arithmetic_coercion (codegen/src/writes/arithmetics_type.rs:222-250); op 100 = LeastSuper
int coerce(int op, int a, int b) {
  if (!is_plain_number(a) || !is_plain_number(b)) return -1;
  bool as = type_class(a) != CLS_UNSIGNED, bs = type_class(b) != CLS_UNSIGNED;
  bool af = type_class(a) == CLS_FLOAT, bf = type_class(b) == CLS_FLOAT;
  bool is_signed = as || bs, is_float = af || bf;
  int bw = type_bits(a) > type_bits(b) ? type_bits(a) : type_bits(b);
  switch (op) {
    case DBHIP_OP_PLUS:
    case DBHIP_OP_MULTIPLY:
      return make_num(next_bits(bw), is_signed, is_float);
    case DBHIP_OP_MINUS:
      return make_num(next_bits(bw), true, is_float);
    case DBHIP_OP_DIVIDE:
    case DBHIP_OP_DIV0:
    case DBHIP_OP_DIVNULL:
      return DBHIP_T_F64;
    case DBHIP_OP_INTDIV:
      return make_num(bw, is_signed, false);
    case DBHIP_OP_MODULO: {
      if (is_float) return DBHIP_T_F64;
      int rs = type_bits(b);
      return make_num(as ? next_bits(rs) : rs, as, false);
    }
    case 100:
      return make_num(bw, is_signed, is_float);
  }
  return -1;
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------------
// to_<number> / try_to_<number> (register_number_to_number, scalars/arithmetic/src/arithmetic.rs:448-700): lossless -> `as`;
// float -> integer: rounding_mode ? f64::round first : as is, then num_traits::cast; everything else lossy: num_traits::cast;
// None -> row error "number overflowed" (cast) or NULL (try_). num_traits 0.2.19's float -> int rule: truncate, Some iff
// MIN - 1 < x < MAX + 1 when the float type is wider than the integer, else MIN <= x < MAX + 1; unsigned: -1 < x.
// 4 rows per lane, row = chunk + 64 k + lane: coalesced 64-row segments, one validity word per (wave, k) from a ballot.
// ---------------------------------------------------------------------------------------------------------------------
namespace {
struct CastParams {
  const void* src;
  const uint8_t* valid;   // may be NULL
  int64_t voff;
  int src_type, dst_type, is_scalar, is_try, rounding, lossless;
  int64_t n;
  void* out;
  uint64_t* bitmap_words;   // try_: validity out (whole words); cast: error bitmap preset to ones (may be NULL)
  unsigned long long* err_count;
};

__device__ __forceinline__ uint64_t cast_load(const void* p, int type, int64_t j) {   // widened like load4_wide
  switch (type) {
    case DBHIP_T_I8: return (uint64_t)(int64_t)((const int8_t*)p)[j];
    case DBHIP_T_I16: return (uint64_t)(int64_t)((const int16_t*)p)[j];
    case DBHIP_T_I32: return (uint64_t)(int64_t)((const int32_t*)p)[j];
    case DBHIP_T_I64: return ((const uint64_t*)p)[j];
    case DBHIP_T_U8: return ((const uint8_t*)p)[j];
    case DBHIP_T_U16: return ((const uint16_t*)p)[j];
    case DBHIP_T_U32: return ((const uint32_t*)p)[j];
    case DBHIP_T_U64: return ((const uint64_t*)p)[j];
    case DBHIP_T_F32: return (uint64_t)__double_as_longlong((double)((const float*)p)[j]);
    default: return ((const uint64_t*)p)[j];   // F64 bits
  }
}
__device__ __forceinline__ void cast_store(void* p, int type, int64_t i, uint64_t r) {
  switch (type) {
    case DBHIP_T_I8: case DBHIP_T_U8: ((uint8_t*)p)[i] = (uint8_t)r; break;
    case DBHIP_T_I16: case DBHIP_T_U16: ((uint16_t*)p)[i] = (uint16_t)r; break;
    case DBHIP_T_I32: case DBHIP_T_U32: ((uint32_t*)p)[i] = (uint32_t)r; break;
    case DBHIP_T_F32: ((float*)p)[i] = (float)__longlong_as_double((long long)r); break;
    default: ((uint64_t*)p)[i] = r; break;
  }
}

__global__ __launch_bounds__(256) void cast_kernel(CastParams P) {
  const int lane = threadIdx.x & 63;
  const int scls = type_class(P.src_type), dcls = type_class(P.dst_type);
  const int sbits = type_bits(P.src_type), dbits = type_bits(P.dst_type);
  const int64_t nchunks = (P.n + 255) >> 8;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t c = wave; c < nchunks; c += nwaves) {
    uint64_t w[4];
    bool in[4], valid[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int64_t i = (c << 8) + 64 * k + lane;
      in[k] = i < P.n;
      const int64_t j = P.is_scalar ? 0 : (in[k] ? i : 0);
      w[k] = P.n > 0 ? cast_load(P.src, P.src_type, j) : 0;
      valid[k] = in[k] && (!P.valid || bit_get(P.valid, P.voff + j));
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int64_t i = (c << 8) + 64 * k + lane;
      uint64_t r;
      bool some = true;
      if (dcls == CLS_FLOAT) {   // int -> float, float -> float: always Some(`as`), one rounding
        double d;
        if (scls == CLS_FLOAT) d = __longlong_as_double((long long)w[k]);
        else if (dbits == 32) d = scls == CLS_SIGNED ? (double)(float)(int64_t)w[k] : (double)(float)w[k];
        else d = scls == CLS_SIGNED ? (double)(int64_t)w[k] : (double)w[k];
        r = (uint64_t)__double_as_longlong(d);
      } else if (scls != CLS_FLOAT) {   // int -> int
        if (!P.lossless) {
          if (dcls == CLS_SIGNED) {
            const int64_t lo = dbits == 64 ? INT64_MIN : -(1LL << (dbits - 1)), hi = dbits == 64 ? INT64_MAX : ((1LL << (dbits - 1)) - 1);
            some = scls == CLS_SIGNED ? ((int64_t)w[k] >= lo && (int64_t)w[k] <= hi) : (w[k] <= (uint64_t)hi);
          } else {
            const uint64_t hi = dbits == 64 ? UINT64_MAX : ((1ULL << dbits) - 1);
            some = scls == CLS_SIGNED ? ((int64_t)w[k] >= 0 && w[k] <= hi) : (w[k] <= hi);
          }
        }
        r = wrap_to(w[k], dbits, dcls == CLS_SIGNED);
      } else {   // float -> int: the round cast
        double x = __longlong_as_double((long long)w[k]);
        int fbits = sbits;
        if (P.rounding) { x = round(x); fbits = 64; }   // AsPrimitive::<f64>::as_(val).round(), then cast::<f64, Dest>
        const bool wider = fbits > dbits;
        if (dcls == CLS_SIGNED) {
          const double mn = -ldexp(1.0, dbits - 1), mx1 = ldexp(1.0, dbits - 1);
          some = wider ? (x > mn - 1.0 && x < mx1) : (x >= mn && x < mx1);
        } else {
          some = x > -1.0 && x < ldexp(1.0, dbits);
        }
        r = f64_to_int_sat(x, P.dst_type);
      }
      if (!some) r = 0;   // DestType::default()
      if (in[k]) cast_store(P.out, P.dst_type, i, r);
      if (P.is_try) {
        const uint64_t m = __ballot(valid[k] && some);
        if (lane == 0 && (c << 8) + 64 * k < P.n) P.bitmap_words[(c << 2) + k] = m;
      } else if (!some && valid[k]) {
        if (P.bitmap_words) atomicAnd((unsigned long long*)&P.bitmap_words[i >> 6], ~(1ULL << (i & 63)));
        if (P.err_count) atomicAdd(P.err_count, 1ULL);
      }
    }
  }
}

bool cast_lossless(int s, int d) {   // NumberDataType::can_lossless_cast_to (types/number.rs:426-443)
  if (s == d) return true;
  const bool sf = type_class(s) == CLS_FLOAT, df = type_class(d) == CLS_FLOAT;
  const int sb = type_bits(s), db = type_bits(d);
  if (sf && df) return sb <= db;
  if (sf) return false;
  if (df) return sb < db;
  const bool ss = type_class(s) == CLS_SIGNED, ds = type_class(d) == CLS_SIGNED;
  if (ss == ds) return sb <= db;
  if (!ss && ds) return sb < 64 && sb * 2 <= db;
  return false;
}
}  // namespace

extern "C" {

int32_t dbhip_arith_result_type(int32_t op, int32_t lhs_type, int32_t rhs_type) {
  return coerce(op, lhs_type, rhs_type);
}

int32_t dbhip_arith(int32_t op, const dbhip_col* lhs, const dbhip_col* rhs, int64_t n,
                    int32_t out_type, void* out, uint8_t* err_bitmap, uint64_t* err_count_dev,
                    void* stream) {
  DBHIP_REQUIRE(lhs && rhs && (out || n == 0), "dbhip_arith: NULL argument");
  DBHIP_REQUIRE(n >= 0, "dbhip_arith: negative n");
  int want = coerce(op, lhs->type, rhs->type);
  if (want < 0 || want != out_type) {
    set_error("dbhip_arith: op %d on types (%d,%d) yields type %d, caller passed %d", op, lhs->type,
              rhs->type, want, out_type);
    return DBHIP_ERR_INVALID;
  }
  hipStream_t s = resolve_stream(stream);
  if (err_bitmap) DBHIP_CHECK(hipMemsetAsync(err_bitmap, 0xFF, (size_t)ceil_div(n, 32) * 4, s));
  if (n == 0) return DBHIP_OK;
  ArithParams p;
  p.a = lhs->data; p.b = rhs->data; p.out = out;
  p.a_validity = lhs->validity; p.b_validity = rhs->validity;
  p.a_voff = lhs->validity_offset; p.b_voff = rhs->validity_offset;
  p.n = n;
  p.err_words = (uint32_t*)err_bitmap;
  p.err_count = (unsigned long long*)err_count_dev;
  p.a_type = lhs->type; p.b_type = rhs->type; p.out_type = out_type;
  p.a_scalar = lhs->is_scalar; p.b_scalar = rhs->is_scalar;
  p.op = op;
  p.m_type = coerce(100, lhs->type, rhs->type);
  int grid = grid_for(ceil_div(n, 4), 256, 1024);
  hipLaunchKernelGGL(arith_kernel, dim3(grid), dim3(256), 0, s, p);
  DBHIP_LAUNCH_CHECK();
  return DBHIP_OK;
}

int32_t dbhip_sum_a_plus_b_mul_c_i64(const int64_t* a, const int64_t* b, const int64_t* c,
                                     int64_t n, int64_t* out_sum_dev, void* stream) {
  DBHIP_REQUIRE(out_sum_dev && (n == 0 || (a && b && c)), "dbhip_sum_a_plus_b_mul_c_i64: NULL argument");
  DBHIP_REQUIRE((((uintptr_t)a | (uintptr_t)b | (uintptr_t)c) & 15) == 0,
                "dbhip_sum_a_plus_b_mul_c_i64: columns must be 16-byte aligned");
  if (n == 0) return DBHIP_OK;
  hipStream_t s = resolve_stream(stream);
  int grid = grid_for(ceil_div(n, 8), 256, 512);
  kernel_timer_start(s);
  hipLaunchKernelGGL(sum_abc_kernel, dim3(grid), dim3(256), 0, s, a, b, c, n,
                     (unsigned long long*)out_sum_dev);
  kernel_timer_stop(s);
  DBHIP_LAUNCH_CHECK();
  return DBHIP_OK;
}

int32_t dbhip_sum(const dbhip_col* col, int64_t n, void* out_sum_dev, void* stream) {
  DBHIP_REQUIRE(col && out_sum_dev, "dbhip_sum: NULL argument");
  int cls = type_class(col->type);
  DBHIP_REQUIRE(cls >= 0 && !col->is_scalar, "dbhip_sum: unsupported column");
  if (n == 0) return DBHIP_OK;
  hipStream_t s = resolve_stream(stream);
  SumParams p{col->data, col->validity, col->validity_offset, n, out_sum_dev, col->type};
  int grid = grid_for(ceil_div(n, 16), 256);
  if (cls == CLS_FLOAT) {
    double* partials = (double*)scratch(sizeof(double) * grid, 0, s);
    if (!partials) return DBHIP_ERR_HIP;
    hipLaunchKernelGGL(sum_f64_partial_kernel, dim3(grid), dim3(256), 0, s, p, partials);
    hipLaunchKernelGGL(sum_f64_final_kernel, dim3(1), dim3(64), 0, s, partials, grid,
                       (double*)out_sum_dev);
  } else {
    hipLaunchKernelGGL(sum_int_kernel, dim3(grid), dim3(256), 0, s, p);
  }
  DBHIP_LAUNCH_CHECK();
  return DBHIP_OK;
}

int32_t dbhip_cast(const dbhip_col* src, int32_t dst_type, int32_t is_try, int32_t rounding_mode, int64_t n, void* out,
                   uint8_t* bitmap, uint64_t* err_count_dev, void* stream) {
  DBHIP_REQUIRE(src, "dbhip_cast: NULL column");
  auto number = [](int t) { return t >= DBHIP_T_I8 && t <= DBHIP_T_F64; };
  if (!number(src->type) || !number(dst_type)) {
    set_error("dbhip_cast: number types only (got %d -> %d); decimals, dates and strings stay with the CPU functions", src->type, dst_type);
    return DBHIP_ERR_UNSUPPORTED;
  }
  if (n == 0) return DBHIP_OK;
  DBHIP_REQUIRE(out && src->data, "dbhip_cast: NULL buffer");
  DBHIP_REQUIRE(!is_try || bitmap, "dbhip_cast: try_ casts need the validity output");
  DBHIP_REQUIRE(!bitmap || ((uintptr_t)bitmap & 7) == 0, "dbhip_cast: the bitmap must be 8-byte aligned (whole 64-bit words are written)");
  hipStream_t s = resolve_stream(stream);
  if (!is_try && bitmap) DBHIP_CHECK(hipMemsetAsync(bitmap, 0xFF, (size_t)ceil_div(n, 64) * 8, s));
  CastParams P;
  P.src = src->data; P.valid = src->validity; P.voff = src->validity_offset; P.src_type = src->type; P.dst_type = dst_type;
  P.is_scalar = src->is_scalar; P.is_try = is_try; P.rounding = rounding_mode; P.lossless = cast_lossless(src->type, dst_type);
  P.n = n; P.out = out; P.bitmap_words = (uint64_t*)bitmap; P.err_count = (unsigned long long*)err_count_dev;
  hipLaunchKernelGGL(cast_kernel, dim3(grid_for(ceil_div(n, 4), 256, 1024)), dim3(256), 0, s, P);
  DBHIP_LAUNCH_CHECK();
  return DBHIP_OK;
}

}  // extern "C"

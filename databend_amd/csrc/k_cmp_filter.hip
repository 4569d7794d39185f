// k_cmp_filter.hip — comparisons -> Bitmap (a5), bitmap ops, filter ->
// selection vector and take/gather (a6).
// Reference semantics:
//   vectorize_cmp_2_arg + Bitmap::collect_bool   register_comparison.rs:52-96,
//                                                bitmap/immutable.rs:474 (LSB-first)
//   OrderedFloat total order (NaN == NaN, NaN largest)   types/number.rs:47-48
//   FilterExecutor::select -> ascending u32 row ids      filter/filter_executor.rs:81-118
//   DataBlock::take                                      kernels/take.rs:43
#include "dev_common.h"
#include "dev_load.h"
#include "runtime.h"

#include <string.h>

#include <vector>

using namespace dbhip;

namespace {

// -1 / 0 / +1 three-way compare helpers ------------------------------------
__device__ __forceinline__ int cmp3_i64(int64_t a, int64_t b) { return (a > b) - (a < b); }
__device__ __forceinline__ int cmp3_u64(uint64_t a, uint64_t b) { return (a > b) - (a < b); }
__device__ __forceinline__ int cmp3_f64(double a, double b) {
  bool an = a != a, bn = b != b;
  if (an || bn) return (int)an - (int)bn;  // NaN is the largest, NaN == NaN
  return (a > b) - (a < b);
}
__device__ __forceinline__ int cmp3_i128(i128 a, i128 b) { return (a > b) - (a < b); }

__device__ __forceinline__ bool apply_cmp(int op, int c) {
  switch (op) {
    case DBHIP_CMP_EQ: return c == 0;
    case DBHIP_CMP_NOTEQ: return c != 0;
    case DBHIP_CMP_LT: return c < 0;
    case DBHIP_CMP_LTE: return c <= 0;
    case DBHIP_CMP_GT: return c > 0;
    default: return c >= 0;
  }
}

struct View {
  uint32_t len, w1, w2, w3;  // binview/view.rs:30-42: inline bytes in w1..w3 when len<=12,
};                            // else w1=prefix, w2=buffer_idx, w3=offset

__device__ __forceinline__ const uint8_t* view_ptr(const View& v, const View* self,
                                                   const void* const* buffers) {
  if (v.len <= 12) return (const uint8_t*)self + 4;
  return (const uint8_t*)buffers[v.w2] + v.w3;
}

__device__ int cmp3_views(const View* a, const void* const* abuf, const View* b,
                          const void* const* bbuf) {
  View va = *a, vb = *b;
  const uint8_t* pa = view_ptr(va, a, abuf);
  const uint8_t* pb = view_ptr(vb, b, bbuf);
  uint32_t m = va.len < vb.len ? va.len : vb.len;
  for (uint32_t i = 0; i < m; ++i) {
    int d = (int)pa[i] - (int)pb[i];
    if (d) return d < 0 ? -1 : 1;
  }
  return (va.len > vb.len) - (va.len < vb.len);
}

struct CmpParams {
  const void* a;
  const void* b;
  const void* const* abuf;
  const void* const* bbuf;
  uint8_t* out;
  int64_t n;
  int64_t out_bytes;
  int type, a_scalar, b_scalar, op;
};

// Pack the 4 result bits of each lane: 8 lanes -> one u32 (rows are lane*4+k).
__device__ __forceinline__ void store_nibbles(uint8_t* out, int64_t out_bytes, int64_t quad,
                                              uint32_t nib) {
  uint32_t w = nib << (4 * (lane_id() & 7));
  w |= __shfl_xor(w, 1, 64);
  w |= __shfl_xor(w, 2, 64);
  w |= __shfl_xor(w, 4, 64);
  if ((lane_id() & 7) == 0) {
    int64_t byte0 = (quad >> 3) * 4;  // quad index of lane group start / 8 -> u32 index
    if (byte0 + 4 <= out_bytes) {
      *(uint32_t*)(out + byte0) = w;
    } else {
      for (int k = 0; k < 4; ++k)
        if (byte0 + k < out_bytes) out[byte0 + k] = (uint8_t)(w >> (8 * k));
    }
  }
}

// Decimal comparison of two different DecimalSizes / storage classes — DecimalCmp::eval + CmpOp::compare
// (decimal/src/comparison.rs:326-384,407-441): both sides are viewed in T = storage class of calc_size (max leading
// digits + max scale, capped at 38), f_a = 10^(s - s_a), f_b = 10^(s - s_b); values of different sign compare as they
// are, otherwise each side is multiplied by its factor with a checked multiply whose overflow decides the order.
struct CmpDecParams {
  const void* a;
  const void* b;
  uint8_t* out;
  int64_t n, out_bytes;
  int a_wide, b_wide, a_scalar, b_scalar, op, t128;
  int fa_pow, fb_pow;  // f_a = 10^fa_pow, f_b = 10^fb_pow
};
__device__ __forceinline__ bool dec_checked_mul(i128 x, i128 f, bool t128, i128* out) {
  if (!t128) {
    const i128 r = x * f;
    if (r > (i128)INT64_MAX || r < (i128)INT64_MIN) return false;
    *out = r;
    return true;
  }
  const bool neg = (x < 0) != (f < 0);
  const u128 ax = x < 0 ? (u128)0 - (u128)x : (u128)x, af = f < 0 ? (u128)0 - (u128)f : (u128)f;
  const u256 pr = u256_mul_128(ax, af);
  const u128 lim = neg ? ((u128)1 << 127) : (((u128)1 << 127) - 1);
  if (pr.hi != 0 || pr.lo > lim) return false;
  *out = neg ? (i128)((u128)0 - pr.lo) : (i128)pr.lo;
  return true;
}
__device__ __forceinline__ int dec_cmp3(i128 a, i128 b, i128 fa, i128 fb, bool t128) {
  const int sa = (a > 0) - (a < 0), sb = (b > 0) - (b < 0);
  if (sa != sb) return (a > b) - (a < b);
  i128 x = a, y = b;
  if (fa != 1 && !dec_checked_mul(a, fa, t128, &x)) return sa > 0 ? 1 : -1;   // a is out of T's range at the common scale
  if (fb != 1 && !dec_checked_mul(b, fb, t128, &y)) return sb > 0 ? -1 : 1;
  return (x > y) - (x < y);
}
__global__ __launch_bounds__(256) void cmp_decimal_kernel(CmpDecParams p) {
  const i128 fa = pow10_i128(p.fa_pow), fb = pow10_i128(p.fb_pow);
  const int64_t n_pad = (p.n + 63) & ~63LL;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_pad; i += (int64_t)gridDim.x * blockDim.x) {
    bool r = false;
    if (i < p.n) {
      const int64_t ja = p.a_scalar ? 0 : i, jb = p.b_scalar ? 0 : i;
      const i128 a = p.a_wide ? ((const i128*)p.a)[ja] : (i128)((const int64_t*)p.a)[ja];
      const i128 b = p.b_wide ? ((const i128*)p.b)[jb] : (i128)((const int64_t*)p.b)[jb];
      r = apply_cmp(p.op, fa == fb ? cmp3_i128(a, b) : dec_cmp3(a, b, fa, fb, p.t128 != 0));
    }
    const uint64_t m = __ballot(r);
    const int lane = lane_id();
    if ((lane & 7) == 0) {
      const int64_t byte = i >> 3;
      if (byte < p.out_bytes) p.out[byte] = (uint8_t)(m >> lane);
    }
  }
}

__global__ __launch_bounds__(256) void cmp_kernel(CmpParams p) {
  const int cls = type_class(p.type);
  const int64_t nquads = (p.n + 3) >> 2;
  // all lanes of a wave iterate together so the shuffles in store_nibbles are convergent
  const int64_t nquads_pad = (nquads + 63) & ~63LL;
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < nquads_pad;
       q += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i0 = q << 2;
    uint32_t nib = 0;
    if (q < nquads) {
      if (p.type == DBHIP_T_DEC128) {
        const i128* a = (const i128*)p.a;
        const i128* b = (const i128*)p.b;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          if (i0 + k < p.n) {
            i128 x = a[p.a_scalar ? 0 : i0 + k], y = b[p.b_scalar ? 0 : i0 + k];
            nib |= (uint32_t)apply_cmp(p.op, cmp3_i128(x, y)) << k;
          }
        }
      } else if (p.type == DBHIP_T_STRING) {
        const View* a = (const View*)p.a;
        const View* b = (const View*)p.b;
        for (int k = 0; k < 4; ++k) {
          if (i0 + k < p.n) {
            int c = cmp3_views(a + (p.a_scalar ? 0 : i0 + k), p.abuf, b + (p.b_scalar ? 0 : i0 + k),
                               p.bbuf);
            nib |= (uint32_t)apply_cmp(p.op, c) << k;
          }
        }
      } else if (p.type == DBHIP_T_BOOL) {
        const uint8_t* a = (const uint8_t*)p.a;
        const uint8_t* b = (const uint8_t*)p.b;
        for (int k = 0; k < 4; ++k) {
          if (i0 + k < p.n) {
            int x = bit_get(a, p.a_scalar ? 0 : i0 + k), y = bit_get(b, p.b_scalar ? 0 : i0 + k);
            nib |= (uint32_t)apply_cmp(p.op, x - y) << k;
          }
        }
      } else {
        uint64_t a[4], b[4];
        load4_wide(p.a, p.type, p.a_scalar, i0, p.n, a);
        load4_wide(p.b, p.type, p.b_scalar, i0, p.n, b);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          int c;
          if (cls == CLS_SIGNED) c = cmp3_i64((int64_t)a[k], (int64_t)b[k]);
          else if (cls == CLS_UNSIGNED) c = cmp3_u64(a[k], b[k]);
          else c = cmp3_f64(__longlong_as_double((long long)a[k]), __longlong_as_double((long long)b[k]));
          nib |= (uint32_t)(apply_cmp(p.op, c) && (i0 + k < p.n)) << k;
        }
      }
    }
    store_nibbles(p.out, p.out_bytes, q & ~7LL, nib);
  }
}

// Narrow numeric types (<= 4 bytes per value): with 4 rows per lane a lane has only 16 bytes in flight and the
// kernel is latency bound (2.7 TB/s on a date column). Here a lane owns 16 consecutive rows — 64 bytes of a
// 4-byte column as four back-to-back 16-byte loads — and writes its 16 result bits as one u16; a wave covers
// 1024 rows = 4 KiB contiguous per operand.
template <typename T>
__device__ __forceinline__ int cmp3_t(T a, T b) {
  if constexpr (!__is_integral(T)) {
    const bool an = a != a, bn = b != b;
    if (an || bn) return (int)an - (int)bn;  // OrderedFloat: NaN largest, NaN == NaN
  }
  return (a > b) - (a < b);
}

template <typename T>
__global__ __launch_bounds__(256) void cmp_narrow_kernel(CmpParams p) {
  constexpr int R = 16;
  const T* a = (const T*)p.a;
  const T* b = (const T*)p.b;
  const int64_t ngroups = (p.n + R - 1) / R;
  const T sa = p.a_scalar ? a[0] : T(0), sb = p.b_scalar ? b[0] : T(0);
  for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < ngroups; g += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i0 = g * R;
    T va[R], vb[R];
    if (i0 + R <= p.n) {
      if (!p.a_scalar) { const VecT<T, R> v = *(const VecT<T, R>*)(a + i0);
#pragma unroll
        for (int k = 0; k < R; ++k) va[k] = v.v[k]; }
      if (!p.b_scalar) { const VecT<T, R> v = *(const VecT<T, R>*)(b + i0);
#pragma unroll
        for (int k = 0; k < R; ++k) vb[k] = v.v[k]; }
    } else {
#pragma unroll
      for (int k = 0; k < R; ++k) {
        va[k] = (!p.a_scalar && i0 + k < p.n) ? a[i0 + k] : T(0);
        vb[k] = (!p.b_scalar && i0 + k < p.n) ? b[i0 + k] : T(0);
      }
    }
    uint32_t bits = 0;
#pragma unroll
    for (int k = 0; k < R; ++k) {
      const T x = p.a_scalar ? sa : va[k], y = p.b_scalar ? sb : vb[k];
      bits |= (uint32_t)(apply_cmp(p.op, cmp3_t<T>(x, y)) && (i0 + k < p.n)) << k;
    }
    const int64_t byte0 = g * 2;
    if (byte0 + 2 <= p.out_bytes) *(uint16_t*)(p.out + byte0) = (uint16_t)bits;
    else if (byte0 < p.out_bytes) p.out[byte0] = (uint8_t)bits;
  }
}

template <typename T>
void launch_cmp_narrow(const CmpParams& p, hipStream_t s) {
  hipLaunchKernelGGL(cmp_narrow_kernel<T>, dim3(grid_for(ceil_div(p.n, 16), 256, 1024)), dim3(256), 0, s, p);
}

__global__ __launch_bounds__(256) void bitmap_binary_kernel(const uint8_t* a, const uint8_t* b,
                                                            uint8_t* out, int64_t nbytes, int64_t n,
                                                            int is_or) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nbytes;
       i += (int64_t)gridDim.x * blockDim.x) {
    uint8_t v = is_or == 1 ? (a[i] | b[i]) : (is_or == 2 ? (a[i] & (uint8_t)~b[i]) : (a[i] & b[i]));
    if (i == nbytes - 1 && (n & 7)) v &= (uint8_t)((1u << (n & 7)) - 1);
    out[i] = v;
  }
}

// Read up to 64 bits starting at absolute bit `pos` of `bm`; bits past `end` are 0.
__device__ __forceinline__ uint64_t load_bits64(const uint8_t* bm, int64_t pos, int64_t end) {
  if (pos >= end) return 0;
  int64_t nb = end - pos;
  if (nb > 64) nb = 64;
  uintptr_t addr = (uintptr_t)(bm + (pos >> 3));
  const uint64_t* w = (const uint64_t*)(addr & ~(uintptr_t)7);
  int sh = (int)((addr & 7) * 8 + (pos & 7));
  uint64_t v = w[0] >> sh;
  if (sh + nb > 64) v |= w[1] << (64 - sh);
  if (nb < 64) v &= (1ULL << nb) - 1;
  return v;
}

// bitmap[idx[i]] = 1 (OR into the words: the pair list of a join names a probe row once per match; consecutive pairs mostly
// hit the same word, so lanes with the same word combine their bits first and one lane issues the atomic)
__global__ __launch_bounds__(256) void bitmap_set_indices_kernel(const uint32_t* idx, int64_t n, uint32_t* words, int64_t nbits) {
  for (int64_t base = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) & ~63LL; base < n; base += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = base + lane_id();
    const bool live = i < n && (int64_t)idx[i < n ? i : 0] < nbits;
    const uint32_t id = live ? idx[i] : 0xFFFFFFFFu;
    const uint32_t w = id >> 5;
    uint32_t bit = live ? 1u << (id & 31) : 0;
    // runs of equal words are adjacent in a sorted pair list: fold a lane into its left neighbour while the word is the same
    const uint32_t wl = __shfl_up(w, 1, 64);
    const bool head = lane_id() == 0 || wl != w;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t ob = __shfl_down(bit, off, 64), ow = __shfl_down(w, off, 64);
      if (lane_id() + off < 64 && ow == w) bit |= ob;
    }
    if (live && head) atomicOr(&words[w], bit);
  }
}

__global__ __launch_bounds__(256) void bitmap_count_kernel(const uint8_t* bm, int64_t off, int64_t n,
                                                           unsigned long long* out) {
  const int64_t nwords = (n + 63) >> 6;
  uint64_t acc = 0;
  for (int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; w < nwords;
       w += (int64_t)gridDim.x * blockDim.x)
    acc += __popcll(load_bits64(bm, off + w * 64, off + n));
  acc = wave_sum_u64(acc);
  if (lane_id() == 0 && acc) atomicAdd(out, (unsigned long long)acc);
}

// ---- filter_select: 3 phases --------------------------------------------------
// A block owns SEL_WORDS_PER_BLOCK 64-bit words (= 16384 rows).
#define SEL_WORDS_PER_BLOCK 256

__global__ __launch_bounds__(256) void sel_count_kernel(const uint8_t* bm, int64_t off, int64_t n,
                                                        uint32_t* block_counts) {
  const int64_t w = (int64_t)blockIdx.x * SEL_WORDS_PER_BLOCK + threadIdx.x;
  uint64_t c = __popcll(load_bits64(bm, off + w * 64, off + n));
  c = wave_sum_u64(c);
  __shared__ uint32_t part[4];
  if (lane_id() == 0) part[threadIdx.x >> 6] = (uint32_t)c;
  __syncthreads();
  if (threadIdx.x == 0) block_counts[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}

// single-block exclusive scan of block counts (u32 counts, u64 offsets)
__global__ __launch_bounds__(1024) void sel_scan_kernel(const uint32_t* counts, uint64_t* offsets,
                                                        int64_t nblocks, unsigned long long* total) {
  __shared__ uint64_t wave_tot[16];
  __shared__ uint64_t carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int64_t base = 0; base < nblocks; base += 1024) {
    int64_t i = base + threadIdx.x;
    uint64_t v = i < nblocks ? counts[i] : 0;
    uint64_t incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      uint64_t t = __shfl_up(incl, d, 64);
      if (lane_id() >= d) incl += t;
    }
    if (lane_id() == 63) wave_tot[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint64_t wbase = 0;
    for (int k = 0; k < (threadIdx.x >> 6); ++k) wbase += wave_tot[k];
    uint64_t c = carry;
    if (i < nblocks) offsets[i] = c + wbase + incl - v;
    __syncthreads();
    if (threadIdx.x == 1023) carry = c + wbase + incl;
    __syncthreads();
  }
  if (threadIdx.x == 0) *total = carry;
}

__global__ __launch_bounds__(256) void sel_write_kernel(const uint8_t* bm, int64_t off, int64_t n,
                                                        const uint64_t* block_offsets,
                                                        uint32_t* out_sel) {
  const int wave = threadIdx.x >> 6, lane = lane_id();
  const int64_t w = (int64_t)blockIdx.x * SEL_WORDS_PER_BLOCK + threadIdx.x;
  const uint64_t bits = load_bits64(bm, off + w * 64, off + n);
  const uint32_t cnt = __popcll(bits);
  // exclusive scan of cnt inside the block
  uint32_t incl = cnt;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    uint32_t t = __shfl_up(incl, d, 64);
    if (lane >= d) incl += t;
  }
  __shared__ uint32_t wave_tot[4];
  if (lane == 63) wave_tot[wave] = incl;
  __syncthreads();
  uint32_t wbase = 0;
  for (int k = 0; k < wave; ++k) wbase += wave_tot[k];
  const uint64_t my_off = block_offsets[blockIdx.x] + wbase + incl - cnt;
  // wave-cooperative expansion: word j of this wave is handled by all 64 lanes,
  // lane l tests bit l -> coalesced stores of ascending row ids
  for (int j = 0; j < 64; ++j) {
    uint64_t wj = __shfl(bits, j, 64);
    if (wj == 0) continue;
    uint64_t oj = __shfl(my_off, j, 64);
    if ((wj >> lane) & 1) {
      uint32_t rank = __popcll(wj & ((1ULL << lane) - 1));
      int64_t row = ((int64_t)blockIdx.x * SEL_WORDS_PER_BLOCK + wave * 64 + j) * 64 + lane;
      out_sel[oj + rank] = (uint32_t)row;
    }
  }
}

// ---- Selector (filter/selector.rs:64-330, select_value/select_column*.rs): a predicate evaluated on the rows of a SELECTION
// (SelectStrategy::True / False: the true or the false list of the step before; All: every row), the rows split in order into
// a true list and — when the caller short-circuits an OR — a false list. ----
// one 64-bit element of a column widened like load4_wide (sign / zero extension, f32 -> f64 bits)
__device__ __forceinline__ uint64_t load1_wide(const void* p, int type, int64_t i) {
  switch (type) {
    case DBHIP_T_I8: return (uint64_t)(int64_t)((const int8_t*)p)[i];
    case DBHIP_T_I16: return (uint64_t)(int64_t)((const int16_t*)p)[i];
    case DBHIP_T_I32: case DBHIP_T_DATE: return (uint64_t)(int64_t)((const int32_t*)p)[i];
    case DBHIP_T_I64: case DBHIP_T_TIMESTAMP: case DBHIP_T_DEC64: return (uint64_t)((const int64_t*)p)[i];
    case DBHIP_T_U8: return ((const uint8_t*)p)[i];
    case DBHIP_T_U16: return ((const uint16_t*)p)[i];
    case DBHIP_T_U32: return ((const uint32_t*)p)[i];
    case DBHIP_T_U64: return ((const uint64_t*)p)[i];
    case DBHIP_T_F32: return (uint64_t)__double_as_longlong((double)((const float*)p)[i]);
    default: return (uint64_t)__double_as_longlong(((const double*)p)[i]);
  }
}
struct SelArgs {
  const void* a; const void* b;                 // operands (b unused for a Boolean column)
  const uint8_t* av; const uint8_t* bv;         // validities (NULL = none)
  int64_t avo, bvo;
  int type, a_scalar, b_scalar, op;             // op < 0: `a` is a Boolean column (select_boolean_column)
  const uint32_t* sel_in;                       // NULL = SelectStrategy::All
  int64_t n;                                    // entries of sel_in, or rows
  uint64_t* bits;                               // out: one bit per ENTRY (whole words)
};
__global__ __launch_bounds__(256) void select_eval_kernel(SelArgs A) {
  const int cls = type_class(A.type);
  const int64_t n_pad = (A.n + 63) & ~63LL;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_pad; i += (int64_t)gridDim.x * 256) {
    bool ret = false;
    if (i < A.n) {
      const int64_t row = A.sel_in ? (int64_t)A.sel_in[i] : i;
      const int64_t ra = A.a_scalar ? 0 : row, rb = A.b_scalar ? 0 : row;
      bool valid = (!A.av || bit_get(A.av, A.avo + ra)) && (!A.bv || bit_get(A.bv, A.bvo + rb));
      if (valid) {
        if (A.op < 0) ret = bit_get((const uint8_t*)A.a, ra);
        else if (A.type == DBHIP_T_DEC128) ret = apply_cmp(A.op, cmp3_i128(((const i128*)A.a)[ra], ((const i128*)A.b)[rb]));
        else if (A.type == DBHIP_T_BOOL) ret = apply_cmp(A.op, (int)bit_get((const uint8_t*)A.a, ra) - (int)bit_get((const uint8_t*)A.b, rb));
        else {
          const uint64_t x = load1_wide(A.a, A.type, ra), y = load1_wide(A.b, A.type, rb);
          int c;
          if (cls == CLS_SIGNED) c = cmp3_i64((int64_t)x, (int64_t)y);
          else if (cls == CLS_UNSIGNED) c = cmp3_u64(x, y);
          else c = cmp3_f64(__longlong_as_double((long long)x), __longlong_as_double((long long)y));
          ret = apply_cmp(A.op, c);
        }
      }
    }
    const uint64_t m = __ballot(ret);
    if (lane_id() == 0) A.bits[i >> 6] = m;
  }
}
// sel_write_kernel for a bitmap over ENTRIES: entry e of the selection goes to the true list (bit set) or the false list, both in
// entry order; `sel_in` maps entries to row ids (NULL: the entry is the row)
__global__ __launch_bounds__(256) void select_split_kernel(const uint64_t* bits, int64_t n, const uint64_t* block_offsets, const uint32_t* sel_in,
                                                           uint32_t* out_true, uint32_t* out_false) {
  const int wave = threadIdx.x >> 6, lane = lane_id();
  const int64_t w = (int64_t)blockIdx.x * SEL_WORDS_PER_BLOCK + threadIdx.x;
  const int64_t nwords = (n + 63) >> 6;
  uint64_t word = w < nwords ? bits[w] : 0;
  const uint32_t cnt = __popcll(word);
  uint32_t incl = cnt;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    uint32_t t = __shfl_up(incl, d, 64);
    if (lane >= d) incl += t;
  }
  __shared__ uint32_t wave_tot[4];
  if (lane == 63) wave_tot[wave] = incl;
  __syncthreads();
  uint32_t wbase = 0;
  for (int k = 0; k < wave; ++k) wbase += wave_tot[k];
  const uint64_t my_off = block_offsets[blockIdx.x] + wbase + incl - cnt;   // true entries before this word
  for (int j = 0; j < 64; ++j) {
    const uint64_t wj = __shfl(word, j, 64);
    const uint64_t oj = __shfl(my_off, j, 64);
    const int64_t e = (((int64_t)blockIdx.x * SEL_WORDS_PER_BLOCK + wave * 64 + j) << 6) + lane;
    if (e >= n) continue;
    const uint32_t row = sel_in ? sel_in[e] : (uint32_t)e;
    const uint32_t before = __popcll(wj & ((1ULL << lane) - 1));
    if ((wj >> lane) & 1) out_true[oj + before] = row;
    else if (out_false) out_false[(uint64_t)(e - lane) - oj + (lane - before)] = row;   // false entries before e = e - true entries before e
  }
}

template <typename T>
__global__ __launch_bounds__(256) void take_kernel(const T* __restrict__ src,
                                                   const uint32_t* __restrict__ sel, int64_t n,
                                                   T* __restrict__ out) {
  const int64_t nquads = (n + 3) >> 2;
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < nquads;
       q += (int64_t)gridDim.x * blockDim.x) {
    int64_t i0 = q << 2;
    if (i0 + 4 <= n) {
      VecT<uint32_t, 4> s = *(const VecT<uint32_t, 4>*)(sel + i0);
      T v0 = src[s.v[0]], v1 = src[s.v[1]], v2 = src[s.v[2]], v3 = src[s.v[3]];
      out[i0] = v0; out[i0 + 1] = v1; out[i0 + 2] = v2; out[i0 + 3] = v3;
    } else {
      for (int64_t i = i0; i < n; ++i) out[i] = src[sel[i]];
    }
  }
}

struct alignas(16) B16 {
  uint64_t a, b;
};

__global__ __launch_bounds__(256) void take_bitmap_kernel(const uint8_t* src, int64_t off,
                                                          const uint32_t* sel, int64_t n,
                                                          uint8_t* out, int64_t out_bytes) {
  const int64_t n_pad = (n + 63) & ~63LL;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_pad;
       i += (int64_t)gridDim.x * blockDim.x) {
    bool b = i < n && bit_get(src, off + sel[i]);
    uint64_t m = __ballot(b);
    if (lane_id() == 0) {
      int64_t byte0 = (i >> 6) * 8;
      for (int k = 0; k < 8; ++k)
        if (byte0 + k < out_bytes) out[byte0 + k] = (uint8_t)(m >> (8 * k));
    }
  }
}

// ---- take_ranges / take_compacted_indices / take_chunks (kernels/take_ranges.rs:40, take_compact.rs:38, take_chunks.rs:70-190) ----
// Ranges and repeat lists are run-length forms of a selection vector: item r covers output rows [off[r], off[r + 1]). One
// thread per output row finds its item by binary search over the offsets (a few thousand items at most per block) and writes
// the source row id — after that every column is gathered by the ordinary take kernels.
__global__ __launch_bounds__(256) void sel_expand_kernel(const uint32_t* first, const uint32_t* off, int n_items, int repeat,
                                                         int64_t num_rows, uint32_t* out_sel) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < num_rows; i += (int64_t)gridDim.x * blockDim.x) {
    int lo = 0, hi = n_items - 1;
    while (lo < hi) {  // last item whose offset is <= i
      const int mid = (lo + hi + 1) >> 1;
      if ((int64_t)off[mid] <= i) lo = mid; else hi = mid - 1;
    }
    out_sel[i] = repeat ? first[lo] : first[lo] + (uint32_t)(i - off[lo]);
  }
}
// (block, row) pairs -> rows of several source blocks (take_blocks / take_column_vec): out[i] = blocks[pair[i].block][pair[i].row]
__global__ __launch_bounds__(256) void take_chunks_kernel(const void* const* blocks, int elem, const uint32_t* pairs, int64_t n, void* out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const uint32_t b = pairs[2 * i], r = pairs[2 * i + 1];
    const uint8_t* src = (const uint8_t*)blocks[b];
    switch (elem) {
      case 1: ((uint8_t*)out)[i] = src[r]; break;
      case 2: ((uint16_t*)out)[i] = ((const uint16_t*)src)[r]; break;
      case 4: ((uint32_t*)out)[i] = ((const uint32_t*)src)[r]; break;
      case 8: ((uint64_t*)out)[i] = ((const uint64_t*)src)[r]; break;
      default: ((u128*)out)[i] = ((const u128*)src)[r]; break;
    }
  }
}
__global__ __launch_bounds__(256) void take_chunks_bitmap_kernel(const uint8_t* const* blocks, const uint32_t* pairs, int64_t n, uint8_t* out,
                                                                 int64_t out_bytes) {
  const int64_t n_pad = (n + 63) & ~63LL;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_pad; i += (int64_t)gridDim.x * blockDim.x) {
    // a NULL block pointer stands for a column without validity: every row valid
    const bool b = i < n && (blocks[pairs[2 * i]] == nullptr || bit_get(blocks[pairs[2 * i]], pairs[2 * i + 1]));
    const uint64_t m = __ballot(b);
    if (lane_id() == 0) {
      const int64_t byte0 = (i >> 6) * 8;
      for (int k = 0; k < 8; ++k)
        if (byte0 + k < out_bytes) out[byte0 + k] = (uint8_t)(m >> (8 * k));
    }
  }
}

}  // namespace

// take for the nullable side of an outer join (left_join.rs:185-260: matched rows take the build columns wrapped with a true
// validity, unmatched probe rows get a null_block): idx == 0xFFFFFFFF -> a zero value and validity 0; otherwise the source row
// and its validity (true when the source has none). One validity word per wave (ballot).
// take for selections that are LOCALLY DENSE (a TransformFilter's ascending selection, the probe side of a join's pairs). A gather
// asks for one 64-byte sector per element although neighbouring elements share lines with rows nobody wants; reading the covered
// source range with coalesced 16-byte loads into LDS and picking the elements there moves whole lines at the streaming rate
// instead (r03, 14.6 M of 150 M rows, 8-byte values: 1.67 ms as a gather, 0.25 ms like this). The unit is a WAVE and 64 entries:
// no barrier, no workgroup reduction (the first version reduced min / max over a workgroup's 512 entries through LDS and two
// barriers per chunk — twice the time of a plain gather on the 98 %-dense selection of Q1's literal plan — and its 61 KB window
// left 2 waves per SIMD; 30 KB leave 4). Per 64 entries:
//   span <= 128 rows (at least half of the rows are wanted): a plain gather is already coalesced;
//   span <= LDS_ROWS / 4: staged through the wave's quarter of the LDS window;
//   else (sparse or unordered): a plain gather. Any selection is handled; the decision is per 64 entries.
// one wave, one column, 64 entries (s0 = this lane's entry, [lo, hi] = the range the 64 entries span): through the wave's LDS
// window `win` (WROWS elements) when the entries are neither dense nor sparse, else a plain gather
template <typename T>
__device__ __forceinline__ uint32_t take_window_shift(const T* src, uint32_t lo) {
  return (uint32_t)(((uintptr_t)(src + (uint64_t)lo) & 15) / sizeof(T));
}
template <typename T, int WROWS>
__device__ __forceinline__ void take_wave_step(const T* __restrict__ src, T* __restrict__ out, T* win, uint32_t s0, uint32_t lo, uint32_t hi,
                                               int64_t i0, int64_t n, int lane) {
  constexpr uint32_t PER16 = 16 / sizeof(T);            // elements per 16-byte vector
  // The window starts on the 16-byte ADDRESS boundary at or below src[lo] (a column is only known to be aligned to its element
  // size: a slice view of a 16-byte aligned buffer is not): every vector read then lies inside an aligned 16-byte granule that
  // holds at least one wanted element, so nothing outside the granules of [lo, hi] is touched — no read past an unpadded
  // buffer's last granule (r03 rounded the INDEX down and the count up, which assumed 16-byte aligned, padded columns).
  const uint32_t shift = take_window_shift<T>(src, lo);   // elements between the boundary and src[lo]
  const uint32_t span = hi - lo + 1 + shift;
  if (span > 128u && span <= (uint32_t)WROWS - PER16) {
    // the covered range as 16-byte vectors, every load of the range in flight before the first LDS store
    const uint32_t nvec = (span + PER16 - 1) / PER16;
    typedef uint32_t tw_u32x4 __attribute__((ext_vector_type(4)));
    const tw_u32x4* gsrc = (const tw_u32x4*)((const T*)(src + (uint64_t)lo) - shift);
    tw_u32x4* lwin = (tw_u32x4*)win;
    constexpr int MAXV = (WROWS / (int)PER16 + 63) / 64;   // vectors per lane at most
    tw_u32x4 reg[MAXV];
#pragma unroll
    for (int u = 0; u < MAXV; ++u) { const uint32_t v = lane + 64 * u; reg[u] = __builtin_nontemporal_load(gsrc + (v < nvec ? v : nvec - 1)); }   // (always defined: a conditional load left the compiler copying the array around, 250 VGPRs)
#pragma unroll
    for (int u = 0; u < MAXV; ++u) { const uint32_t v = lane + 64 * u; if (v < nvec) lwin[v] = reg[u]; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (i0 < n) out[i0] = win[s0 - lo + shift];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();   // the window is rewritten by the next step
  } else {
    if (i0 < n) out[i0] = src[s0];
  }
}
// this lane's entry of chunk c and the range [lo, hi] of the wave's 64 entries
__device__ __forceinline__ uint32_t take_wave_range(const uint32_t* __restrict__ sel, int64_t i0, int64_t n, uint32_t* lo_out, uint32_t* hi_out) {
  const uint32_t s0 = i0 < n ? sel[i0] : 0xFFFFFFFFu;
  uint32_t lo = s0;
  uint32_t hi = (i0 < n ? s0 : 0u);
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const uint32_t a = __shfl_xor(lo, off, 64), b = __shfl_xor(hi, off, 64);
    lo = a < lo ? a : lo;
    hi = b > hi ? b : hi;
  }
  *lo_out = lo; *hi_out = hi;
  return s0;
}

template <typename T, int LDS_ROWS>
__global__ __launch_bounds__(256) void take_window_kernel(const T* __restrict__ src, const uint32_t* __restrict__ sel, int64_t n, T* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) T win_all[LDS_ROWS];
  constexpr int WROWS = LDS_ROWS / 4;
  constexpr uint32_t PER16 = 16 / sizeof(T);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  T* win = win_all + wave * WROWS;
  // FOUR steps of 64 entries per iteration: their selection words are requested together, and the steps that are plain gathers
  // (dense or sparse entries) have their four gathers in flight together — one step per iteration left a wave with two dependent
  // round trips per 64 entries and the 98 %-dense takes of Q1's literal plan at a third of the rate of r02's gather kernel
  const int64_t ngroups = (n + 255) / 256;
  for (int64_t c = (int64_t)blockIdx.x * 4 + wave; c < ngroups; c += (int64_t)gridDim.x * 4) {
    uint32_t s0[4], lo[4], hi[4];
    int64_t i0[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      i0[u] = c * 256 + u * 64 + lane;
      s0[u] = i0[u] < n ? sel[i0[u]] : 0xFFFFFFFFu;
    }
    // A plain gather is right for ANY entries, so "dense" may be guessed from the step's first and last entry alone (an ascending
    // selection — a filter's — with last - first <= 128): such groups skip the min / max reductions altogether.
    bool quick = c * 256 + 256 <= n;
#pragma unroll
    for (int u = 0; u < 4; ++u)
      quick &= (uint32_t)(__builtin_amdgcn_readlane((int)s0[u], 63) - __builtin_amdgcn_readlane((int)s0[u], 0)) <= 128u;
    if (quick) {
      T v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = src[s0[u]];
#pragma unroll
      for (int u = 0; u < 4; ++u) out[i0[u]] = v[u];
      continue;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      uint32_t l = s0[u], h = i0[u] < n ? s0[u] : 0u;
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) {
        const uint32_t a = __shfl_xor(l, off, 64), b2 = __shfl_xor(h, off, 64);
        l = a < l ? a : l;
        h = b2 > h ? b2 : h;
      }
      lo[u] = l; hi[u] = h;
    }
    bool all_direct = true;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const uint32_t span = hi[u] - lo[u] + 1 + take_window_shift<T>(src, lo[u]);
      all_direct &= !(span > 128u && span <= (uint32_t)WROWS - PER16);   // (wave-uniform)
    }
    if (all_direct) {
      T v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) if (i0[u] < n) v[u] = src[s0[u]];
#pragma unroll
      for (int u = 0; u < 4; ++u) if (i0[u] < n) out[i0[u]] = v[u];
    } else {
      // some step goes through the window: the four steps one after the other through ONE copy of the step code (unrolled, its
      // staging registers quadruple: 250 VGPRs, one wave per SIMD); the selection words come from the cache again
#pragma clang loop unroll(disable)
      for (int u = 0; u < 4; ++u) {
        const int64_t i = c * 256 + u * 64 + lane;
        uint32_t l, h;
        const uint32_t su = take_wave_range(sel, i, n, &l, &h);
        take_wave_step<T, WROWS>(src, out, win, su, l, h, i, n, lane);
      }
    }
  }
}

// DataBlock::take over SEVERAL columns of a block with one selection (kernels/take.rs:43 takes every column of the block):
// the selection is read once, its range found once per 64 entries, and the columns go through the same wave-private window
// one after the other (r03: the eight takes behind Q3's two probes were eight launches that each re-read the pair list)
constexpr int TAKE_MAX_COLS = 8;
struct TakeCols {
  const void* src[TAKE_MAX_COLS];
  void* out[TAKE_MAX_COLS];
  int elem[TAKE_MAX_COLS];
  int n;
};
__global__ __launch_bounds__(256) void take_block_kernel(TakeCols P, const uint32_t* __restrict__ sel, int64_t n) {
  __shared__ __attribute__((aligned(16))) uint8_t win_bytes[30720];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint8_t* win = win_bytes + wave * 7680;
  const int64_t nchunks = (n + 63) / 64;
  for (int64_t c = (int64_t)blockIdx.x * 4 + wave; c < nchunks; c += (int64_t)gridDim.x * 4) {
    const int64_t i0 = c * 64 + lane;
    uint32_t lo, hi;
    const uint32_t s0 = take_wave_range(sel, i0, n, &lo, &hi);
    for (int k = 0; k < P.n; ++k) {   // (uniform)
      switch (P.elem[k]) {
        case 4: take_wave_step<uint32_t, 1920>((const uint32_t*)P.src[k], (uint32_t*)P.out[k], (uint32_t*)win, s0, lo, hi, i0, n, lane); break;
        case 8: take_wave_step<uint64_t, 960>((const uint64_t*)P.src[k], (uint64_t*)P.out[k], (uint64_t*)win, s0, lo, hi, i0, n, lane); break;
        case 16: take_wave_step<B16, 480>((const B16*)P.src[k], (B16*)P.out[k], (B16*)win, s0, lo, hi, i0, n, lane); break;
        case 2: if (i0 < n) ((uint16_t*)P.out[k])[i0] = ((const uint16_t*)P.src[k])[s0]; break;
        default: if (i0 < n) ((uint8_t*)P.out[k])[i0] = ((const uint8_t*)P.src[k])[s0]; break;
      }
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void take_outer_kernel(const T* __restrict__ src, const uint8_t* __restrict__ src_valid, int64_t src_voff,
                                                         const uint32_t* __restrict__ idx, int64_t n, T* __restrict__ out,
                                                         uint64_t* __restrict__ out_valid_words, uint8_t* __restrict__ out_valid_bytes) {
  const int64_t n_pad = (n + 63) & ~63LL;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_pad; i += (int64_t)gridDim.x * 256) {
    bool valid = false;
    if (i < n) {
      const uint32_t j = idx[i];
      T v{};
      if (j != 0xFFFFFFFFu) {
        v = src[j];
        valid = !src_valid || bit_get(src_valid, src_voff + j);
      }
      out[i] = v;
    }
    const uint64_t m = __ballot(valid);
    if ((threadIdx.x & 63) == 0) {
      const int64_t w = i >> 6;
      if (i + 64 <= n) out_valid_words[w] = m;   // (the buffer is 8-byte aligned and padded: dbhip.h)
      else
        for (int b = 0; b < (int)((n - i + 7) >> 3); ++b) out_valid_bytes[w * 8 + b] = (uint8_t)(m >> (8 * b));
    }
  }
}

namespace dbhip {
int32_t cmp_decimal256(int32_t op, const dbhip_col* lhs, const dbhip_col* rhs, int64_t n, uint8_t* out_bitmap, void* stream);  // k_decimal256.hip
}

extern "C" {

int32_t dbhip_cmp(int32_t op, const dbhip_col* lhs, const dbhip_col* rhs, int64_t n,
                  uint8_t* out_bitmap, void* stream) {
  DBHIP_REQUIRE(lhs && rhs && (out_bitmap || n == 0), "dbhip_cmp: NULL argument");
  DBHIP_REQUIRE(op >= DBHIP_CMP_EQ && op <= DBHIP_CMP_GTE, "dbhip_cmp: bad operator");
  if ((lhs->type == DBHIP_T_DEC256 || rhs->type == DBHIP_T_DEC256)) {  // the Decimal256 class: k_decimal256.hip
    const bool l_any = lhs->type == DBHIP_T_DEC64 || lhs->type == DBHIP_T_DEC128 || lhs->type == DBHIP_T_DEC256;
    const bool r_any = rhs->type == DBHIP_T_DEC64 || rhs->type == DBHIP_T_DEC128 || rhs->type == DBHIP_T_DEC256;
    DBHIP_REQUIRE(l_any && r_any, "dbhip_cmp: a Decimal256 column compares with decimals only (the planner casts the other side)");
    return cmp_decimal256(op, lhs, rhs, n, out_bitmap, stream);
  }
  const bool l_dec = lhs->type == DBHIP_T_DEC64 || lhs->type == DBHIP_T_DEC128, r_dec = rhs->type == DBHIP_T_DEC64 || rhs->type == DBHIP_T_DEC128;
  if (l_dec && r_dec && (lhs->type != rhs->type || lhs->scale != rhs->scale)) {
    // decimals of different DecimalSize: no cast is planned for them (register_decimal_compare_op, comparison.rs:61-99)
    if (n == 0) return DBHIP_OK;
    DBHIP_REQUIRE(lhs->precision >= 1 && lhs->precision <= 38 && rhs->precision >= 1 && rhs->precision <= 38 &&
                      lhs->scale <= lhs->precision && rhs->scale <= rhs->precision, "dbhip_cmp: bad DecimalSize");
    const int scale = lhs->scale > rhs->scale ? lhs->scale : rhs->scale;
    const int la = lhs->precision - lhs->scale, lb = rhs->precision - rhs->scale;
    int precision = (la > lb ? la : lb) + scale;   // calc_size (comparison.rs:369-384)
    if (precision > 38) precision = 38;
    CmpDecParams q;
    q.a = lhs->data; q.b = rhs->data; q.out = out_bitmap; q.n = n; q.out_bytes = ceil_div(n, 8);
    q.a_wide = lhs->type == DBHIP_T_DEC128; q.b_wide = rhs->type == DBHIP_T_DEC128;
    q.a_scalar = lhs->is_scalar; q.b_scalar = rhs->is_scalar; q.op = op; q.t128 = precision > 18;
    q.fa_pow = scale - lhs->scale; q.fb_pow = scale - rhs->scale;
    hipLaunchKernelGGL(cmp_decimal_kernel, dim3(grid_for(n, 256)), dim3(256), 0, resolve_stream(stream), q);
    DBHIP_LAUNCH_CHECK();
    return DBHIP_OK;
  }
  if (lhs->type != rhs->type) {
    set_error("dbhip_cmp: operand types differ (%d vs %d); the planner casts to a common type",
              lhs->type, rhs->type);
    return DBHIP_ERR_INVALID;
  }
  DBHIP_REQUIRE(type_class(lhs->type) >= 0 || lhs->type == DBHIP_T_DEC128 ||
                    lhs->type == DBHIP_T_STRING || lhs->type == DBHIP_T_BOOL,
                "dbhip_cmp: unsupported type");
  if (n == 0) return DBHIP_OK;
  CmpParams p;
  p.a = lhs->data; p.b = rhs->data;
  p.abuf = lhs->buffers; p.bbuf = rhs->buffers;
  p.out = out_bitmap; p.n = n; p.out_bytes = ceil_div(n, 8);
  p.type = lhs->type; p.a_scalar = lhs->is_scalar; p.b_scalar = rhs->is_scalar; p.op = op;
  hipStream_t s = resolve_stream(stream);
  // narrow numeric columns: 16 rows per lane (needs naturally aligned columns, which Buffer<T> guarantees;
  // the 16-row vector load wants 16-byte alignment of the column start)
  const bool aligned = (((uintptr_t)p.a | (uintptr_t)p.b) & 15) == 0 || (p.a_scalar && (((uintptr_t)p.b) & 15) == 0) ||
                       (p.b_scalar && (((uintptr_t)p.a) & 15) == 0);
  bool done = true;
  if (!aligned) done = false;
  else switch (p.type) {
    case DBHIP_T_I8: launch_cmp_narrow<int8_t>(p, s); break;
    case DBHIP_T_U8: launch_cmp_narrow<uint8_t>(p, s); break;
    case DBHIP_T_I16: launch_cmp_narrow<int16_t>(p, s); break;
    case DBHIP_T_U16: launch_cmp_narrow<uint16_t>(p, s); break;
    case DBHIP_T_I32: case DBHIP_T_DATE: launch_cmp_narrow<int32_t>(p, s); break;
    case DBHIP_T_U32: launch_cmp_narrow<uint32_t>(p, s); break;
    case DBHIP_T_F32: launch_cmp_narrow<float>(p, s); break;
    default: done = false; break;
  }
  if (!done) {
    int grid = grid_for(ceil_div(n, 4), 256);
    hipLaunchKernelGGL(cmp_kernel, dim3(grid), dim3(256), 0, s, p);
  }
  DBHIP_LAUNCH_CHECK();
  return DBHIP_OK;
}

int32_t dbhip_bitmap_binary(int32_t is_or, const uint8_t* a, const uint8_t* b, int64_t n,
                            uint8_t* out, void* stream) {
  if (n == 0) return DBHIP_OK;
  DBHIP_REQUIRE(a && b && out, "dbhip_bitmap_binary: NULL argument");
  DBHIP_REQUIRE(is_or >= 0 && is_or <= 2, "dbhip_bitmap_binary: op must be 0 (AND), 1 (OR) or 2 (AND NOT)");
  int64_t nbytes = ceil_div(n, 8);
  hipLaunchKernelGGL(bitmap_binary_kernel, dim3(grid_for(nbytes, 256)), dim3(256), 0,
                     resolve_stream(stream), a, b, out, nbytes, n, is_or);
  DBHIP_LAUNCH_CHECK();
  return DBHIP_OK;
}

int32_t dbhip_bitmap_set_indices(const uint32_t* idx, int64_t n_idx, uint8_t* bitmap, int64_t nbits, void* stream) {
  DBHIP_REQUIRE(n_idx >= 0 && nbits >= 0, "dbhip_bitmap_set_indices: negative count");
  if (n_idx == 0) return DBHIP_OK;
  DBHIP_REQUIRE(idx && bitmap, "dbhip_bitmap_set_indices: NULL argument");
  DBHIP_REQUIRE(((uintptr_t)bitmap & 3) == 0, "dbhip_bitmap_set_indices: the bitmap must be 4-byte aligned (it is written by words)");
  hipLaunchKernelGGL(bitmap_set_indices_kernel, dim3(grid_for(n_idx, 256)), dim3(256), 0, resolve_stream(stream), idx, n_idx, (uint32_t*)bitmap, nbits);
  DBHIP_LAUNCH_CHECK();
  return DBHIP_OK;
}

int32_t dbhip_bitmap_count(const uint8_t* bitmap, int64_t bit_offset, int64_t n,
                           uint64_t* out_count_dev, void* stream) {
  DBHIP_REQUIRE(out_count_dev, "dbhip_bitmap_count: NULL out");
  hipStream_t s = resolve_stream(stream);
  DBHIP_CHECK(hipMemsetAsync(out_count_dev, 0, 8, s));
  if (n == 0) return DBHIP_OK;
  DBHIP_REQUIRE(bitmap, "dbhip_bitmap_count: NULL bitmap");
  hipLaunchKernelGGL(bitmap_count_kernel, dim3(grid_for(ceil_div(n, 64), 256)), dim3(256), 0, s,
                     bitmap, bit_offset, n, (unsigned long long*)out_count_dev);
  DBHIP_LAUNCH_CHECK();
  return DBHIP_OK;
}

int32_t dbhip_filter_select(const uint8_t* bitmap, int64_t bit_offset, int64_t n,
                            uint32_t* out_sel, uint64_t* out_count_dev, void* stream) {
  DBHIP_REQUIRE(out_count_dev, "dbhip_filter_select: NULL out_count");
  DBHIP_REQUIRE(n <= 0xFFFFFFFFLL, "dbhip_filter_select: more than 2^32 rows in one block");
  hipStream_t s = resolve_stream(stream);
  if (n == 0) {
    DBHIP_CHECK(hipMemsetAsync(out_count_dev, 0, 8, s));
    return DBHIP_OK;
  }
  DBHIP_REQUIRE(bitmap && out_sel, "dbhip_filter_select: NULL argument");
  int64_t nwords = ceil_div(n, 64);
  int64_t nblocks = ceil_div(nwords, SEL_WORDS_PER_BLOCK);
  uint8_t* ws = (uint8_t*)scratch((size_t)nblocks * 12 + 64, 1, s);
  if (!ws) return DBHIP_ERR_HIP;
  uint64_t* offsets = (uint64_t*)ws;
  uint32_t* counts = (uint32_t*)(ws + nblocks * 8);
  hipLaunchKernelGGL(sel_count_kernel, dim3((unsigned)nblocks), dim3(256), 0, s, bitmap, bit_offset,
                     n, counts);
  hipLaunchKernelGGL(sel_scan_kernel, dim3(1), dim3(1024), 0, s, counts, offsets, nblocks,
                     (unsigned long long*)out_count_dev);
  hipLaunchKernelGGL(sel_write_kernel, dim3((unsigned)nblocks), dim3(256), 0, s, bitmap, bit_offset,
                     n, offsets, out_sel);
  DBHIP_LAUNCH_CHECK();
  return DBHIP_OK;
}

static int32_t select_run(SelArgs A, uint32_t* out_true, uint32_t* out_false, uint64_t* out_counts_dev, hipStream_t s) {
  const int64_t nwords = ceil_div(A.n, 64);
  const int64_t nblocks = ceil_div(nwords, SEL_WORDS_PER_BLOCK);
  uint8_t* ws = (uint8_t*)scratch((size_t)nblocks * SEL_WORDS_PER_BLOCK * 8 + (size_t)nblocks * 12 + 128, 1, s);
  if (!ws) return DBHIP_ERR_HIP;
  uint64_t* bits = (uint64_t*)ws;
  uint64_t* offsets = bits + nblocks * SEL_WORDS_PER_BLOCK;
  uint32_t* counts = (uint32_t*)(offsets + nblocks);
  DBHIP_CHECK(hipMemsetAsync(bits + nwords, 0, (size_t)(nblocks * SEL_WORDS_PER_BLOCK - nwords) * 8, s));
  A.bits = bits;
  hipLaunchKernelGGL(select_eval_kernel, dim3(grid_for(A.n, 256)), dim3(256), 0, s, A);
  hipLaunchKernelGGL(sel_count_kernel, dim3((unsigned)nblocks), dim3(256), 0, s, (const uint8_t*)bits, (int64_t)0, A.n, counts);
  hipLaunchKernelGGL(sel_scan_kernel, dim3(1), dim3(1024), 0, s, counts, offsets, nblocks, (unsigned long long*)out_counts_dev);
  hipLaunchKernelGGL(select_split_kernel, dim3((unsigned)nblocks), dim3(256), 0, s, bits, A.n, offsets, A.sel_in, out_true, out_false);
  DBHIP_LAUNCH_CHECK();
  return DBHIP_OK;
}

int32_t dbhip_select_cmp(int32_t op, const dbhip_col* lhs, const dbhip_col* rhs, const uint32_t* sel_in, int64_t n, uint32_t* out_true,
                         uint32_t* out_false, uint64_t* out_count_true_dev, void* stream) {
  DBHIP_REQUIRE(lhs && rhs && out_count_true_dev, "dbhip_select_cmp: NULL argument");
  DBHIP_REQUIRE(op >= DBHIP_CMP_EQ && op <= DBHIP_CMP_GTE, "dbhip_select_cmp: bad comparison");
  DBHIP_REQUIRE(n >= 0 && n <= 0xFFFFFFFFLL, "dbhip_select_cmp: more than 2^32 entries");
  hipStream_t s = resolve_stream(stream);
  if (n == 0) { DBHIP_CHECK(hipMemsetAsync(out_count_true_dev, 0, 8, s)); return DBHIP_OK; }
  DBHIP_REQUIRE(out_true && lhs->data && rhs->data, "dbhip_select_cmp: NULL buffer");
  if (lhs->type != rhs->type || !(type_class(lhs->type) >= 0 || lhs->type == DBHIP_T_DEC128 || lhs->type == DBHIP_T_BOOL)) {
    set_error("dbhip_select_cmp: operand types %d / %d (equal fixed-width types only; strings and decimals of different sizes go through dbhip_cmp)", lhs->type, rhs->type);
    return DBHIP_ERR_UNSUPPORTED;
  }
  if ((lhs->type == DBHIP_T_DEC64 || lhs->type == DBHIP_T_DEC128) && lhs->scale != rhs->scale) {
    set_error("dbhip_select_cmp: decimal operands of different scales go through dbhip_cmp");
    return DBHIP_ERR_UNSUPPORTED;
  }
  SelArgs A;
  memset(&A, 0, sizeof(A));
  A.a = lhs->data; A.b = rhs->data; A.av = lhs->validity; A.bv = rhs->validity; A.avo = lhs->validity_offset; A.bvo = rhs->validity_offset;
  A.type = lhs->type; A.a_scalar = lhs->is_scalar; A.b_scalar = rhs->is_scalar; A.op = op; A.sel_in = sel_in; A.n = n;
  return select_run(A, out_true, out_false, out_count_true_dev, s);
}

int32_t dbhip_select_bool(const dbhip_col* predicate, const uint32_t* sel_in, int64_t n, uint32_t* out_true, uint32_t* out_false,
                          uint64_t* out_count_true_dev, void* stream) {
  DBHIP_REQUIRE(predicate && out_count_true_dev && predicate->type == DBHIP_T_BOOL, "dbhip_select_bool: a Boolean column");
  DBHIP_REQUIRE(n >= 0 && n <= 0xFFFFFFFFLL, "dbhip_select_bool: more than 2^32 entries");
  hipStream_t s = resolve_stream(stream);
  if (n == 0) { DBHIP_CHECK(hipMemsetAsync(out_count_true_dev, 0, 8, s)); return DBHIP_OK; }
  DBHIP_REQUIRE(out_true && predicate->data, "dbhip_select_bool: NULL buffer");
  SelArgs A;
  memset(&A, 0, sizeof(A));
  A.a = predicate->data; A.av = predicate->validity; A.avo = predicate->validity_offset; A.type = DBHIP_T_BOOL; A.a_scalar = predicate->is_scalar;
  A.op = -1; A.sel_in = sel_in; A.n = n;
  return select_run(A, out_true, out_false, out_count_true_dev, s);
}

int32_t dbhip_take(const void* src, int32_t elem_size, const uint32_t* sel, int64_t n_sel,
                   void* out, void* stream) {
  if (n_sel == 0) return DBHIP_OK;
  DBHIP_REQUIRE(src && sel && out, "dbhip_take: NULL argument");
  hipStream_t s = resolve_stream(stream);
  int grid = grid_for(ceil_div(n_sel, 4), 256, 1024);
  if (n_sel >= (1 << 16) && (elem_size == 4 || elem_size == 8 || elem_size == 16)) {
    // large selections: the windowed kernel (a wave decides per 64 entries between the LDS window and a plain gather)
    const int64_t nchunks = ceil_div(n_sel, 1024);
    const int wg = (int)(nchunks < 2048 ? nchunks : 2048);
    switch (elem_size) {
      case 4: hipLaunchKernelGGL((take_window_kernel<uint32_t, 7680>), dim3(wg), dim3(256), 0, s, (const uint32_t*)src, sel, n_sel, (uint32_t*)out); break;
      case 8: hipLaunchKernelGGL((take_window_kernel<uint64_t, 3840>), dim3(wg), dim3(256), 0, s, (const uint64_t*)src, sel, n_sel, (uint64_t*)out); break;
      default: hipLaunchKernelGGL((take_window_kernel<B16, 1920>), dim3(wg), dim3(256), 0, s, (const B16*)src, sel, n_sel, (B16*)out); break;
    }
    DBHIP_LAUNCH_CHECK();
    return DBHIP_OK;
  }
  switch (elem_size) {
    case 1: hipLaunchKernelGGL(take_kernel<uint8_t>, dim3(grid), dim3(256), 0, s, (const uint8_t*)src, sel, n_sel, (uint8_t*)out); break;
    case 2: hipLaunchKernelGGL(take_kernel<uint16_t>, dim3(grid), dim3(256), 0, s, (const uint16_t*)src, sel, n_sel, (uint16_t*)out); break;
    case 4: hipLaunchKernelGGL(take_kernel<uint32_t>, dim3(grid), dim3(256), 0, s, (const uint32_t*)src, sel, n_sel, (uint32_t*)out); break;
    case 8: hipLaunchKernelGGL(take_kernel<uint64_t>, dim3(grid), dim3(256), 0, s, (const uint64_t*)src, sel, n_sel, (uint64_t*)out); break;
    case 16: hipLaunchKernelGGL(take_kernel<B16>, dim3(grid), dim3(256), 0, s, (const B16*)src, sel, n_sel, (B16*)out); break;
    default:
      set_error("dbhip_take: unsupported element size %d", elem_size);
      return DBHIP_ERR_INVALID;
  }
  DBHIP_LAUNCH_CHECK();
  return DBHIP_OK;
}

int32_t dbhip_take_block(const void* const* srcs_host, const int32_t* elem_sizes_host, int32_t ncols, const uint32_t* sel, int64_t n_sel,
                         void* const* outs_host, void* stream) {
  DBHIP_REQUIRE(ncols >= 0 && ncols <= TAKE_MAX_COLS && (ncols == 0 || (srcs_host && elem_sizes_host && outs_host)), "dbhip_take_block: 0..8 columns");
  if (n_sel == 0 || ncols == 0) return DBHIP_OK;
  DBHIP_REQUIRE(sel, "dbhip_take_block: NULL selection");
  TakeCols P;
  memset(&P, 0, sizeof(P));
  P.n = ncols;
  for (int k = 0; k < ncols; ++k) {
    const int e = elem_sizes_host[k];
    if (!(e == 1 || e == 2 || e == 4 || e == 8 || e == 16) || !srcs_host[k] || !outs_host[k]) {
      set_error("dbhip_take_block: column %d: element size %d / NULL buffer", k, e);
      return DBHIP_ERR_INVALID;
    }
    P.src[k] = srcs_host[k]; P.out[k] = outs_host[k]; P.elem[k] = e;
  }
  hipStream_t s = resolve_stream(stream);
  const int64_t nchunks = ceil_div(n_sel, 256);
  hipLaunchKernelGGL(take_block_kernel, dim3((unsigned)(nchunks < 2048 ? nchunks : 2048)), dim3(256), 0, s, P, sel, n_sel);
  DBHIP_LAUNCH_CHECK();
  return DBHIP_OK;
}

int32_t dbhip_take_bitmap(const uint8_t* src, int64_t bit_offset, const uint32_t* sel,
                          int64_t n_sel, uint8_t* out, void* stream) {
  if (n_sel == 0) return DBHIP_OK;
  DBHIP_REQUIRE(src && sel && out, "dbhip_take_bitmap: NULL argument");
  hipLaunchKernelGGL(take_bitmap_kernel, dim3(grid_for(n_sel, 256)), dim3(256), 0,
                     resolve_stream(stream), src, bit_offset, sel, n_sel, out, ceil_div(n_sel, 8));
  DBHIP_LAUNCH_CHECK();
  return DBHIP_OK;
}

int32_t dbhip_take_outer(const void* src, const uint8_t* src_validity, int64_t src_validity_offset, int32_t elem_size,
                         const uint32_t* idx, int64_t n, void* out, uint8_t* out_validity, void* stream) {
  if (n == 0) return DBHIP_OK;
  DBHIP_REQUIRE(src && idx && out && out_validity, "dbhip_take_outer: NULL argument");
  DBHIP_REQUIRE(((uintptr_t)out_validity & 7) == 0, "dbhip_take_outer: the validity buffer must be 8-byte aligned");
  hipStream_t s = resolve_stream(stream);
  const int grid = grid_for(n, 256, 1024);
#define TO(T) hipLaunchKernelGGL(take_outer_kernel<T>, dim3(grid), dim3(256), 0, s, (const T*)src, src_validity, src_validity_offset, idx, n, (T*)out, \
                                 (uint64_t*)out_validity, out_validity)
  switch (elem_size) {
    case 1: TO(uint8_t); break;
    case 2: TO(uint16_t); break;
    case 4: TO(uint32_t); break;
    case 8: TO(uint64_t); break;
    case 16: TO(B16); break;
    default:
      set_error("dbhip_take_outer: unsupported element size %d", elem_size);
      return DBHIP_ERR_INVALID;
  }
#undef TO
  DBHIP_LAUNCH_CHECK();
  return DBHIP_OK;
}

static int32_t sel_expand(const uint32_t* items_host, int32_t n_items, int repeat, uint32_t* out_sel, int64_t num_rows, void* stream,
                          const char* who) {
  if (num_rows == 0) return DBHIP_OK;
  DBHIP_REQUIRE(items_host && n_items >= 1 && out_sel, "selection from ranges / repeats: NULL argument");
  std::vector<uint32_t> host((size_t)2 * n_items);
  uint64_t total = 0;
  for (int r = 0; r < n_items; ++r) {
    const uint32_t x = items_host[2 * r], y = items_host[2 * r + 1];
    const uint64_t len = repeat ? y : (y >= x ? y - x : 0);
    if (!repeat && y < x) { set_error("%s: range %d ends before it starts", who, r); return DBHIP_ERR_INVALID; }
    host[r] = x;
    host[n_items + r] = (uint32_t)total;
    total += len;
  }
  if ((int64_t)total != num_rows) {  // debug_assert in the reference (take_compact.rs:43-47)
    set_error("%s: the items cover %llu rows, num_rows is %lld", who, (unsigned long long)total, (long long)num_rows);
    return DBHIP_ERR_INVALID;
  }
  hipStream_t s = resolve_stream(stream);
  uint32_t* dev = (uint32_t*)scratch((size_t)2 * n_items * 4, 7, s);
  if (!dev) return DBHIP_ERR_HIP;
  DBHIP_CHECK(hipMemcpyAsync(dev, host.data(), (size_t)2 * n_items * 4, hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(sel_expand_kernel, dim3(grid_for(num_rows, 256)), dim3(256), 0, s, dev, dev + n_items, n_items, repeat, num_rows, out_sel);
  DBHIP_LAUNCH_CHECK();
  DBHIP_CHECK(hipStreamSynchronize(s));  // `host` (pageable) was the source of an async copy
  return DBHIP_OK;
}

int32_t dbhip_sel_from_ranges(const uint32_t* ranges_host, int32_t n_ranges, uint32_t* out_sel, int64_t num_rows, void* stream) {
  return sel_expand(ranges_host, n_ranges, 0, out_sel, num_rows, stream, "dbhip_sel_from_ranges");
}

int32_t dbhip_sel_from_repeats(const uint32_t* repeats_host, int32_t n_repeats, uint32_t* out_sel, int64_t num_rows, void* stream) {
  return sel_expand(repeats_host, n_repeats, 1, out_sel, num_rows, stream, "dbhip_sel_from_repeats");
}

int32_t dbhip_take_chunks(const void* const* blocks_host, int32_t n_blocks, int32_t elem_size, const uint32_t* pairs, int64_t n,
                          void* out, void* stream) {
  if (n == 0) return DBHIP_OK;
  DBHIP_REQUIRE(blocks_host && n_blocks >= 1 && pairs && out, "dbhip_take_chunks: NULL argument");
  DBHIP_REQUIRE(elem_size == 1 || elem_size == 2 || elem_size == 4 || elem_size == 8 || elem_size == 16 || elem_size == 0,
                "dbhip_take_chunks: elem_size must be 0 (bitmap), 1, 2, 4, 8 or 16");
  hipStream_t s = resolve_stream(stream);
  const void** dev = (const void**)scratch((size_t)n_blocks * 8, 7, s);
  if (!dev) return DBHIP_ERR_HIP;
  DBHIP_CHECK(hipMemcpyAsync(dev, blocks_host, (size_t)n_blocks * 8, hipMemcpyHostToDevice, s));
  if (elem_size == 0)
    hipLaunchKernelGGL(take_chunks_bitmap_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, (const uint8_t* const*)dev, pairs, n, (uint8_t*)out, ceil_div(n, 8));
  else
    hipLaunchKernelGGL(take_chunks_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, (const void* const*)dev, elem_size, pairs, n, out);
  DBHIP_LAUNCH_CHECK();
  DBHIP_CHECK(hipStreamSynchronize(s));  // the pointer table came from caller-owned pageable memory
  return DBHIP_OK;
}

}  // extern "C"

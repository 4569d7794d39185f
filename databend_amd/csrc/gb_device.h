// gb_device.h — device functions shared by the hash-aggregation kernels
// (k_groupby.hip) and the fused Q1 pipeline (k_q1.hip).
#pragma once
#include "dev_common.h"
#include "gb_layout.h"

// order-preserving u64 key for MIN/MAX states: signed ints flip the sign bit,
// floats use the OrderedFloat total order (NaN largest, types/number.rs:47-48).
__device__ __forceinline__ uint64_t ord_encode(uint64_t raw, int type) {
  switch (type) {
    case DBHIP_T_U8: case DBHIP_T_U16: case DBHIP_T_U32: case DBHIP_T_U64: case DBHIP_T_BOOL:
      return raw;
    case DBHIP_T_F32: {
      float f = __uint_as_float((uint32_t)raw);
      if (f != f) return ~0ULL;
      double d = (double)f;
      uint64_t b = (uint64_t)__double_as_longlong(d);
      return (b >> 63) ? ~b : (b | 0x8000000000000000ULL);
    }
    case DBHIP_T_F64: {
      double d = __longlong_as_double((long long)raw);
      if (d != d) return ~0ULL;
      return (raw >> 63) ? ~raw : (raw | 0x8000000000000000ULL);
    }
    default:
      return raw ^ 0x8000000000000000ULL;
  }
}

__device__ __forceinline__ uint64_t ord_decode(uint64_t enc, int type) {
  switch (type) {
    case DBHIP_T_U8: case DBHIP_T_U16: case DBHIP_T_U32: case DBHIP_T_U64: case DBHIP_T_BOOL:
      return enc;
    case DBHIP_T_F32: {
      if (enc == ~0ULL) return 0x7fc00000u;
      uint64_t b = (enc >> 63) ? (enc & 0x7fffffffffffffffULL) : ~enc;
      return (uint64_t)__float_as_uint((float)__longlong_as_double((long long)b));
    }
    case DBHIP_T_F64: {
      if (enc == ~0ULL) return 0x7ff8000000000000ULL;
      return (enc >> 63) ? (enc & 0x7fffffffffffffffULL) : ~enc;
    }
    default:
      return enc ^ 0x8000000000000000ULL;
  }
}

// Loads the value of `col` at `row` as canonical words (see gb_layout.h).
// Returns false for strings longer than 12 bytes (unsupported as keys).
__device__ __forceinline__ bool gb_load_words(const GbCol& c, int64_t row, uint64_t w[2], bool* valid) {
  int64_t j = c.is_scalar ? 0 : row;
  *valid = !c.validity || bit_get(c.validity, c.voff + j);
  w[0] = 0;
  w[1] = 0;
  switch (c.type) {
    case DBHIP_T_BOOL: w[0] = bit_get((const uint8_t*)c.data, j); break;
    case DBHIP_T_I8: w[0] = (uint64_t)(int64_t)((const int8_t*)c.data)[j]; break;
    case DBHIP_T_I16: w[0] = (uint64_t)(int64_t)((const int16_t*)c.data)[j]; break;
    case DBHIP_T_I32: case DBHIP_T_DATE: w[0] = (uint64_t)(int64_t)((const int32_t*)c.data)[j]; break;
    case DBHIP_T_I64: case DBHIP_T_TIMESTAMP: case DBHIP_T_DEC64:
      w[0] = (uint64_t)((const int64_t*)c.data)[j]; break;
    case DBHIP_T_U8: w[0] = ((const uint8_t*)c.data)[j]; break;
    case DBHIP_T_U16: w[0] = ((const uint16_t*)c.data)[j]; break;
    case DBHIP_T_U32: case DBHIP_T_F32: w[0] = ((const uint32_t*)c.data)[j]; break;
    case DBHIP_T_U64: case DBHIP_T_F64: w[0] = ((const uint64_t*)c.data)[j]; break;
    case DBHIP_T_DEC128: {
      const uint64_t* p = (const uint64_t*)c.data + 2 * j;
      w[0] = p[0];
      w[1] = p[1];
    } break;
    case DBHIP_T_STRING: {
      const uint32_t* p = (const uint32_t*)c.data + 4 * j;
      uint32_t len = p[0];
      if (len > 12) return false;
      uint32_t d1 = p[1], d2 = p[2], d3 = p[3];
      // zero the bytes past len so equal strings are equal words
      if (len < 4) { d1 &= (len == 0) ? 0u : (0xffffffffu >> (8 * (4 - len))); d2 = 0; d3 = 0; }
      else if (len < 8) { d2 &= (len == 4) ? 0u : (0xffffffffu >> (8 * (8 - len))); d3 = 0; }
      else if (len < 12) { d3 &= (len == 8) ? 0u : (0xffffffffu >> (8 * (12 - len))); }
      w[0] = ((uint64_t)d1 << 32) | len;
      w[1] = ((uint64_t)d3 << 32) | d2;
    } break;
    default: return false;
  }
  if (!*valid) { w[0] = 0; w[1] = 0; }
  return true;
}

// The same for N rows of ONE column with the type switch OUTSIDE the row loop: inside a case the N loads are
// independent instructions in one basic block, so they are all in flight together. (With the switch inside a per-row
// loop every load sits in its own block behind a scalar branch and costs a full memory round trip: 8 rows x 2
// columns = 16 serialised HBM latencies per tile made the aggregation kernels latency bound.)
template <int N>
__device__ __forceinline__ bool gb_load_words_n(const GbCol& c, const int64_t (&row)[N], uint64_t (&w0)[N], uint64_t (&w1)[N],
                                                bool (&valid)[N]) {
  int64_t j[N];
  bool ok = true;
#pragma unroll
  for (int u = 0; u < N; ++u) { j[u] = c.is_scalar ? 0 : row[u]; w0[u] = 0; w1[u] = 0; valid[u] = true; }
  if (c.validity) {
#pragma unroll
    for (int u = 0; u < N; ++u) valid[u] = bit_get(c.validity, c.voff + j[u]);
  }
  switch (c.type) {
    case DBHIP_T_BOOL:
#pragma unroll
      for (int u = 0; u < N; ++u) w0[u] = bit_get((const uint8_t*)c.data, j[u]);
      break;
    case DBHIP_T_I8:
#pragma unroll
      for (int u = 0; u < N; ++u) w0[u] = (uint64_t)(int64_t)((const int8_t*)c.data)[j[u]];
      break;
    case DBHIP_T_I16:
#pragma unroll
      for (int u = 0; u < N; ++u) w0[u] = (uint64_t)(int64_t)((const int16_t*)c.data)[j[u]];
      break;
    case DBHIP_T_I32: case DBHIP_T_DATE:
#pragma unroll
      for (int u = 0; u < N; ++u) w0[u] = (uint64_t)(int64_t)((const int32_t*)c.data)[j[u]];
      break;
    case DBHIP_T_I64: case DBHIP_T_TIMESTAMP: case DBHIP_T_DEC64: case DBHIP_T_U64: case DBHIP_T_F64:
#pragma unroll
      for (int u = 0; u < N; ++u) w0[u] = ((const uint64_t*)c.data)[j[u]];
      break;
    case DBHIP_T_U8:
#pragma unroll
      for (int u = 0; u < N; ++u) w0[u] = ((const uint8_t*)c.data)[j[u]];
      break;
    case DBHIP_T_U16:
#pragma unroll
      for (int u = 0; u < N; ++u) w0[u] = ((const uint16_t*)c.data)[j[u]];
      break;
    case DBHIP_T_U32: case DBHIP_T_F32:
#pragma unroll
      for (int u = 0; u < N; ++u) w0[u] = ((const uint32_t*)c.data)[j[u]];
      break;
    case DBHIP_T_DEC128:
#pragma unroll
      for (int u = 0; u < N; ++u) {
        const uint64_t* p = (const uint64_t*)c.data + 2 * j[u];
        w0[u] = p[0];
        w1[u] = p[1];
      }
      break;
    case DBHIP_T_STRING: {
      uint32_t d0[N], d1[N], d2[N], d3[N];
#pragma unroll
      for (int u = 0; u < N; ++u) {
        const uint32_t* p = (const uint32_t*)c.data + 4 * j[u];
        d0[u] = p[0]; d1[u] = p[1]; d2[u] = p[2]; d3[u] = p[3];
      }
#pragma unroll
      for (int u = 0; u < N; ++u) {
        const uint32_t len = d0[u];
        if (len > 12) { ok = false; continue; }
        uint32_t a = d1[u], b = d2[u], cc = d3[u];
        // zero the bytes past len so equal strings are equal words
        if (len < 4) { a &= (len == 0) ? 0u : (0xffffffffu >> (8 * (4 - len))); b = 0; cc = 0; }
        else if (len < 8) { b &= (len == 4) ? 0u : (0xffffffffu >> (8 * (8 - len))); cc = 0; }
        else if (len < 12) { cc &= (len == 8) ? 0u : (0xffffffffu >> (8 * (12 - len))); }
        w0[u] = ((uint64_t)a << 32) | len;
        w1[u] = ((uint64_t)cc << 32) | b;
      }
    } break;
    default:
      ok = false;
      break;
  }
#pragma unroll
  for (int u = 0; u < N; ++u)
    if (!valid[u]) { w0[u] = 0; w1[u] = 0; }
  return ok;
}

// AggHash of one key value given its canonical words (group_hash.rs:513-632).
__device__ __forceinline__ uint64_t gb_hash_words(int type, const uint64_t w[2], bool valid) {
  if (!valid) return DBHIP_NULL_HASH_VAL;
  switch (type) {
    case DBHIP_T_BOOL: return w[0];                                   // :581-585
    case DBHIP_T_F32: {                                               // :599-609
      float f = __uint_as_float((uint32_t)w[0]);
      uint32_t bits = (f != f) ? 0x7fc00000u : (uint32_t)w[0];
      return agg_hash_u64((uint64_t)bits);
    }
    case DBHIP_T_F64: {                                               // :611-620
      double d = __longlong_as_double((long long)w[0]);
      uint64_t bits = (d != d) ? 0x7ff8000000000000ULL : w[0];
      return agg_hash_u64(bits);
    }
    case DBHIP_T_DEC128:                                              // :587-591
      return agg_hash_i128((i128)(((u128)w[1] << 64) | w[0]));
    case DBHIP_T_STRING: {                                            // :522-553
      uint32_t len = (uint32_t)w[0];
      return agg_hash_inline_view(len, (uint32_t)(w[0] >> 32), (uint32_t)w[1], (uint32_t)(w[1] >> 32));
    }
    default:
      return agg_hash_u64(w[0]);                                      // :555-570
  }
}

// 128-bit wrapping add into two adjacent u64 words with global atomics.
__device__ __forceinline__ void atomic_add_u128(uint64_t* p, uint64_t lo, uint64_t hi) {
  unsigned long long old = atomicAdd((unsigned long long*)p, (unsigned long long)lo);
  uint64_t carry = ((uint64_t)old + lo) < lo ? 1 : 0;
  uint64_t addhi = hi + carry;
  if (addhi) atomicAdd((unsigned long long*)(p + 1), (unsigned long long)addhi);
}

// 192-bit add (lo, hi, ext) with global atomics; every carry is observed by exactly one adder.
__device__ __forceinline__ void atomic_add_u192(uint64_t* p, uint64_t lo, uint64_t hi, uint64_t ext) {
  unsigned long long old = atomicAdd((unsigned long long*)p, (unsigned long long)lo);
  uint64_t c1 = ((uint64_t)old + lo) < lo ? 1 : 0;
  uint64_t addhi = hi + c1;
  uint64_t c2 = addhi < hi ? 1 : 0;
  if (addhi) {
    unsigned long long oh = atomicAdd((unsigned long long*)(p + 1), (unsigned long long)addhi);
    c2 |= ((uint64_t)oh + addhi) < addhi ? 1 : 0;
  }
  uint64_t addext = ext + c2;
  if (addext) atomicAdd((unsigned long long*)(p + 2), (unsigned long long)addext);
}

// SUM over Decimal256 (r04; aggregate_sum.rs:183-300 with T = i256): FIVE words = the exact 320-bit two's complement total (the value's
// four words + its sign extension), so that "left [DECIMAL_MIN, DECIMAL_MAX]" is decided on the exact total like the Decimal128 sum.
// Word-by-word atomic adds; every carry is observed by exactly one adder.
#define GB_SUM256_WORDS 5
__device__ __forceinline__ void atomic_add_words(uint64_t* p, const uint64_t* v, int n) {
  uint64_t carry = 0;
  for (int i = 0; i < n; ++i) {
    const uint64_t add = v[i] + carry;
    uint64_t c = add < carry ? 1 : 0;   // v[i] = ~0 and a carry in: nothing to add here, the carry moves on
    if (add) {
      const unsigned long long old = atomicAdd((unsigned long long*)(p + i), (unsigned long long)add);
      c |= ((uint64_t)old + add) < add ? 1 : 0;
    }
    carry = c;
  }
}
__device__ __forceinline__ void wg_add_words(uint64_t* p, const uint64_t* v, int n) {
  uint64_t carry = 0;
  for (int i = 0; i < n; ++i) {
    const uint64_t add = v[i] + carry;
    uint64_t c = add < carry ? 1 : 0;
    if (add) {
      const uint64_t old = __hip_atomic_fetch_add((unsigned long long*)(p + i), (unsigned long long)add, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      c |= (old + add) < add ? 1 : 0;
    }
    carry = c;
  }
}
__device__ __forceinline__ void plain_add_words(uint64_t* p, const uint64_t* v, int n) {
  uint64_t carry = 0;
  for (int i = 0; i < n; ++i) {
    const uint64_t a = p[i] + v[i];
    const uint64_t c1 = a < v[i] ? 1 : 0;
    const uint64_t b = a + carry;
    const uint64_t c2 = b < a ? 1 : 0;
    p[i] = b;
    carry = c1 | c2;
  }
}
__host__ __device__ __forceinline__ bool gb_sum256(const GbLayout& L, int a) {
  return L.agg_kind[a] == DBHIP_AGG_SUM && L.agg_type[a] == DBHIP_T_DEC256;
}

// MIN / MAX over Decimal128 (r03): the state is THREE words — [0] the high 64 bits with the sign flipped (so that unsigned order
// is value order), [1] the has-value word like every other min / max state, [2] the low 64 bits — compared lexicographically as
// ([0], [2]). No 128-bit atomic exists, so a concurrent merge takes a PER-STATE SPIN LOCK in bit 63 of the has word: try-lock inside
// the loop body (a lane that gets the lock does its compare-and-store and releases before the wave iterates again, so lanes of one
// wave that want the same state cannot dead-lock each other); values go through agent-scope loads / stores, the release store of
// the has word publishes them. After a kernel the word is 0 or 1 again. Layouts with such a state stay on the row path.
#define GB_MM_LOCK (1ULL << 63)
__host__ __device__ __forceinline__ bool gb_minmax_str(const GbLayout& L, int a) {
  return (L.agg_kind[a] == DBHIP_AGG_MIN || L.agg_kind[a] == DBHIP_AGG_MAX) && L.agg_type[a] == DBHIP_T_STRING;
}
__host__ __device__ __forceinline__ bool gb_minmax256(const GbLayout& L, int a) {
  return (L.agg_kind[a] == DBHIP_AGG_MIN || L.agg_kind[a] == DBHIP_AGG_MAX) && L.agg_type[a] == DBHIP_T_DEC256;
}
__host__ __device__ __forceinline__ bool gb_minmax_wide(const GbLayout& L, int a) {
  return (L.agg_kind[a] == DBHIP_AGG_MIN || L.agg_kind[a] == DBHIP_AGG_MAX) &&
         (L.agg_type[a] == DBHIP_T_DEC128 || L.agg_type[a] == DBHIP_T_STRING || L.agg_type[a] == DBHIP_T_DEC256);
}
__device__ __forceinline__ bool gb_mm_better(bool is_min, uint64_t hi, uint64_t lo, uint64_t chi, uint64_t clo) {
  return is_min ? (hi < chi || (hi == chi && lo < clo)) : (hi > chi || (hi == chi && lo > clo));
}
__device__ __forceinline__ void gb_minmax_wide_locked(bool is_min, uint64_t* dst, const uint64_t* v) {
  if (!v[1]) return;
  unsigned long long* has = (unsigned long long*)(dst + 1);
  bool done = false;
  while (!done) {
    const unsigned long long old = atomicOr(has, (unsigned long long)GB_MM_LOCK);
    if (!(old & GB_MM_LOCK)) {
      const uint64_t chi = __hip_atomic_load(dst, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const uint64_t clo = __hip_atomic_load(dst + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (!(old & 1ULL) || gb_mm_better(is_min, v[0], v[2], chi, clo)) {
        __hip_atomic_store(dst, v[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(dst + 2, v[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      __hip_atomic_store(has, 1ULL, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      done = true;
    } else {
      __builtin_amdgcn_s_sleep(8);   // back off: thousands of waves hammering one word starve the holder's own accesses
    }
  }
}
// MIN / MAX over String (r04; aggregate_min_max_any.rs:62-110, StringState): THREE words like the Decimal128 state and merged under
// the same per-state lock — [0] the view's first 8 bytes (len | first four bytes << 32), [1] the has-value / lock word, [2] bytes
// 4..11 of a string of at most 12 bytes, or the device ADDRESS of the bytes of a longer one. While a block is being added the address
// may point into the block's own data buffers (or another table's arena); before the call returns the winners are copied into this
// table's arena (gb_pin_strings_*, k_groupby.hip), so a state never outlives the bytes it refers to. Order: bytes, then length
// (Rust's `str` / `[u8]` Ord).
__device__ __forceinline__ uint32_t gb_str_byte(uint64_t w0, uint64_t w2, uint32_t len, uint32_t i) {
  if (len <= 12) return (uint32_t)((i < 4 ? (w0 >> (32 + 8 * i)) : (w2 >> (8 * (i - 4)))) & 0xFFu);
  return ((const uint8_t*)w2)[i];
}
// -1 / 0 / 1
__device__ __forceinline__ int gb_str_cmp(uint64_t a0, uint64_t a2, uint64_t b0, uint64_t b2) {
  const uint32_t la = (uint32_t)a0, lb = (uint32_t)b0;
  // the first four bytes, zero padded (a pad byte against a real byte orders the shorter string first, which is right)
  const uint32_t pa = __builtin_bswap32((uint32_t)(a0 >> 32)), pb = __builtin_bswap32((uint32_t)(b0 >> 32));
  if (pa != pb) return pa < pb ? -1 : 1;
  const uint32_t m = la < lb ? la : lb;
  for (uint32_t i = 4; i < m; ++i) {
    const uint32_t x = gb_str_byte(a0, a2, la, i), y = gb_str_byte(b0, b2, lb, i);
    if (x != y) return x < y ? -1 : 1;
  }
  return la == lb ? 0 : (la < lb ? -1 : 1);
}
__device__ __forceinline__ void gb_minmax_str_locked(bool is_min, uint64_t* dst, const uint64_t* v) {
  if (!v[1]) return;
  unsigned long long* has = (unsigned long long*)(dst + 1);
  bool done = false;
  while (!done) {
    const unsigned long long old = atomicOr(has, (unsigned long long)GB_MM_LOCK);
    if (!(old & GB_MM_LOCK)) {
      const uint64_t c0 = __hip_atomic_load(dst, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const uint64_t c2 = __hip_atomic_load(dst + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      bool better = !(old & 1ULL);
      if (!better) { const int c = gb_str_cmp(v[0], v[2], c0, c2); better = is_min ? c < 0 : c > 0; }
      if (better) {
        __hip_atomic_store(dst, v[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(dst + 2, v[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      __hip_atomic_store(has, 1ULL, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      done = true;
    } else {
      __builtin_amdgcn_s_sleep(8);
    }
  }
}
__device__ __forceinline__ void gb_minmax_str_plain(bool is_min, uint64_t* dst, const uint64_t* v) {   // ONE writer
  if (!v[1]) return;
  bool better = !dst[1];
  if (!better) { const int c = gb_str_cmp(v[0], v[2], dst[0], dst[2]); better = is_min ? c < 0 : c > 0; }
  if (better) { dst[0] = v[0]; dst[2] = v[2]; }
  dst[1] = 1;
}

// MIN / MAX over Decimal256 (r05; aggregate_min_max_any_decimal.rs:45-138 with T = i256): FIVE words — [0] the top 64 bits with the sign
// flipped, [1] the has-value / lock word, [2] [3] [4] the lower words from high to low — compared lexicographically as ([0], [2], [3], [4]),
// merged under the same per-state lock as the three-word states.
#define GB_MM256_WORDS 5
__device__ __forceinline__ bool gb_mm256_better(bool is_min, const uint64_t* v, uint64_t c0, uint64_t c2, uint64_t c3, uint64_t c4) {
  const uint64_t a[4] = {v[0], v[2], v[3], v[4]}, b[4] = {c0, c2, c3, c4};
  for (int q = 0; q < 4; ++q)
    if (a[q] != b[q]) return is_min ? a[q] < b[q] : a[q] > b[q];
  return false;
}
__device__ __forceinline__ void gb_minmax256_locked(bool is_min, uint64_t* dst, const uint64_t* v) {
  if (!v[1]) return;
  unsigned long long* has = (unsigned long long*)(dst + 1);
  bool done = false;
  while (!done) {
    const unsigned long long old = atomicOr(has, (unsigned long long)GB_MM_LOCK);
    if (!(old & GB_MM_LOCK)) {
      const uint64_t c0 = __hip_atomic_load(dst, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const uint64_t c2 = __hip_atomic_load(dst + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const uint64_t c3 = __hip_atomic_load(dst + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const uint64_t c4 = __hip_atomic_load(dst + 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (!(old & 1ULL) || gb_mm256_better(is_min, v, c0, c2, c3, c4)) {
        __hip_atomic_store(dst, v[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(dst + 2, v[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(dst + 3, v[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(dst + 4, v[4], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      __hip_atomic_store(has, 1ULL, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      done = true;
    } else {
      __builtin_amdgcn_s_sleep(8);
    }
  }
}
__device__ __forceinline__ void gb_minmax256_plain(bool is_min, uint64_t* dst, const uint64_t* v) {   // ONE writer
  if (!v[1]) return;
  if (!dst[1] || gb_mm256_better(is_min, v, dst[0], dst[2], dst[3], dst[4])) { dst[0] = v[0]; dst[2] = v[2]; dst[3] = v[3]; dst[4] = v[4]; }
  dst[1] = 1;
}

__device__ __forceinline__ void gb_minmax_wide_plain(bool is_min, uint64_t* dst, const uint64_t* v) {   // ONE writer
  if (!v[1]) return;
  if (!dst[1] || gb_mm_better(is_min, v[0], v[2], dst[0], dst[2])) { dst[0] = v[0]; dst[2] = v[2]; }
  dst[1] = 1;
}

// merge one state contribution `v` (agg_words words) into the state at `dst`
__device__ __forceinline__ void gb_atomic_merge(const GbLayout& L, int a, uint64_t* dst,
                                                const uint64_t* v) {
  switch (L.agg_kind[a]) {
    case DBHIP_AGG_COUNT:
      if (v[0]) atomicAdd((unsigned long long*)dst, (unsigned long long)v[0]);
      break;
    case DBHIP_AGG_SUM: {
      const int fw = L.agg_flag[a];
      if (L.agg_type[a] == DBHIP_T_DEC256) {
        atomic_add_words(dst, v, GB_SUM256_WORDS);
      } else if (L.agg_words[a] - (fw ? 1 : 0) == 3) {
        if (v[0] | v[1] | v[2]) atomic_add_u192(dst, v[0], v[1], v[2]);
      } else if (L.agg_type[a] == DBHIP_T_F32 || L.agg_type[a] == DBHIP_T_F64) {
        atomicAdd((double*)dst, __longlong_as_double((long long)v[0]));
      } else if (v[0]) {
        atomicAdd((unsigned long long*)dst, (unsigned long long)v[0]);
      }
      if (fw && v[fw]) atomicOr((unsigned long long*)(dst + fw), 1ULL);
    } break;
    case DBHIP_AGG_MIN:
      if (L.agg_type[a] == DBHIP_T_STRING) { gb_minmax_str_locked(true, dst, v); break; }
      if (L.agg_type[a] == DBHIP_T_DEC256) { gb_minmax256_locked(true, dst, v); break; }
      if (L.agg_words[a] == 3) { gb_minmax_wide_locked(true, dst, v); break; }
      if (v[1]) {
        atomicMin((unsigned long long*)dst, (unsigned long long)v[0]);
        atomicOr((unsigned long long*)(dst + 1), 1ULL);
      }
      break;
    default:  // MAX
      if (L.agg_type[a] == DBHIP_T_STRING) { gb_minmax_str_locked(false, dst, v); break; }
      if (L.agg_type[a] == DBHIP_T_DEC256) { gb_minmax256_locked(false, dst, v); break; }
      if (L.agg_words[a] == 3) { gb_minmax_wide_locked(false, dst, v); break; }
      if (v[1]) {
        atomicMax((unsigned long long*)dst, (unsigned long long)v[0]);
        atomicOr((unsigned long long*)(dst + 1), 1ULL);
      }
      break;
  }
}

// gb_atomic_merge with WORKGROUP-scope atomics: for a table slice that only one workgroup touches during the launch (the
// partition-exclusive insert, k_groupby.hip) — the atomic is resolved in the XCD's L2 instead of travelling the fabric
#define GB_WG_ADD(p, v) __hip_atomic_fetch_add((unsigned long long*)(p), (unsigned long long)(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
#define GB_WG_OR(p, v) __hip_atomic_fetch_or((unsigned long long*)(p), (unsigned long long)(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
__device__ __forceinline__ void gb_wg_merge(const GbLayout& L, int a, uint64_t* dst, const uint64_t* v) {
  switch (L.agg_kind[a]) {
    case DBHIP_AGG_COUNT:
      if (v[0]) GB_WG_ADD(dst, v[0]);
      break;
    case DBHIP_AGG_SUM: {
      const int fw = L.agg_flag[a];
      if (L.agg_type[a] == DBHIP_T_DEC256) {
        wg_add_words(dst, v, GB_SUM256_WORDS);
      } else if (L.agg_words[a] - (fw ? 1 : 0) == 3) {
        if (v[0] | v[1] | v[2]) {
          const uint64_t old = GB_WG_ADD(dst, v[0]);
          const uint64_t c1 = (old + v[0]) < v[0] ? 1 : 0;
          const uint64_t addhi = v[1] + c1;
          uint64_t c2 = addhi < v[1] ? 1 : 0;
          if (addhi) {
            const uint64_t oh = GB_WG_ADD(dst + 1, addhi);
            c2 |= (oh + addhi) < addhi ? 1 : 0;
          }
          const uint64_t addext = v[2] + c2;
          if (addext) GB_WG_ADD(dst + 2, addext);
        }
      } else if (L.agg_type[a] == DBHIP_T_F32 || L.agg_type[a] == DBHIP_T_F64) {
        __hip_atomic_fetch_add((double*)dst, __longlong_as_double((long long)v[0]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      } else if (v[0]) {
        GB_WG_ADD(dst, v[0]);
      }
      if (fw && v[fw]) GB_WG_OR(dst + fw, 1ULL);
    } break;
    case DBHIP_AGG_MIN:
      if (L.agg_type[a] == DBHIP_T_STRING) { gb_minmax_str_locked(true, dst, v); break; }
      if (L.agg_type[a] == DBHIP_T_DEC256) { gb_minmax256_locked(true, dst, v); break; }
      if (L.agg_words[a] == 3) { gb_minmax_wide_locked(true, dst, v); break; }
      if (v[1]) {
        __hip_atomic_fetch_min((unsigned long long*)dst, (unsigned long long)v[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        GB_WG_OR(dst + 1, 1ULL);
      }
      break;
    default:  // MAX
      if (L.agg_type[a] == DBHIP_T_STRING) { gb_minmax_str_locked(false, dst, v); break; }
      if (L.agg_type[a] == DBHIP_T_DEC256) { gb_minmax256_locked(false, dst, v); break; }
      if (L.agg_words[a] == 3) { gb_minmax_wide_locked(false, dst, v); break; }
      if (v[1]) {
        __hip_atomic_fetch_max((unsigned long long*)dst, (unsigned long long)v[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        GB_WG_OR(dst + 1, 1ULL);
      }
      break;
  }
}

// the same merge without atomics: `dst` has ONE writer (the partition-exclusive merge, k_groupby.hip)
__device__ __forceinline__ void gb_plain_merge(const GbLayout& L, int a, uint64_t* dst, const uint64_t* v) {
  switch (L.agg_kind[a]) {
    case DBHIP_AGG_COUNT:
      dst[0] += v[0];
      break;
    case DBHIP_AGG_SUM: {
      const int fw = L.agg_flag[a];
      if (L.agg_type[a] == DBHIP_T_DEC256) {
        plain_add_words(dst, v, GB_SUM256_WORDS);
      } else if (L.agg_words[a] - (fw ? 1 : 0) == 3) {
        const uint64_t lo = dst[0] + v[0];
        const uint64_t c1 = lo < v[0] ? 1 : 0;
        const uint64_t h1 = dst[1] + v[1];
        const uint64_t c2a = h1 < v[1] ? 1 : 0;
        const uint64_t hi = h1 + c1;
        const uint64_t c2 = c2a | (hi < h1 ? 1 : 0);
        dst[0] = lo; dst[1] = hi; dst[2] += v[2] + c2;
      } else if (L.agg_type[a] == DBHIP_T_F32 || L.agg_type[a] == DBHIP_T_F64) {
        dst[0] = (uint64_t)__double_as_longlong(__longlong_as_double((long long)dst[0]) + __longlong_as_double((long long)v[0]));
      } else {
        dst[0] += v[0];
      }
      if (fw && v[fw]) dst[fw] |= 1ULL;
    } break;
    case DBHIP_AGG_MIN:
      if (L.agg_type[a] == DBHIP_T_STRING) { gb_minmax_str_plain(true, dst, v); break; }
      if (L.agg_type[a] == DBHIP_T_DEC256) { gb_minmax256_plain(true, dst, v); break; }
      if (L.agg_words[a] == 3) { gb_minmax_wide_plain(true, dst, v); break; }
      if (v[1]) { dst[0] = v[0] < dst[0] ? v[0] : dst[0]; dst[1] |= 1ULL; }
      break;
    default:  // MAX
      if (L.agg_type[a] == DBHIP_T_STRING) { gb_minmax_str_plain(false, dst, v); break; }
      if (L.agg_type[a] == DBHIP_T_DEC256) { gb_minmax256_plain(false, dst, v); break; }
      if (L.agg_words[a] == 3) { gb_minmax_wide_plain(false, dst, v); break; }
      if (v[1]) { dst[0] = v[0] > dst[0] ? v[0] : dst[0]; dst[1] |= 1ULL; }
      break;
  }
}

// identity element of a state
__device__ __forceinline__ void gb_state_identity(const GbLayout& L, int a, uint64_t* dst) {
  for (int k = 0; k < L.agg_words[a]; ++k) dst[k] = 0;
  if (L.agg_kind[a] == DBHIP_AGG_MIN && L.agg_type[a] != DBHIP_T_STRING) { dst[0] = ~0ULL; if (L.agg_words[a] == 3) dst[2] = ~0ULL; }   // (a String state without a value is all zero)
}

// state contribution of ONE input row for aggregate a from the argument's canonical words (w0, w1) and validity
// (count(*) passes valid = true); v holds GB_MAX_STATE_WORDS words
__device__ __forceinline__ void gb_row_contrib(const GbLayout& L, int a, uint64_t w0, uint64_t w1, bool valid, uint64_t* v) {
  v[0] = 0; v[1] = 0; v[2] = 0; v[3] = 0;
  switch (L.agg_kind[a]) {
    case DBHIP_AGG_COUNT:
      v[0] = valid ? 1 : 0;
      break;
    case DBHIP_AGG_SUM: {
      const int fw = L.agg_flag[a];
      if (L.agg_type[a] == DBHIP_T_F32) w0 = (uint64_t)__double_as_longlong((double)__uint_as_float((uint32_t)w0));
      v[0] = valid ? w0 : 0;
      if (L.agg_words[a] - (fw ? 1 : 0) == 3) {
        v[1] = valid ? w1 : 0;
        v[2] = (valid && (w1 >> 63)) ? ~0ULL : 0;  // sign extension to 192 bits
      }
      if (fw) v[fw] = valid ? 1 : 0;
    } break;
    default:  // MIN / MAX
      if (L.agg_type[a] == DBHIP_T_STRING) { v[0] = valid ? w0 : 0; v[2] = valid ? w1 : 0; }   // String: (len | prefix, tail or address)
      else if (L.agg_words[a] == 3) { v[0] = w1 ^ (1ULL << 63); v[2] = w0; }   // Decimal128: (sign-flipped high word, low word)
      else v[0] = ord_encode(w0, L.agg_type[a]);
      v[1] = valid ? 1 : 0;
      break;
  }
}

// dev_common.h — device-side helpers shared by the gfx950 kernels.
// wave = 64 lanes on CDNA4; all wave-width constants are hard-coded to 64.
#pragma once
#ifndef __HIPCC_RTC__   // (hiprtc: the run-time compiled kernels get these from a prelude, k_fagg.hip)
#include <hip/hip_runtime.h>
#include <stdint.h>
#endif

#ifndef __HIPCC_RTC__
#include "../../include/dbhip.h"
#else
#include "dbhip.h"
#endif

typedef __int128 i128;
typedef unsigned __int128 u128;

#define DBHIP_WAVE 64

// ---------------------------------------------------------------------------
// group hash (reference: src/query/expression/src/aggregate/group_hash.rs)
// ---------------------------------------------------------------------------
#define DBHIP_NULL_HASH_VAL 0xd1cefa08eb382d69ULL  // group_hash.rs:38

// impl_agg_hash_for_primitive_types (group_hash.rs:555-570); x is the value
// cast `as u64` (sign-extended for signed types).
__host__ __device__ __forceinline__ uint64_t agg_hash_u64(uint64_t x) {
  x ^= x >> 32;
  x *= 0xd6e8feb86659fd93ULL;
  x ^= x >> 32;
  x *= 0xd6e8feb86659fd93ULL;
  x ^= x >> 32;
  return x;
}

// impl AggHash for [u8] (group_hash.rs:522-553): MurmurHash64A-style, but the
// tail bytes are folded big-endian-ordered: byte i of a tail of length L is
// shifted by 8*(L-i-1).
__host__ __device__ __forceinline__ uint64_t agg_hash_bytes(const uint8_t* p, uint32_t len) {
  const uint64_t M = 0xc6a4a7935bd1e995ULL;
  const uint64_t SEED = 0xe17a1465ULL;
  const int R = 47;
  uint64_t h = SEED ^ ((uint64_t)len * M);
  uint32_t nblocks = len / 8;
  for (uint32_t i = 0; i < nblocks; ++i) {
    uint64_t k = 0;
    for (int b = 0; b < 8; ++b) k |= (uint64_t)p[i * 8 + b] << (8 * b);  // read_unaligned LE
    k *= M;
    k ^= k >> R;
    k *= M;
    h ^= k;
    h *= M;
  }
  uint32_t tail = len - nblocks * 8;
  const uint8_t* t = p + nblocks * 8;
  for (uint32_t i = 0; i < tail; ++i) h ^= (uint64_t)t[i] << (8 * (tail - i - 1));
  h ^= h >> R;
  h *= M;
  h ^= h >> R;
  return h;
}

// Same hash for a string that sits inline in a 16-byte view (len <= 12):
// w1..w3 are the view's dwords 1..3 (bytes 0..11 of the string, little endian).
__device__ __forceinline__ uint64_t agg_hash_inline_view(uint32_t len, uint32_t w1, uint32_t w2,
                                                         uint32_t w3) {
  const uint64_t M = 0xc6a4a7935bd1e995ULL;
  const uint64_t SEED = 0xe17a1465ULL;
  const int R = 47;
  uint64_t h = SEED ^ ((uint64_t)len * M);
  uint64_t lo = ((uint64_t)w2 << 32) | w1;
  uint32_t tail_len = len;
  uint64_t tail_src = lo;  // little-endian bytes of the tail
  if (len >= 8) {
    uint64_t k = lo;
    k *= M;
    k ^= k >> R;
    k *= M;
    h ^= k;
    h *= M;
    tail_len = len - 8;
    tail_src = w3;
  }
  // tail byte i (LE position i) is shifted by 8*(tail_len-i-1): i.e. the byte-reversed
  // value of the low tail_len bytes.
  uint64_t folded = 0;
  for (uint32_t i = 0; i < 7; ++i) {
    if (i < tail_len) folded |= ((tail_src >> (8 * i)) & 0xff) << (8 * (tail_len - i - 1));
  }
  h ^= folded;
  h ^= h >> R;
  h *= M;
  h ^= h >> R;
  return h;
}

// i256 hashes as its 32 little-endian bytes (group_hash.rs:593-597): four 8-byte blocks, no tail.
__host__ __device__ __forceinline__ uint64_t agg_hash_i256(uint64_t w0, uint64_t w1, uint64_t w2, uint64_t w3) {
  const uint64_t M = 0xc6a4a7935bd1e995ULL;
  const uint64_t SEED = 0xe17a1465ULL;
  const int R = 47;
  uint64_t h = SEED ^ (32ULL * M);
  const uint64_t w[4] = {w0, w1, w2, w3};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    uint64_t k = w[i];
    k *= M;
    k ^= k >> R;
    k *= M;
    h ^= k;
    h *= M;
  }
  h ^= h >> R;
  h *= M;
  h ^= h >> R;
  return h;
}

// i128 hashes as its 16 little-endian bytes (group_hash.rs:587-591).
__device__ __forceinline__ uint64_t agg_hash_i128(i128 v) {
  const uint64_t M = 0xc6a4a7935bd1e995ULL;
  const uint64_t SEED = 0xe17a1465ULL;
  const int R = 47;
  uint64_t h = SEED ^ (16ULL * M);
  uint64_t w[2] = {(uint64_t)(u128)v, (uint64_t)((u128)v >> 64)};
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    uint64_t k = w[i];
    k *= M;
    k ^= k >> R;
    k *= M;
    h ^= k;
    h *= M;
  }
  h ^= h >> R;
  h *= M;
  h ^= h >> R;
  return h;
}

__host__ __device__ __forceinline__ uint64_t merge_hash(uint64_t a, uint64_t b) {
  return a * DBHIP_NULL_HASH_VAL ^ b;  // group_hash.rs:509-511
}

// ---------------------------------------------------------------------------
// bitmaps (LSB-first, src/common/column/src/bitmap/immutable.rs:78-85)
// ---------------------------------------------------------------------------
__device__ __forceinline__ bool bit_get(const uint8_t* bm, int64_t i) {
  return (bm[i >> 3] >> (i & 7)) & 1;
}

// ---------------------------------------------------------------------------
// wave-level helpers
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint64_t wave_sum_u64(uint64_t v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

__device__ __forceinline__ u128 wave_sum_u128(u128 v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    uint64_t lo = __shfl_xor((uint64_t)v, off, 64);
    uint64_t hi = __shfl_xor((uint64_t)(v >> 64), off, 64);
    v += ((u128)hi << 64) | lo;
  }
  return v;
}

// exact 192-bit wave sum: returns the low 128 bits, *ext accumulates bits 128..191
__device__ __forceinline__ u128 wave_sum_u192(u128 v, uint64_t* ext) {
  uint64_t e = *ext;
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    uint64_t lo = __shfl_xor((uint64_t)v, off, 64);
    uint64_t hi = __shfl_xor((uint64_t)(v >> 64), off, 64);
    uint64_t oe = __shfl_xor(e, off, 64);
    u128 o = ((u128)hi << 64) | lo;
    u128 r = v + o;
    e += oe + (r < v ? 1 : 0);
    v = r;
  }
  *ext = e;
  return v;
}

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

// ---------------------------------------------------------------------------
// 256-bit helper for Decimal128 rounding paths (types/decimal.rs:1024-1060 use
// ethnum i256). Sign-magnitude: callers split sign and magnitude first.
// ---------------------------------------------------------------------------
struct u256 {
  u128 lo, hi;
};

__host__ __device__ __forceinline__ u256 u256_mul_128(u128 a, u128 b) {
  uint64_t a0 = (uint64_t)a, a1 = (uint64_t)(a >> 64);
  uint64_t b0 = (uint64_t)b, b1 = (uint64_t)(b >> 64);
  u128 p00 = (u128)a0 * b0;
  u128 p01 = (u128)a0 * b1;
  u128 p10 = (u128)a1 * b0;
  u128 p11 = (u128)a1 * b1;
  u128 mid = (p00 >> 64) + (uint64_t)p01 + (uint64_t)p10;
  u256 r;
  r.lo = ((u128)(uint64_t)mid << 64) | (uint64_t)p00;
  r.hi = p11 + (p01 >> 64) + (p10 >> 64) + (mid >> 64);
  return r;
}

__host__ __device__ __forceinline__ u256 u256_add_128(u256 a, u128 b) {
  u256 r;
  r.lo = a.lo + b;
  r.hi = a.hi + (r.lo < a.lo ? 1 : 0);
  return r;
}

__host__ __device__ __forceinline__ bool u256_lt_128(u256 a, u128 b) { return a.hi == 0 && a.lo < b; }

// a -= b (b 128-bit), requires a >= b
__host__ __device__ __forceinline__ u256 u256_sub_128(u256 a, u128 b) {
  u256 r;
  r.lo = a.lo - b;
  r.hi = a.hi - (a.lo < b ? 1 : 0);
  return r;
}

// (u1 : u0) / v for u1 < v (the quotient fits 64 bits), exact. Round 5: TWO double-precision estimates with exact 128-bit remainders instead
// of Knuth's two 64 / 32 quotient digits — a 64-bit integer division has no instruction on CDNA (the compiler expands each into ~100
// instructions; the two-digit form needs two of them plus corrections: the decimal divide of a fused program came to ~3,800 instructions
// per four rows), a double division is ~15.
//   e  = (u1 2^64 + u0) / v in double: relative error < 2^-50, so the estimate q is within 2^14 of the quotient
//   r  = (u1 : u0) - q v exactly (128-bit, signed: |r| < 2^14 v < 2^78)
//   c  = floor(r / v) in double: |r / v| < 2^14 is far inside the 53 bits, so c is the remaining correction up to +-1
// and one compare-and-fix ends it. Checked against the 128-bit `/` of the host compiler over edge values and 2 x 10^7 random triples
// (tests/div_host_check.cpp).
__host__ __device__ inline uint64_t udiv_2by1(uint64_t u1, uint64_t u0, uint64_t v, uint64_t* rem) {
  const double dv = (double)v;
  const double e = ((double)u1 * 18446744073709551616.0 + (double)u0) / dv;
  uint64_t q = e >= 18446744073709549568.0 ? ~0ULL : (uint64_t)e;   // (the largest double below 2^64)
  const u128 n = ((u128)u1 << 64) | u0;
  i128 r = (i128)(n - (u128)q * (u128)v);
  // (sign and magnitude: a two's complement low word next to 2^64 would lose the small negative remainders in the conversion)
  const u128 mag = r < 0 ? (u128)0 - (u128)r : (u128)r;
  const double dm = (double)(uint64_t)(mag >> 64) * 18446744073709551616.0 + (double)(uint64_t)mag;
  const double er = (r < 0 ? -dm : dm) / dv;
  // floor without a library call (|er| < 2^15)
  int64_t c = (int64_t)er;
  if ((double)c > er) --c;
  q += (uint64_t)c;
  r -= (i128)c * (i128)(u128)v;
  if (r < 0) { --q; r += (i128)(u128)v; }
  else if (r >= (i128)(u128)v) { ++q; r -= (i128)(u128)v; }
  if (rem) *rem = (uint64_t)r;
  return q;
}

// n / d for a divisor that fits 64 bits
__host__ __device__ inline u128 udiv128_by_64(u128 n, uint64_t d, uint64_t* rem) {
  const uint64_t nh = (uint64_t)(n >> 64), nl = (uint64_t)n;
  uint64_t r = 0;
  const uint64_t qh = nh ? udiv_2by1(0, nh, d, &r) : 0;    // (nh / d the same way: no 64-bit integer division either)
  const uint64_t ql = udiv_2by1(r, nl, d, rem);
  return ((u128)qh << 64) | ql;
}

// q = a / d (d != 0); *rem gets the remainder. Numerators below 2^128 over divisors below 2^64 — every decimal division whose
// operands come from Decimal64 columns, and most others — take the two-digit path above; the rest the binary long division.
__host__ __device__ inline u256 u256_div_128(u256 a, u128 d, u128* rem) {
  u256 q;
  q.lo = 0;
  q.hi = 0;
  if (a.hi == 0) {
    if ((uint64_t)(d >> 64) == 0) {
      uint64_t r64;
      q.lo = udiv128_by_64(a.lo, (uint64_t)d, &r64);
      if (rem) *rem = r64;
      return q;
    }
    q.lo = a.lo / d;
    if (rem) *rem = a.lo % d;
    return q;
  }
  // r holds up to 129 bits: track overflow bit separately
  u128 r = 0;
  for (int i = 255; i >= 0; --i) {
    bool carry = (r >> 127) & 1;
    r <<= 1;
    u128 bit = (i >= 128) ? ((a.hi >> (i - 128)) & 1) : ((a.lo >> i) & 1);
    r |= bit;
    if (carry || r >= d) {
      r -= d;
      if (i >= 128)
        q.hi |= ((u128)1 << (i - 128));
      else
        q.lo |= ((u128)1 << i);
    }
  }
  if (rem) *rem = r;
  return q;
}

// 10^k as i128 for k in [0, 38]. Device: a table (k is wave-uniform at every call site -> one scalar load; the loop of up
// to 38 128-bit multiplies cost ~100 instructions per decimal node per row); host: the loop.
static __device__ const uint64_t kDbhipPow10[39][2] = {
  {0x0000000000000001ULL, 0x0000000000000000ULL},
  {0x000000000000000aULL, 0x0000000000000000ULL},
  {0x0000000000000064ULL, 0x0000000000000000ULL},
  {0x00000000000003e8ULL, 0x0000000000000000ULL},
  {0x0000000000002710ULL, 0x0000000000000000ULL},
  {0x00000000000186a0ULL, 0x0000000000000000ULL},
  {0x00000000000f4240ULL, 0x0000000000000000ULL},
  {0x0000000000989680ULL, 0x0000000000000000ULL},
  {0x0000000005f5e100ULL, 0x0000000000000000ULL},
  {0x000000003b9aca00ULL, 0x0000000000000000ULL},
  {0x00000002540be400ULL, 0x0000000000000000ULL},
  {0x000000174876e800ULL, 0x0000000000000000ULL},
  {0x000000e8d4a51000ULL, 0x0000000000000000ULL},
  {0x000009184e72a000ULL, 0x0000000000000000ULL},
  {0x00005af3107a4000ULL, 0x0000000000000000ULL},
  {0x00038d7ea4c68000ULL, 0x0000000000000000ULL},
  {0x002386f26fc10000ULL, 0x0000000000000000ULL},
  {0x016345785d8a0000ULL, 0x0000000000000000ULL},
  {0x0de0b6b3a7640000ULL, 0x0000000000000000ULL},
  {0x8ac7230489e80000ULL, 0x0000000000000000ULL},
  {0x6bc75e2d63100000ULL, 0x0000000000000005ULL},
  {0x35c9adc5dea00000ULL, 0x0000000000000036ULL},
  {0x19e0c9bab2400000ULL, 0x000000000000021eULL},
  {0x02c7e14af6800000ULL, 0x000000000000152dULL},
  {0x1bcecceda1000000ULL, 0x000000000000d3c2ULL},
  {0x161401484a000000ULL, 0x0000000000084595ULL},
  {0xdcc80cd2e4000000ULL, 0x000000000052b7d2ULL},
  {0x9fd0803ce8000000ULL, 0x00000000033b2e3cULL},
  {0x3e25026110000000ULL, 0x00000000204fce5eULL},
  {0x6d7217caa0000000ULL, 0x00000001431e0faeULL},
  {0x4674edea40000000ULL, 0x0000000c9f2c9cd0ULL},
  {0xc0914b2680000000ULL, 0x0000007e37be2022ULL},
  {0x85acef8100000000ULL, 0x000004ee2d6d415bULL},
  {0x38c15b0a00000000ULL, 0x0000314dc6448d93ULL},
  {0x378d8e6400000000ULL, 0x0001ed09bead87c0ULL},
  {0x2b878fe800000000ULL, 0x0013426172c74d82ULL},
  {0xb34b9f1000000000ULL, 0x00c097ce7bc90715ULL},
  {0x00f436a000000000ULL, 0x0785ee10d5da46d9ULL},
  {0x098a224000000000ULL, 0x4b3b4ca85a86c47aULL}};
__host__ __device__ __forceinline__ i128 pow10_i128(int k) {
#if defined(__HIP_DEVICE_COMPILE__)
  k = k < 0 ? 0 : (k > 38 ? 38 : k);
  return (i128)(((u128)kDbhipPow10[k][1] << 64) | kDbhipPow10[k][0]);
#else
  i128 r = 1;
  for (int i = 0; i < k; ++i) r *= 10;
  return r;
#endif
}

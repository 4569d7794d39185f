// k_scatter.hip — the hash-shuffle exchange's scatter indices (SURVEY §8e "hash join": co-partition both sides by
// hash(key) % n).
//
// Reference: HashFlightScatter / OneHashKeyFlightScatter (src/query/service/src/servers/flight/v1/scatter/
// flight_scatter_hash.rs:57-330) evaluate `siphash64(key) % scatter_size` per row (one key), or feed every key's siphash64
// into a std DefaultHasher and take `finish() % scatter_size` (several keys, combine_hash_keys :213-233); a NULL key gives the
// default scatter index (get_hash_values :258-310). `siphash64` (src/query/functions/src/scalars/hash.rs:50-122,323-328,
// DFHash :436-545; decimals: scalars/decimal/src/hash.rs:144-160) is SipHash-1-3 with keys (0, 0) — the `siphasher` crate
// (Cargo.lock: siphasher 1.0.1; absent from /root/reference, so the PUBLISHED algorithm is restated: Aumasson & Bernstein,
// "SipHash: a fast short-input PRF") over the value's bytes: integers / Date / Timestamp / float bit patterns in
// little-endian native width, Boolean as one byte, String as its bytes (no length, no terminator), Decimal as the scale byte
// followed by the value widened to i128 (precision <= 38 in any storage class). DefaultHasher::default() is the same
// SipHash-1-3 with zero keys (unspecified by std, but what the reference's nodes run). Pinned on the reference's golden
// values (tests/golden/siphash.json <- functions/tests/it/scalars/testdata/hash.txt, hash.rs:563-600 bucket_hash_v1 vectors).
// A GPU node that computes the same indices can sit in the reference's shuffle next to CPU nodes.
// One thread per row: the state is four registers, a fixed-width key is one or two compression rounds.
#include <string.h>
#include "dev_common.h"
#include "runtime.h"

using namespace dbhip;

namespace {

struct Sip {
  uint64_t v0, v1, v2, v3;
  __device__ __forceinline__ void init() {   // keys (0, 0)
    v0 = 0x736f6d6570736575ULL; v1 = 0x646f72616e646f6dULL; v2 = 0x6c7967656e657261ULL; v3 = 0x7465646279746573ULL;
  }
  static __device__ __forceinline__ uint64_t rotl(uint64_t x, int b) { return (x << b) | (x >> (64 - b)); }
  __device__ __forceinline__ void round() {
    v0 += v1; v1 = rotl(v1, 13); v1 ^= v0; v0 = rotl(v0, 32);
    v2 += v3; v3 = rotl(v3, 16); v3 ^= v2;
    v0 += v3; v3 = rotl(v3, 21); v3 ^= v0;
    v2 += v1; v1 = rotl(v1, 17); v1 ^= v2; v2 = rotl(v2, 32);
  }
  __device__ __forceinline__ void block(uint64_t m) { v3 ^= m; round(); v0 ^= m; }       // c = 1
  __device__ __forceinline__ uint64_t finish(uint64_t tail, uint64_t total_len) {          // d = 3
    block(tail | (total_len << 56));
    v2 ^= 0xff;
    round(); round(); round();
    return v0 ^ v1 ^ v2 ^ v3;
  }
};

struct SipCol {
  const void* data;
  const uint8_t* validity;
  int64_t voff;
  const void* const* buffers;
  int32_t type, is_scalar, scale;
};

// siphash64 of one value (the caller has checked validity)
__device__ __forceinline__ uint64_t sip_value(const SipCol& c, int64_t row) {
  const int64_t j = c.is_scalar ? 0 : row;
  Sip s;
  s.init();
  switch (c.type) {
    case DBHIP_T_BOOL: return s.finish(bit_get((const uint8_t*)c.data, j) ? 1 : 0, 1);
    case DBHIP_T_I8: case DBHIP_T_U8: return s.finish(((const uint8_t*)c.data)[j], 1);
    case DBHIP_T_I16: case DBHIP_T_U16: return s.finish(((const uint16_t*)c.data)[j], 2);
    case DBHIP_T_I32: case DBHIP_T_U32: case DBHIP_T_F32: case DBHIP_T_DATE: return s.finish(((const uint32_t*)c.data)[j], 4);
    case DBHIP_T_I64: case DBHIP_T_U64: case DBHIP_T_F64: case DBHIP_T_TIMESTAMP:
      s.block(((const uint64_t*)c.data)[j]);
      return s.finish(0, 8);
    case DBHIP_T_DEC64: case DBHIP_T_DEC128: case DBHIP_T_DEC256: {   // [scale u8][i128 little endian] = 17 bytes
      uint64_t lo, hi;
      if (c.type == DBHIP_T_DEC64) { lo = ((const uint64_t*)c.data)[j]; hi = (uint64_t)((int64_t)lo >> 63); }
      else { const uint64_t* p = (const uint64_t*)c.data + (c.type == DBHIP_T_DEC128 ? 2 : 4) * j; lo = p[0]; hi = p[1]; }
      s.block((uint64_t)(uint8_t)c.scale | (lo << 8));
      s.block((lo >> 56) | (hi << 8));
      return s.finish(hi >> 56, 17);
    }
    case DBHIP_T_STRING: {
      const uint32_t* v = (const uint32_t*)c.data + 4 * j;
      const uint32_t len = v[0];
      const uint8_t* p = len <= 12 ? (const uint8_t*)(v + 1) : (const uint8_t*)c.buffers[v[2]] + v[3];
      uint32_t at = 0;
      for (; at + 8 <= len; at += 8) {
        uint64_t m = 0;
#pragma unroll
        for (int b = 0; b < 8; ++b) m |= (uint64_t)p[at + b] << (8 * b);
        s.block(m);
      }
      uint64_t tail = 0;
      for (uint32_t b = 0; at + b < len; ++b) tail |= (uint64_t)p[at + b] << (8 * b);
      return s.finish(tail, len);
    }
  }
  return 0;
}

__global__ __launch_bounds__(256) void siphash64_kernel(SipCol c, int64_t n, uint64_t* out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const bool valid = !c.validity || bit_get(c.validity, c.voff + (c.is_scalar ? 0 : i));
    out[i] = valid ? sip_value(c, i) : 0;   // passthrough_nullable: the value under a NULL is not defined; 0 here
  }
}

// the longest value of a view column (a column handed over WITHOUT data buffers may only hold inline values)
__global__ __launch_bounds__(256) void sip_max_len_kernel(const uint32_t* views, int64_t n, uint32_t* out) {
  uint32_t m = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) m = views[4 * i] > m ? views[4 * i] : m;
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) { const uint32_t o = __shfl_xor(m, off, 64); m = o > m ? o : m; }
  if (lane_id() == 0 && m) atomicMax(out, m);
}

struct SipKeys { SipCol k[8]; int nkeys; };

constexpr int SCATTER_HIST_MAX = 4096;

// out_index[i] = the destination of row i; counts[d] = rows per destination
__global__ __launch_bounds__(256) void scatter_indices_kernel(SipKeys K, int64_t n, uint64_t m, uint64_t default_index, uint32_t* out_index,
                                                              unsigned long long* counts) {
  __shared__ uint32_t hist[SCATTER_HIST_MAX];
  const bool lds = m <= SCATTER_HIST_MAX;
  if (lds) for (uint32_t i = threadIdx.x; i < (uint32_t)m; i += blockDim.x) hist[i] = 0;
  __syncthreads();
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    uint64_t idx;
    if (K.nkeys == 1) {      // OneHashKeyFlightScatter: modulo(siphash(key), m), NULL -> default_scatter_index
      const SipCol& c = K.k[0];
      const bool valid = !c.validity || bit_get(c.validity, c.voff + (c.is_scalar ? 0 : i));
      idx = valid ? sip_value(c, i) % m : default_index;
    } else {                 // HashFlightScatter::combine_hash_keys: DefaultHasher over the keys' siphash64 (NULL -> 0), % m
      Sip h;
      h.init();
      for (int k = 0; k < K.nkeys; ++k) {
        const SipCol& c = K.k[k];
        const bool valid = !c.validity || bit_get(c.validity, c.voff + (c.is_scalar ? 0 : i));
        h.block(valid ? sip_value(c, i) : 0);
      }
      idx = h.finish(0, (uint64_t)(8 * K.nkeys) & 0xff) % m;
    }
    out_index[i] = (uint32_t)idx;
    if (idx < m) { if (lds) atomicAdd(&hist[idx], 1u); else atomicAdd(&counts[idx], 1ULL); }
  }
  __syncthreads();
  if (lds) for (uint32_t i = threadIdx.x; i < (uint32_t)m; i += blockDim.x) if (hist[i]) atomicAdd(&counts[i], (unsigned long long)hist[i]);
}

int32_t make_sip_col(const dbhip_col& c, int64_t n, hipStream_t s, const char* fn, int k, SipCol* out) {
  const int t = c.type;
  if (t == DBHIP_T_STRING && !c.buffers && n > 0) {   // inline views only: verify, a long view would dereference a missing buffer table
    uint32_t* flag = (uint32_t*)scratch(64, 9, s);
    if (!flag) return DBHIP_ERR_HIP;
    DBHIP_CHECK(hipMemsetAsync(flag, 0, 4, s));
    const int64_t rows = c.is_scalar ? 1 : n;
    hipLaunchKernelGGL(sip_max_len_kernel, dim3(grid_for(rows, 256)), dim3(256), 0, s, (const uint32_t*)c.data, rows, flag);
    uint32_t mx = 0;
    DBHIP_CHECK(hipMemcpyAsync(&mx, flag, 4, hipMemcpyDeviceToHost, s));
    DBHIP_CHECK(hipStreamSynchronize(s));
    if (mx > 12) { set_error("%s: string key %d holds values longer than 12 bytes but no data buffers", fn, k); return DBHIP_ERR_INVALID; }
  }
  const bool ok = (t >= DBHIP_T_BOOL && t <= DBHIP_T_STRING) || t == DBHIP_T_DEC256;
  if (!ok) { set_error("%s: key %d has unsupported type %d", fn, k, t); return DBHIP_ERR_UNSUPPORTED; }
  if ((t == DBHIP_T_DEC64 || t == DBHIP_T_DEC128 || t == DBHIP_T_DEC256) && (c.precision < 1 || c.precision > 38)) {
    set_error("%s: key %d: Decimal(%d, %d) — siphash64 over a precision above 38 hashes the i256 image (decimal/src/hash.rs:172-186), not built; "
              "precision and scale must be set", fn, k, c.precision, c.scale);
    return DBHIP_ERR_UNSUPPORTED;
  }
  *out = SipCol{c.data, c.validity, c.validity_offset, c.buffers, t, c.is_scalar, c.scale};
  return DBHIP_OK;
}

}  // namespace

extern "C" {

int32_t dbhip_siphash64(const dbhip_col* col, int64_t n, uint64_t* out, void* stream) {
  DBHIP_REQUIRE(col && n >= 0, "dbhip_siphash64: bad argument");
  if (n == 0) return DBHIP_OK;
  DBHIP_REQUIRE(out && col->data, "dbhip_siphash64: NULL buffer");
  SipCol c;
  hipStream_t s = resolve_stream(stream);
  int32_t rc = make_sip_col(*col, n, s, "dbhip_siphash64", 0, &c);
  if (rc) return rc;
  hipLaunchKernelGGL(siphash64_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, c, n, out);
  DBHIP_LAUNCH_CHECK();
  return DBHIP_OK;
}

int32_t dbhip_scatter_indices(const dbhip_col* keys, int32_t nkeys, int64_t n, uint32_t scatter_size, uint64_t default_index,
                              uint32_t* out_index, uint64_t* out_counts, void* stream) {
  DBHIP_REQUIRE(keys && nkeys >= 1 && nkeys <= 8 && n >= 0, "dbhip_scatter_indices: 1..8 hash keys");
  DBHIP_REQUIRE(scatter_size >= 1 && out_counts, "dbhip_scatter_indices: scatter_size must be positive, counts not NULL");
  DBHIP_REQUIRE(default_index < scatter_size, "dbhip_scatter_indices: the default scatter index must be a destination");
  hipStream_t s = resolve_stream(stream);
  DBHIP_CHECK(hipMemsetAsync(out_counts, 0, (size_t)scatter_size * 8, s));
  if (n == 0) return DBHIP_OK;
  DBHIP_REQUIRE(out_index, "dbhip_scatter_indices: NULL out");
  SipKeys K;
  memset(&K, 0, sizeof(K));
  K.nkeys = nkeys;
  for (int k = 0; k < nkeys; ++k) {
    int32_t rc = make_sip_col(keys[k], n, s, "dbhip_scatter_indices", k, &K.k[k]);
    if (rc) return rc;
  }
  hipLaunchKernelGGL(scatter_indices_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, K, n, (uint64_t)scatter_size, default_index, out_index,
                     (unsigned long long*)out_counts);
  DBHIP_LAUNCH_CHECK();
  return DBHIP_OK;
}

}  // extern "C"

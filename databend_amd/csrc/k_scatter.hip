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

#include <vector>

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


// ---------------------------------------------------------------------------------------------------------------------
// DataBlock::scatter over whole columns (values + validity Bitmaps + Boolean / String / Decimal256 columns) and
// DataBlock::concat — the two ends of every exchange (kernels/scatter.rs:20-66, kernels/concat.rs:62-340).
// ---------------------------------------------------------------------------------------------------------------------
struct B32 { uint64_t w[4]; };

__global__ __launch_bounds__(256) void sc_iota_kernel(uint32_t* out, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) out[i] = (uint32_t)i;
}

__global__ __launch_bounds__(256) void sc_hist_kernel(const uint32_t* __restrict__ index, int64_t n, uint32_t m, unsigned long long* counts,
                                                      unsigned long long* bad) {
  __shared__ uint32_t hist[SCATTER_HIST_MAX];
  const bool lds = m <= SCATTER_HIST_MAX;
  if (lds) for (uint32_t i = threadIdx.x; i < m; i += 256) hist[i] = 0;
  __syncthreads();
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const uint32_t d = index[i];
    if (d >= m) { atomicAdd(bad, 1ULL); continue; }
    if (lds) atomicAdd(&hist[d], 1u); else atomicAdd(&counts[d], 1ULL);
  }
  __syncthreads();
  if (lds) for (uint32_t i = threadIdx.x; i < m; i += 256) if (hist[i]) atomicAdd(&counts[i], (unsigned long long)hist[i]);
}

__global__ __launch_bounds__(256) void sc_gather32_kernel(const B32* __restrict__ src, const uint32_t* __restrict__ perm, int64_t n, B32* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) out[i] = src[perm[i]];
}

// Bitmaps of the scattered block: destination d's rows are perm[row_start[d] .. row_start[d + 1]); its Bitmap is a
// stand-alone one that starts on the 64-bit word word_start[d] = row_start[d] / 64 + d of `out` (destinations never share a
// word, so each is a valid offset-0 Bitmap for every other entry point). One wave per output word, one ballot per word.
__global__ __launch_bounds__(256) void sc_scatter_bits_kernel(const uint8_t* __restrict__ src, int64_t src_off, const uint32_t* __restrict__ perm,
                                                              const int64_t* __restrict__ row_start, uint32_t m, int64_t total_words,
                                                              uint64_t* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  for (int64_t W = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6; W < total_words; W += ((int64_t)gridDim.x * 256) >> 6) {
    // destination of word W: the last d with word_start[d] <= W
    uint32_t lo = 0, hi = m;   // invariant: word_start[lo] <= W, answer in [lo, hi)
    while (hi - lo > 1) {
      const uint32_t mid = (lo + hi) >> 1;
      if ((row_start[mid] >> 6) + (int64_t)mid <= W) lo = mid; else hi = mid;
    }
    const int64_t r = row_start[lo] + ((W - ((row_start[lo] >> 6) + (int64_t)lo)) << 6) + lane;
    bool bit = false;
    if (r < row_start[lo + 1]) bit = bit_get(src, src_off + (int64_t)perm[r]);
    const uint64_t word = __ballot(bit);
    if (lane == 0) out[W] = word;
  }
}

// up to 64 bits of a Bitmap starting at any bit (only bytes that hold wanted bits are touched)
__device__ __forceinline__ uint64_t sc_read_bits(const uint8_t* p, int64_t bitpos, int nbits) {
  const uint8_t* b = p + (bitpos >> 3);
  const int sh = (int)(bitpos & 7);
  const int nbytes = (sh + nbits + 7) >> 3;   // <= 9
  uint64_t lo = 0;
  for (int k = 0; k < nbytes && k < 8; ++k) lo |= (uint64_t)b[k] << (8 * k);
  uint64_t v = lo >> sh;
  if (nbytes == 9) v |= (uint64_t)b[8] << (64 - sh);
  return nbits == 64 ? v : v & ((1ULL << nbits) - 1);
}

constexpr int CONCAT_MAX_BLOCKS = 64;   // blocks per launch (more: several launches over the same output)
struct ConcatBits {
  const uint8_t* src[CONCAT_MAX_BLOCKS];   // NULL = all ones (a block without validity)
  int64_t src_off[CONCAT_MAX_BLOCKS];
  int64_t out_start[CONCAT_MAX_BLOCKS + 1];   // output bit of the block's first row; [nblocks] = end
  int nblocks;
};
// Output words of this launch's blocks; the first word, when an earlier launch's blocks end inside it, keeps their bits.
__global__ __launch_bounds__(256) void sc_concat_bits_kernel(ConcatBits A, uint64_t* __restrict__ out, int merge_first) {
  const int64_t w0 = A.out_start[0] >> 6, w1 = (A.out_start[A.nblocks] + 63) >> 6;
  for (int64_t w = w0 + (int64_t)blockIdx.x * 256 + threadIdx.x; w < w1; w += (int64_t)gridDim.x * 256) {
    const int64_t lo_bit = w << 6, hi_bit = lo_bit + 64;
    uint64_t word = (merge_first && w == w0) ? out[w] & ((1ULL << (A.out_start[0] & 63)) - 1) : 0;
    for (int b = 0; b < A.nblocks; ++b) {
      const int64_t s = A.out_start[b] > lo_bit ? A.out_start[b] : lo_bit;
      const int64_t e = A.out_start[b + 1] < hi_bit ? A.out_start[b + 1] : hi_bit;
      if (s >= e) continue;
      const int nb = (int)(e - s);
      const uint64_t bits = A.src[b] ? sc_read_bits(A.src[b], A.src_off[b] + (s - A.out_start[b]), nb) : (nb == 64 ? ~0ULL : ((1ULL << nb) - 1));
      word |= bits << (s - lo_bit);
    }
    out[w] = word;
  }
}

// views of one block with the buffer index of long values rebased (the concatenated column's buffer table is the blocks' tables
// back to back); inline values (<= 12 bytes) are copied as they are
__global__ __launch_bounds__(256) void sc_rebase_views_kernel(const uint32_t* __restrict__ src, int64_t n, uint32_t buffer_base, int is_scalar,
                                                              uint32_t* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const uint4 v = ((const uint4*)src)[is_scalar ? 0 : i];
    uint4 o = v;
    if (v.x > 12) o.z = v.z + buffer_base;
    ((uint4*)out)[i] = o;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void sc_fill_kernel(const T* __restrict__ one, int64_t n, T* __restrict__ out) {
  const T v = one[0];
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) out[i] = v;
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

int32_t dbhip_scatter_columns(const dbhip_col* cols, int32_t ncols, const uint32_t* index, int64_t n, uint32_t scatter_size,
                              void* const* out_data_host, uint8_t* const* out_validity_host, int64_t* out_row_starts_host, void* stream) {
  return dbhip_scatter_columns_counted_internal(cols, ncols, index, n, scatter_size, out_data_host, out_validity_host, out_row_starts_host, nullptr, stream);
}

}  // extern "C"
// rows per destination, left on the device: counts_dev[scatter_size] + one counter of indices that are not below scatter_size
// (k_comm.hip posts the exchange of the counts from here, before any column is scattered)
int32_t dbhip_scatter_count_internal(const uint32_t* index, int64_t n, uint32_t scatter_size, uint64_t* counts_dev, hipStream_t s) {
  DBHIP_REQUIRE(n >= 0 && n < 0xFFFFFFFFLL && scatter_size >= 1 && scatter_size <= (1u << 24) && counts_dev && (index || n == 0), "dbhip_exchange_begin: bad argument");
  DBHIP_CHECK(hipMemsetAsync(counts_dev, 0, (size_t)(scatter_size + 1) * 8, s));
  if (n > 0) {
    hipLaunchKernelGGL(sc_hist_kernel, dim3(grid_for(n, 256, 1024)), dim3(256), 0, s, index, n, scatter_size, (unsigned long long*)counts_dev,
                       (unsigned long long*)counts_dev + scatter_size);
    DBHIP_LAUNCH_CHECK();
  }
  return DBHIP_OK;
}
// `known_counts_host` (scatter_size + 1 words of dbhip_scatter_count_internal, already read back by the caller): no histogram, no drain here
int32_t dbhip_scatter_columns_counted_internal(const dbhip_col* cols, int32_t ncols, const uint32_t* index, int64_t n, uint32_t scatter_size,
                                               void* const* out_data_host, uint8_t* const* out_validity_host, int64_t* out_row_starts_host,
                                               const uint64_t* known_counts_host, void* stream) {
  DBHIP_REQUIRE(ncols >= 0 && n >= 0 && n < 0xFFFFFFFFLL && scatter_size >= 1 && scatter_size <= (1u << 24), "dbhip_scatter_columns: bad argument");
  DBHIP_REQUIRE(out_row_starts_host && (ncols == 0 || (cols && out_data_host && out_validity_host)), "dbhip_scatter_columns: NULL argument");
  for (int c = 0; c < ncols; ++c) {
    const int t = cols[c].type;
    if (!((t >= DBHIP_T_BOOL && t <= DBHIP_T_STRING) || t == DBHIP_T_DEC256) || cols[c].is_scalar) {
      set_error("dbhip_scatter_columns: column %d: unsupported type %d or a scalar (a constant entry scatters as itself)", c, t);
      return DBHIP_ERR_UNSUPPORTED;
    }
    DBHIP_REQUIRE(n == 0 || (cols[c].data && out_data_host[c]), "dbhip_scatter_columns: NULL column buffer");
    DBHIP_REQUIRE(n == 0 || !cols[c].validity || out_validity_host[c], "dbhip_scatter_columns: a nullable column needs an output validity buffer");
    DBHIP_REQUIRE(n == 0 || !cols[c].validity || ((uintptr_t)out_validity_host[c] & 7) == 0, "dbhip_scatter_columns: output Bitmaps must be 8-byte aligned");
    DBHIP_REQUIRE(n == 0 || t != DBHIP_T_BOOL || ((uintptr_t)out_data_host[c] & 7) == 0, "dbhip_scatter_columns: output Bitmaps must be 8-byte aligned");
  }
  hipStream_t s = resolve_stream(stream);
  // rows per destination -> row starts (host and device)
  const size_t meta_bytes = (size_t)(scatter_size + 1) * 8 * 2 + 64;
  uint8_t* meta = (uint8_t*)scratch(meta_bytes, 16, s);
  if (!meta) return DBHIP_ERR_HIP;
  unsigned long long* counts = (unsigned long long*)meta;          // [scatter_size] + [1] out-of-range counter
  int64_t* row_start = (int64_t*)(meta + (size_t)(scatter_size + 1) * 8);   // [scatter_size + 1]
  std::vector<unsigned long long> hc((size_t)scatter_size + 1);
  if (known_counts_host) {
    DBHIP_REQUIRE(n == 0 || index, "dbhip_scatter_columns: NULL index");
    for (size_t d = 0; d < hc.size(); ++d) hc[d] = known_counts_host[d];
  } else {
    int32_t rc = dbhip_scatter_count_internal(index, n, scatter_size, (uint64_t*)counts, s);
    if (rc) return rc;
    DBHIP_CHECK(hipMemcpyAsync(hc.data(), counts, hc.size() * 8, hipMemcpyDeviceToHost, s));
    DBHIP_CHECK(hipStreamSynchronize(s));
  }
  if (hc[scatter_size]) {
    set_error("dbhip_scatter_columns: %llu indices are not below scatter_size %u", hc[scatter_size], scatter_size);
    return DBHIP_ERR_INVALID;
  }
  out_row_starts_host[0] = 0;
  for (uint32_t d = 0; d < scatter_size; ++d) out_row_starts_host[d + 1] = out_row_starts_host[d] + (int64_t)hc[d];
  if (n == 0 || ncols == 0) return DBHIP_OK;
  DBHIP_CHECK(hipMemcpyAsync(row_start, out_row_starts_host, (size_t)(scatter_size + 1) * 8, hipMemcpyHostToDevice, s));
  // the stable permutation (output row -> source row): the scatter of the row numbers themselves
  uint32_t* ws = (uint32_t*)scratch((size_t)n * 8 + 64, 17, s);
  if (!ws) return DBHIP_ERR_HIP;
  uint32_t* iota = ws;
  uint32_t* perm = ws + n;
  hipLaunchKernelGGL(sc_iota_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, iota, n);
  {
    const void* src1[1] = {iota};
    void* out1[1] = {perm};
    const int32_t es1[1] = {4};
    int32_t rc = dbhip_scatter_block(src1, es1, 1, index, n, scatter_size, out1, stream);
    if (rc) return rc;
  }
  // value buffers: fixed-width columns and String views through the block take (<= 8 columns per launch); Decimal256 on its own
  const void* srcs[8];
  void* outs[8];
  int32_t es[8];
  int g = 0;
  auto flush = [&]() -> int32_t {
    if (!g) return DBHIP_OK;
    int32_t rc = dbhip_take_block(srcs, es, g, perm, n, outs, stream);
    g = 0;
    return rc;
  };
  const int64_t total_words = (n >> 6) + (int64_t)scatter_size + 1;
  const int bits_grid = grid_for(total_words * 64, 256);
  for (int c = 0; c < ncols; ++c) {
    const int t = cols[c].type;
    int32_t rc;
    if (t == DBHIP_T_BOOL) {
      hipLaunchKernelGGL(sc_scatter_bits_kernel, dim3(bits_grid), dim3(256), 0, s, (const uint8_t*)cols[c].data, (int64_t)0, perm, row_start, scatter_size,
                         total_words, (uint64_t*)out_data_host[c]);
    } else if (t == DBHIP_T_DEC256) {
      hipLaunchKernelGGL(sc_gather32_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, (const B32*)cols[c].data, perm, n, (B32*)out_data_host[c]);
    } else {
      srcs[g] = cols[c].data; outs[g] = out_data_host[c]; es[g] = type_size(t); ++g;
      if (g == 8 && (rc = flush())) return rc;
    }
    if (cols[c].validity)
      hipLaunchKernelGGL(sc_scatter_bits_kernel, dim3(bits_grid), dim3(256), 0, s, cols[c].validity, cols[c].validity_offset, perm, row_start,
                         scatter_size, total_words, (uint64_t*)out_validity_host[c]);
  }
  int32_t rc = flush();
  if (rc) return rc;
  DBHIP_LAUNCH_CHECK();
  DBHIP_CHECK(hipStreamSynchronize(s));  // scratch is reused by the next call
  return DBHIP_OK;
}
extern "C" {

int32_t dbhip_concat_columns(const dbhip_col* cols, const int64_t* rows_host, const int64_t* bool_bit_offsets_host, int32_t nblocks,
                             void* out_data, uint8_t* out_validity, const void** out_buffers_dev, int32_t* out_n_buffers_host, void* stream) {
  DBHIP_REQUIRE(nblocks >= 1 && cols && rows_host, "dbhip_concat_columns: at least one block");
  const int t = cols[0].type;
  int64_t total = 0;
  bool any_validity = false;
  int64_t nbuf = 0;
  for (int b = 0; b < nblocks; ++b) {
    DBHIP_REQUIRE(cols[b].type == t, "dbhip_concat_columns: the blocks' columns must have one type (DataBlock::concat checks the schema)");
    DBHIP_REQUIRE(rows_host[b] >= 0, "dbhip_concat_columns: negative row count");
    DBHIP_REQUIRE(rows_host[b] == 0 || cols[b].data, "dbhip_concat_columns: NULL column buffer");
    total += rows_host[b];
    any_validity |= cols[b].validity != nullptr;
    if (t == DBHIP_T_STRING) nbuf += cols[b].n_buffers;
  }
  if (!((t >= DBHIP_T_BOOL && t <= DBHIP_T_STRING) || t == DBHIP_T_DEC256)) { set_error("dbhip_concat_columns: unsupported type %d", t); return DBHIP_ERR_UNSUPPORTED; }
  DBHIP_REQUIRE(total < 0xFFFFFFFFLL && nbuf < (1LL << 31), "dbhip_concat_columns: more than 2^32 rows / 2^31 buffers");
  if (out_n_buffers_host) *out_n_buffers_host = (int32_t)nbuf;
  if (total == 0) return DBHIP_OK;
  DBHIP_REQUIRE(out_data && (!any_validity || out_validity), "dbhip_concat_columns: NULL output (a nullable block needs out_validity)");
  DBHIP_REQUIRE(t != DBHIP_T_STRING || nbuf == 0 || out_buffers_dev, "dbhip_concat_columns: string blocks with data buffers need out_buffers_dev");
  DBHIP_REQUIRE((((uintptr_t)out_validity) & 7) == 0 && (t != DBHIP_T_BOOL || (((uintptr_t)out_data) & 7) == 0), "dbhip_concat_columns: output Bitmaps must be 8-byte aligned");
  hipStream_t s = resolve_stream(stream);
  const int esz = type_size(t);
  // values
  if (t == DBHIP_T_BOOL) {
    for (int b0 = 0; b0 < nblocks; b0 += CONCAT_MAX_BLOCKS) {
      ConcatBits A;
      memset(&A, 0, sizeof(A));
      A.nblocks = nblocks - b0 < CONCAT_MAX_BLOCKS ? nblocks - b0 : CONCAT_MAX_BLOCKS;
      int64_t at = 0;
      for (int b = 0; b < b0; ++b) at += rows_host[b];
      for (int k = 0; k < A.nblocks; ++k) {
        DBHIP_REQUIRE(!cols[b0 + k].is_scalar, "dbhip_concat_columns: a constant Boolean entry must be expanded by the caller");
        A.src[k] = (const uint8_t*)cols[b0 + k].data;
        A.src_off[k] = bool_bit_offsets_host ? bool_bit_offsets_host[b0 + k] : 0;
        A.out_start[k] = at;
        at += rows_host[b0 + k];
      }
      A.out_start[A.nblocks] = at;
      const int64_t words = ((at + 63) >> 6) - (A.out_start[0] >> 6);
      if (words > 0) hipLaunchKernelGGL(sc_concat_bits_kernel, dim3(grid_for(words, 256)), dim3(256), 0, s, A, (uint64_t*)out_data, b0 > 0 ? 1 : 0);
    }
  } else {
    int64_t at = 0, bbase = 0;
    for (int b = 0; b < nblocks; ++b) {
      const int64_t r = rows_host[b];
      uint8_t* dst = (uint8_t*)out_data + (size_t)at * esz;
      if (r > 0) {
        if (t == DBHIP_T_STRING) {
          hipLaunchKernelGGL(sc_rebase_views_kernel, dim3(grid_for(r, 256)), dim3(256), 0, s, (const uint32_t*)cols[b].data, r, (uint32_t)bbase,
                             cols[b].is_scalar, (uint32_t*)dst);
        } else if (cols[b].is_scalar) {
          const dim3 g(grid_for(r, 256)), blk(256);
          switch (esz) {
            case 1: hipLaunchKernelGGL(sc_fill_kernel<uint8_t>, g, blk, 0, s, (const uint8_t*)cols[b].data, r, (uint8_t*)dst); break;
            case 2: hipLaunchKernelGGL(sc_fill_kernel<uint16_t>, g, blk, 0, s, (const uint16_t*)cols[b].data, r, (uint16_t*)dst); break;
            case 4: hipLaunchKernelGGL(sc_fill_kernel<uint32_t>, g, blk, 0, s, (const uint32_t*)cols[b].data, r, (uint32_t*)dst); break;
            case 8: hipLaunchKernelGGL(sc_fill_kernel<uint64_t>, g, blk, 0, s, (const uint64_t*)cols[b].data, r, (uint64_t*)dst); break;
            case 16: hipLaunchKernelGGL(sc_fill_kernel<uint4>, g, blk, 0, s, (const uint4*)cols[b].data, r, (uint4*)dst); break;
            default: hipLaunchKernelGGL(sc_fill_kernel<B32>, g, blk, 0, s, (const B32*)cols[b].data, r, (B32*)dst); break;
          }
        } else {
          DBHIP_CHECK(hipMemcpyAsync(dst, cols[b].data, (size_t)r * esz, hipMemcpyDeviceToDevice, s));
        }
      }
      if (t == DBHIP_T_STRING && cols[b].n_buffers > 0) {
        DBHIP_REQUIRE(cols[b].buffers, "dbhip_concat_columns: n_buffers > 0 but no buffer table");
        DBHIP_CHECK(hipMemcpyAsync((void*)(out_buffers_dev + bbase), cols[b].buffers, (size_t)cols[b].n_buffers * sizeof(void*), hipMemcpyDeviceToDevice, s));
        bbase += cols[b].n_buffers;
      }
      at += r;
    }
  }
  // validity (concat.rs: a block that is not nullable where another is counts as all valid)
  if (any_validity) {
    for (int b0 = 0; b0 < nblocks; b0 += CONCAT_MAX_BLOCKS) {
      ConcatBits A;
      memset(&A, 0, sizeof(A));
      A.nblocks = nblocks - b0 < CONCAT_MAX_BLOCKS ? nblocks - b0 : CONCAT_MAX_BLOCKS;
      int64_t at = 0;
      for (int b = 0; b < b0; ++b) at += rows_host[b];
      for (int k = 0; k < A.nblocks; ++k) {
        DBHIP_REQUIRE(!(cols[b0 + k].is_scalar && cols[b0 + k].validity), "dbhip_concat_columns: a constant entry with a validity must be expanded by the caller");
        A.src[k] = cols[b0 + k].validity;
        A.src_off[k] = cols[b0 + k].validity_offset;
        A.out_start[k] = at;
        at += rows_host[b0 + k];
      }
      A.out_start[A.nblocks] = at;
      const int64_t words = ((at + 63) >> 6) - (A.out_start[0] >> 6);
      if (words > 0) hipLaunchKernelGGL(sc_concat_bits_kernel, dim3(grid_for(words, 256)), dim3(256), 0, s, A, (uint64_t*)out_validity, b0 > 0 ? 1 : 0);
    }
  }
  DBHIP_LAUNCH_CHECK();
  return DBHIP_OK;
}

}  // extern "C"

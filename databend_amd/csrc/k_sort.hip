// k_sort.hip — sort permutation.
//
// Reference: DataBlock::sort_with_type -> SortCompare (kernels/sort.rs:91-113,
// kernels/sort_compare.rs:33-283): a u32 permutation refined column by column with
// sort_unstable_by, asc/desc and nulls-first/last per column, optional row limit
// (LimitType::LimitRows, :197-209). Only the KEY SEQUENCE is defined (ties are unordered there).
// Device algorithm: LSD radix sort, stable, 8 bits per pass, over order-preserving u64 encodings
// (the device analogue of the reference's fixed-width row encoding, sorts/core/row_convert/fixed.rs):
// keys are processed from the last to the first; for each key an encode kernel reads the column
// THROUGH the current permutation, then one pass per key byte:
//   histogram   per-tile digit counts in LDS                -> hist[digit][tile]
//   scan        device-wide exclusive scan (dev_scan.h)     -> base offset of (digit, tile)
//   scatter     wave-level digit matching with 8 ballots gives each key its stable rank inside
//               the tile; (key, row id) pairs move to their final position of this pass
// Nullable keys get one extra (most significant) pass on the null flag. Ties end up in ascending
// row-id order (stable), which is one of the orders the reference may produce.
#include "dev_common.h"
#include "dev_scan.h"
#include "runtime.h"

using namespace dbhip;

namespace {

constexpr int SORT_ITEMS = 8;                 // keys per thread
constexpr int SORT_TILE = 256 * SORT_ITEMS;   // keys per block

struct SortCol {
  const void* data;
  const uint8_t* validity;
  int64_t voff;
  int type;
  int desc;
  int nulls_first;
};

// order-preserving u64 image of one value (ascending)
__device__ __forceinline__ uint64_t sort_encode(const SortCol& c, uint32_t row, int part) {
  switch (c.type) {
    case DBHIP_T_BOOL: return bit_get((const uint8_t*)c.data, row);
    case DBHIP_T_I8: return (uint64_t)(uint8_t)(((const int8_t*)c.data)[row] ^ 0x80);
    case DBHIP_T_I16: return (uint64_t)(uint16_t)(((const int16_t*)c.data)[row] ^ 0x8000);
    case DBHIP_T_I32: case DBHIP_T_DATE: return (uint64_t)(uint32_t)(((const int32_t*)c.data)[row] ^ 0x80000000);
    case DBHIP_T_I64: case DBHIP_T_TIMESTAMP: case DBHIP_T_DEC64:
      return ((const uint64_t*)c.data)[row] ^ 0x8000000000000000ULL;
    case DBHIP_T_U8: return ((const uint8_t*)c.data)[row];
    case DBHIP_T_U16: return ((const uint16_t*)c.data)[row];
    case DBHIP_T_U32: return ((const uint32_t*)c.data)[row];
    case DBHIP_T_U64: return ((const uint64_t*)c.data)[row];
    case DBHIP_T_F32: {  // OrderedFloat: NaN largest, -0.0 == 0.0
      float f = ((const float*)c.data)[row];
      if (f != f) return 0xFFFFFFFFULL;
      if (f == 0.0f) f = 0.0f;
      uint32_t b = __float_as_uint(f);
      return (uint64_t)((b >> 31) ? ~b : (b | 0x80000000u));
    }
    case DBHIP_T_F64: {
      double f = ((const double*)c.data)[row];
      if (f != f) return ~0ULL;
      if (f == 0.0) f = 0.0;
      uint64_t b = (uint64_t)__double_as_longlong(f);
      return (b >> 63) ? ~b : (b | 0x8000000000000000ULL);
    }
    case DBHIP_T_DEC128: {
      const uint64_t* p = (const uint64_t*)c.data + 2 * (uint64_t)row;
      return part == 0 ? p[0] : (p[1] ^ 0x8000000000000000ULL);
    }
    case DBHIP_T_STRING: {
      // inline view {len, 12 bytes}: memcmp order of the zero-padded bytes, then the length
      // (a proper prefix sorts first) — the image of the reference's variable row encoding
      // (sorts/core/row_convert/variable.rs) for strings of at most 12 bytes.
      // part 1 (most significant): bytes 0..7 big-endian; part 0: bytes 8..11 big-endian << 32 | len
      const uint32_t* v = (const uint32_t*)c.data + 4 * (uint64_t)row;
      const uint32_t len = v[0];
      uint32_t d1 = v[1], d2 = v[2], d3 = v[3];
      const uint32_t l = len > 12 ? 4 : len;  // long strings: only the 4-byte prefix is inline (flagged by the caller)
      if (l < 4) { d1 &= (l == 0) ? 0u : (0xffffffffu >> (8 * (4 - l))); d2 = 0; d3 = 0; }
      else if (l < 8) { d2 &= (l == 4) ? 0u : (0xffffffffu >> (8 * (8 - l))); d3 = 0; }
      else if (l < 12) { d3 &= (l == 8) ? 0u : (0xffffffffu >> (8 * (12 - l))); }
      if (part == 1) return ((uint64_t)__builtin_bswap32(d1) << 32) | __builtin_bswap32(d2);
      return ((uint64_t)__builtin_bswap32(d3) << 32) | len;
    }
  }
  return 0;
}

__host__ __device__ inline int sort_key_bytes(int type) {
  switch (type) {
    case DBHIP_T_BOOL: case DBHIP_T_I8: case DBHIP_T_U8: return 1;
    case DBHIP_T_I16: case DBHIP_T_U16: return 2;
    case DBHIP_T_I32: case DBHIP_T_U32: case DBHIP_T_F32: case DBHIP_T_DATE: return 4;
    default: return 8;
  }
}

// flags strings that are not inline (len > 12): their order needs the data buffer, not built yet
__global__ __launch_bounds__(256) void sort_check_inline_kernel(const uint32_t* views, int64_t n, uint32_t* flag) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    if (views[4 * i] > 12) *flag = 1u;
}

__global__ __launch_bounds__(256) void sort_iota_kernel(uint32_t* perm, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    perm[i] = (uint32_t)i;
}

// enc[j] = image of col[perm[j]]; mode 0: value part 0, 1: value part 1 (dec128 high), 2: null flag
__global__ __launch_bounds__(256) void sort_encode_kernel(SortCol c, const uint32_t* perm, int64_t n, int mode,
                                                          uint64_t* enc) {
  for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (int64_t)gridDim.x * blockDim.x) {
    uint32_t row = perm[j];
    bool valid = !c.validity || bit_get(c.validity, c.voff + row);
    uint64_t e;
    if (mode == 2) {
      e = (valid ? 1 : 0) ^ (c.nulls_first ? 0 : 1);  // nulls first: null -> 0 ; nulls last: null -> 1
    } else {
      e = valid ? sort_encode(c, row, mode) : 0;
      if (c.desc) e = ~e;
      if (!valid) e = 0;  // NULL rows tie on the value passes; the flag pass places them
    }
    enc[j] = e;
  }
}

__global__ __launch_bounds__(256) void sort_hist_kernel(const uint64_t* keys, int64_t n, int shift, uint32_t* hist,
                                                        int64_t ntiles) {
  __shared__ uint32_t h[256];
  h[threadIdx.x] = 0;
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * SORT_TILE;
#pragma unroll
  for (int r = 0; r < SORT_ITEMS; ++r) {
    int64_t i = base + r * 256 + threadIdx.x;
    if (i < n) atomicAdd(&h[(keys[i] >> shift) & 0xFF], 1u);
  }
  __syncthreads();
  hist[(int64_t)threadIdx.x * ntiles + blockIdx.x] = h[threadIdx.x];
}

__global__ __launch_bounds__(256) void sort_scatter_kernel(const uint64_t* keys, const uint32_t* vals, int64_t n,
                                                           int shift, const uint64_t* offs, int64_t ntiles,
                                                           uint64_t* out_keys, uint32_t* out_vals) {
  __shared__ uint32_t running[256];     // keys of each digit already placed by earlier rounds
  __shared__ uint32_t wave_cnt[4][256];  // per-wave digit counts of the current round
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  running[tid] = 0;
#pragma unroll
  for (int w = 0; w < 4; ++w) wave_cnt[w][tid] = 0;
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * SORT_TILE;
  const uint64_t my_digit_base = offs[(int64_t)tid * ntiles + blockIdx.x];  // thread t owns digit t
  __shared__ uint64_t digit_base[256];
  digit_base[tid] = my_digit_base;
  __syncthreads();
  for (int r = 0; r < SORT_ITEMS; ++r) {
    const int64_t i = base + r * 256 + tid;
    const bool active = i < n;
    uint64_t key = active ? keys[i] : 0;
    uint32_t val = active ? vals[i] : 0;
    const uint32_t digit = (uint32_t)(key >> shift) & 0xFF;
    // lanes of this wave holding the same digit
    uint64_t m = __ballot(active);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      uint64_t bal = __ballot((digit >> b) & 1);
      m &= ((digit >> b) & 1) ? bal : ~bal;
    }
    const uint32_t rank = __popcll(m & ((1ULL << lane) - 1));
    if (active && rank == 0) wave_cnt[wave][digit] = __popcll(m);
    __syncthreads();
    if (active) {
      uint32_t before = running[digit];
      for (int w = 0; w < wave; ++w) before += wave_cnt[w][digit];
      uint64_t pos = digit_base[digit] + before + rank;
      out_keys[pos] = key;
      out_vals[pos] = val;
    }
    __syncthreads();
    running[tid] += wave_cnt[0][tid] + wave_cnt[1][tid] + wave_cnt[2][tid] + wave_cnt[3][tid];
#pragma unroll
    for (int w = 0; w < 4; ++w) wave_cnt[w][tid] = 0;
    __syncthreads();
  }
}

}  // namespace

extern "C" {

int32_t dbhip_sort_perm(const dbhip_col* keys, const uint8_t* desc_host, const uint8_t* nulls_first_host,
                        int32_t nkeys, int64_t n, int64_t limit, uint32_t* out_perm, void* stream) {
  DBHIP_REQUIRE(keys && nkeys >= 1 && nkeys <= 8, "dbhip_sort_perm: 1..8 sort keys");
  DBHIP_REQUIRE(n >= 0 && n < 0xFFFFFFFFLL, "dbhip_sort_perm: row count out of range");
  if (n == 0) return DBHIP_OK;
  DBHIP_REQUIRE(out_perm, "dbhip_sort_perm: NULL out");
  for (int k = 0; k < nkeys; ++k) {
    int t = keys[k].type;
    bool ok = (t >= DBHIP_T_BOOL && t <= DBHIP_T_STRING) && !keys[k].is_scalar;
    if (!ok) {
      set_error("dbhip_sort_perm: key %d has unsupported type %d (fixed-width and short-string columns only)", k, t);
      return DBHIP_ERR_UNSUPPORTED;
    }
  }
  hipStream_t s = resolve_stream(stream);
  const int64_t ntiles = ceil_div(n, SORT_TILE);
  const int64_t nh = 256 * ntiles;
  // scratch: 2 key buffers, 2 perm buffers, hist, offsets, scan block sums
  size_t bytes = (size_t)n * 8 * 2 + (size_t)n * 4 * 2 + (size_t)nh * 4 + (size_t)nh * 8 + (size_t)(nh / SCAN_TILE + 2) * 8 + 1024;
  uint8_t* ws = (uint8_t*)scratch(bytes + 64, 7);
  if (!ws) return DBHIP_ERR_HIP;
  uint32_t* long_flag = (uint32_t*)(ws + bytes);
  uint64_t* kb[2] = {(uint64_t*)ws, (uint64_t*)ws + n};
  uint64_t* offs = kb[1] + n;
  uint64_t* blk = offs + nh;
  uint32_t* hist = (uint32_t*)(blk + nh / SCAN_TILE + 2);
  uint32_t* pb[2] = {hist + nh, hist + nh + n};
  int cur = 0;
  const int grid = grid_for(n, 256);
  bool any_string = false;
  for (int k = 0; k < nkeys; ++k) any_string |= keys[k].type == DBHIP_T_STRING;
  if (any_string) {
    DBHIP_CHECK(hipMemsetAsync(long_flag, 0, 4, s));
    for (int k = 0; k < nkeys; ++k)
      if (keys[k].type == DBHIP_T_STRING)
        hipLaunchKernelGGL(sort_check_inline_kernel, dim3(grid), dim3(256), 0, s, (const uint32_t*)keys[k].data, n, long_flag);
    uint32_t f = 0;
    DBHIP_CHECK(hipMemcpyAsync(&f, long_flag, 4, hipMemcpyDeviceToHost, s));
    DBHIP_CHECK(hipStreamSynchronize(s));
    if (f) {
      set_error("dbhip_sort_perm: a string sort key is longer than 12 bytes; keep the CPU operator for this block");
      return DBHIP_ERR_UNSUPPORTED;
    }
  }
  hipLaunchKernelGGL(sort_iota_kernel, dim3(grid), dim3(256), 0, s, pb[cur], n);

  auto radix_passes = [&](int nbytes) -> int32_t {
    for (int b = 0; b < nbytes; ++b) {
      hipLaunchKernelGGL(sort_hist_kernel, dim3((unsigned)ntiles), dim3(256), 0, s, kb[cur], n, 8 * b, hist, ntiles);
      int32_t rc = dbscan::exclusive_scan_u32(hist, nh, blk, offs, s);
      if (rc) return rc;
      hipLaunchKernelGGL(sort_scatter_kernel, dim3((unsigned)ntiles), dim3(256), 0, s, kb[cur], pb[cur], n, 8 * b, offs,
                         ntiles, kb[cur ^ 1], pb[cur ^ 1]);
      cur ^= 1;
    }
    DBHIP_LAUNCH_CHECK();
    return DBHIP_OK;
  };

  for (int k = nkeys - 1; k >= 0; --k) {
    SortCol c{keys[k].data, keys[k].validity, keys[k].validity_offset, keys[k].type,
              desc_host ? desc_host[k] : 0, nulls_first_host ? nulls_first_host[k] : 0};
    const int parts = (c.type == DBHIP_T_DEC128 || c.type == DBHIP_T_STRING) ? 2 : 1;
    for (int part = 0; part < parts; ++part) {
      // the permutation is shared by both buffers of a pass: encode reads pb[cur], writes kb[cur]
      hipLaunchKernelGGL(sort_encode_kernel, dim3(grid), dim3(256), 0, s, c, pb[cur], n, part, kb[cur]);
      int32_t rc = radix_passes(sort_key_bytes(c.type));
      if (rc) return rc;
    }
    if (c.validity) {
      hipLaunchKernelGGL(sort_encode_kernel, dim3(grid), dim3(256), 0, s, c, pb[cur], n, 2, kb[cur]);
      int32_t rc = radix_passes(1);
      if (rc) return rc;
    }
  }
  int64_t m = (limit > 0 && limit < n) ? limit : n;
  DBHIP_CHECK(hipMemcpyAsync(out_perm, pb[cur], (size_t)m * 4, hipMemcpyDeviceToDevice, s));
  DBHIP_CHECK(hipStreamSynchronize(s));  // scratch is reused by the next call
  return DBHIP_OK;
}

}  // extern "C"

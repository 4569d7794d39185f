// k_sort.hip — sort permutation.
//
// Reference: DataBlock::sort_with_type -> SortCompare (kernels/sort.rs:91-113,
// kernels/sort_compare.rs:33-283): a u32 permutation refined column by column with
// sort_unstable_by, asc/desc and nulls-first/last per column, optional row limit
// (LimitType::LimitRows, :197-209). Only the KEY SEQUENCE is defined (ties are unordered there).
// Device algorithm: LSD radix sort, stable, 8 bits per pass (round 5: onesweep passes for 64-bit images of 2^20 rows and more,
// see sort_onesweep_kernel), over order-preserving u64 encodings
// (the device analogue of the reference's fixed-width row encoding, sorts/core/row_convert/fixed.rs):
// keys are processed from the last to the first; for each key an encode kernel reads the column
// THROUGH the current permutation, then one pass per key byte:
//   histogram   per-tile digit counts in LDS                -> hist[digit][tile]
//   scan        device-wide exclusive scan (dev_scan.h)     -> base offset of (digit, tile)
//   scatter     wave-level digit matching with 8 ballots gives each key its stable rank inside
//               the tile; (key, row id) pairs move to their final position of this pass
// Nullable keys get one extra (most significant) pass on the null flag. Ties end up in ascending
// row-id order (stable), which is one of the orders the reference may produce.
#include <string.h>
#include "dev_common.h"
#include "dev_scan.h"
#include "runtime.h"

using namespace dbhip;

namespace {

constexpr int SORT_ITEMS = 16;                // keys per thread
constexpr int SORT_TILE = 256 * SORT_ITEMS;   // keys per block (4 waves x 1024 contiguous keys)

struct SortCol {
  const void* data;
  const uint8_t* validity;
  int64_t voff;
  int type;
  int desc;
  int nulls_first;
  // String columns that hold a value longer than 12 bytes: nparts = 1 + ceil(max length / 8) images per value (part 0 = the
  // length, part q = bytes [8 (nparts - 1 - q), +8) big-endian, zero padded past the end) and the column's data buffers;
  // 0 = all values inline (the two-part image of the 16-byte view)
  int nparts;
  const void* const* buffers;
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
      if (c.nparts) {   // strings of any length: memcmp order of the bytes, then the length (variable.rs: a proper prefix sorts first)
        const uint32_t* v = (const uint32_t*)c.data + 4 * (uint64_t)row;
        const uint32_t len = v[0];
        if (part == 0) return len;
        const uint32_t at = 8u * (uint32_t)(c.nparts - 1 - part);
        if (at >= len) return 0;
        const uint8_t* p = len <= 12 ? (const uint8_t*)(v + 1) : (const uint8_t*)c.buffers[v[2]] + v[3];
        const uint32_t take = len - at < 8 ? len - at : 8;
        uint64_t img = 0;
        for (uint32_t b = 0; b < take; ++b) img |= (uint64_t)p[at + b] << (8 * (7 - b));
        return img;
      }
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

// the longest value of a string column (values of more than 12 bytes need the data buffers and more key images)
__global__ __launch_bounds__(256) void sort_max_len_kernel(const uint32_t* views, int64_t n, uint32_t* out) {
  uint32_t m = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) m = views[4 * i] > m ? views[4 * i] : m;
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) { const uint32_t o = __shfl_xor(m, off, 64); m = o > m ? o : m; }
  if (lane_id() == 0 && m) atomicMax(out, m);
}

__global__ __launch_bounds__(256) void sort_iota_kernel(uint32_t* perm, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    perm[i] = (uint32_t)i;
}

// enc[j] = image of col[perm[j]]; mode 0: value part 0, 1: value part 1 (dec128 high), 2: null flag
// Also reduces OR / AND over all images into span[0] / span[1]: bytes where both agree are constant
// over the column and their radix pass is skipped.
// K = uint32_t for key columns of at most 4 bytes (and the null-flag pass): the radix passes then move 8 instead of
// 12 bytes per key and pass.
template <typename K>
__global__ __launch_bounds__(256) void sort_encode_kernel(SortCol c, const uint32_t* perm, int64_t n, int mode,
                                                          K* enc, unsigned long long* span) {
  uint64_t acc_or = 0, acc_and = ~0ULL;
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
    if (sizeof(K) == 4) e &= 0xFFFFFFFFull;
    enc[j] = (K)e;
    acc_or |= e;
    acc_and &= e;
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    acc_or |= __shfl_xor(acc_or, off, 64);
    acc_and &= __shfl_xor(acc_and, off, 64);
  }
  if (lane_id() == 0) {
    atomicOr(&span[0], (unsigned long long)acc_or);
    atomicAnd(&span[1], (unsigned long long)acc_and);
  }
}

template <typename K, int NT = 256>
__global__ __launch_bounds__(NT) void sort_hist_kernel(const K* keys, int64_t n, int shift, uint32_t* hist,
                                                       int64_t ntiles) {
  __shared__ uint32_t h[256];
  if (threadIdx.x < 256) h[threadIdx.x] = 0;
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * (NT * SORT_ITEMS);
#pragma unroll
  for (int r = 0; r < SORT_ITEMS; ++r) {
    int64_t i = base + r * NT + threadIdx.x;
    if (i < n) atomicAdd(&h[(keys[i] >> shift) & 0xFF], 1u);
  }
  __syncthreads();
  if (threadIdx.x < 256) hist[(int64_t)threadIdx.x * ntiles + blockIdx.x] = h[threadIdx.x];
}

// One tile = 4096 keys; wave w owns the contiguous keys [w*1024, (w+1)*1024) in 16 rounds of 64.
//   phase 1  per round: lanes holding the same digit find each other with 8 ballots; the key's rank among
//            its wave's keys of that digit = (count so far in wcount[w][digit]) + (rank inside the round)
//   phase 2  thread d: tile offset of digit d (block scan of the totals) and each wave's start inside it
//   phase 3  keys and row ids are written to their digit-sorted position IN LDS
//   phase 4  the tile leaves in digit order: lanes write consecutive addresses per digit run
// Stable: (wave, round, lane) order is index order.
template <typename K, int NT = 256>
__global__ __launch_bounds__(NT) void sort_scatter_kernel(const K* keys, const uint32_t* vals, int64_t n,
                                                          int shift, const uint64_t* offs, int64_t ntiles,
                                                          K* out_keys, uint32_t* out_vals) {
  constexpr int NW = NT / 64, TILE = NT * SORT_ITEMS;   // NT = 512: 8192-key tiles (twice the run length per digit: 256-byte key runs)
  __shared__ K lkeys[TILE];
  __shared__ uint32_t lvals[TILE];
  __shared__ uint32_t wcount[NW][256];  // phase 1: keys of (wave, digit); phase 2 on: first LDS slot of (wave, digit)
  __shared__ uint32_t tile_off[256];
  __shared__ uint32_t wave_tot[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid < 256) {
#pragma unroll
    for (int w = 0; w < NW; ++w) wcount[w][tid] = 0;
  }
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * TILE + (int64_t)wave * (64 * SORT_ITEMS);
  K key[SORT_ITEMS];
  uint32_t val[SORT_ITEMS];
  uint32_t lrank[SORT_ITEMS];
#pragma unroll
  for (int r = 0; r < SORT_ITEMS; ++r) {
    const int64_t i = base + r * 64 + lane;
    key[r] = i < n ? keys[i] : (K)~0ULL;
    val[r] = i < n ? vals[i] : 0;
  }
  volatile uint32_t* wc = wcount[wave];
#pragma unroll
  for (int r = 0; r < SORT_ITEMS; ++r) {
    const int64_t i = base + r * 64 + lane;
    const bool active = i < n;
    const uint32_t digit = (uint32_t)(key[r] >> shift) & 0xFF;
    uint64_t m = __ballot(active);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const uint64_t bal = __ballot((digit >> b) & 1);
      m &= ((digit >> b) & 1) ? bal : ~bal;
    }
    const uint32_t rank = __popcll(m & ((1ULL << lane) - 1));
    const uint32_t prev = wc[digit];
    __builtin_amdgcn_wave_barrier();
    if (active && rank == 0) wc[digit] = prev + __popcll(m);
    __builtin_amdgcn_wave_barrier();
    lrank[r] = prev + rank;
  }
  __syncthreads();
  {
    // threads 0..255 (waves 0..3): one digit each — its count over the tile's waves, the exclusive scan over the digits
    uint32_t cw[NW];
    uint32_t tot = 0, incl = 0;
    if (tid < 256) {
#pragma unroll
      for (int w = 0; w < NW; ++w) { cw[w] = wcount[w][tid]; tot += cw[w]; }
      incl = tot;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(incl, d, 64);
        if (lane >= d) incl += o;
      }
      if (lane == 63) wave_tot[wave] = incl;
    }
    __syncthreads();
    if (tid < 256) {
      uint32_t wb = 0;
      for (int w = 0; w < wave; ++w) wb += wave_tot[w];
      uint32_t off = wb + incl - tot;
      tile_off[tid] = off;
#pragma unroll
      for (int w = 0; w < NW; ++w) { wcount[w][tid] = off; off += cw[w]; }
    }
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < SORT_ITEMS; ++r) {
    const int64_t i = base + r * 64 + lane;
    if (i < n) {
      const uint32_t digit = (uint32_t)(key[r] >> shift) & 0xFF;
      const uint32_t p = wcount[wave][digit] + lrank[r];
      lkeys[p] = key[r];
      lvals[p] = val[r];
    }
  }
  __syncthreads();
  const int64_t tile_base = (int64_t)blockIdx.x * TILE;
  const int tile_n = (int)((n - tile_base) < TILE ? (n - tile_base) : TILE);
#pragma unroll 4
  for (int j = tid; j < tile_n; j += NT) {
    const K k = lkeys[j];
    const uint32_t digit = (uint32_t)(k >> shift) & 0xFF;
    const uint64_t pos = offs[(int64_t)digit * ntiles + blockIdx.x] + (uint32_t)(j - tile_off[digit]);
    if (out_keys) out_keys[pos] = k;   // (the last pass over a key image: only the permutation is still needed)
    out_vals[pos] = lvals[j];
  }
}

// ---- onesweep (round 5) -------------------------------------------------------------------------------------------------------
// The passes over ONE key image share a single up-front histogram (all digits of the image: the keys are read once instead of once
// per pass), and a pass finds where its tiles go without a histogram matrix and a device-wide scan: tile t publishes its digit
// counts, then looks back over the tiles before it until it meets one that already knows its inclusive prefix (decoupled
// look-back). Per key and pass 12 B in + 12 B out instead of 8 + 12 + 12, and three launches per pass become one.
//   status[pass][tile][digit] (u64): bits 63..62 = 0 nothing yet / 1 the tile's own count / 2 the inclusive prefix up to this tile
//   tiles take their number from a ticket counter, so a tile only ever waits for tiles that are already running
//   the wait is bounded: a look-back that does not get an answer in ~2^24 polls raises ctl[1], every tile leaves, and the call
//   fails with an error instead of hanging the device
constexpr uint64_t OS_FLAG_AGG = 1ULL << 62, OS_FLAG_PREFIX = 2ULL << 62, OS_COUNT_MASK = (1ULL << 62) - 1;

template <typename K, int NB>
__global__ __launch_bounds__(256) void sort_onesweep_hist_kernel(const K* __restrict__ keys, int64_t n, uint32_t vary_bytes, unsigned long long* __restrict__ ghist) {
  __shared__ uint32_t h[NB][256];
  for (int i = threadIdx.x; i < NB * 256; i += 256) (&h[0][0])[i] = 0;
  __syncthreads();
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const K k = keys[i];
#pragma unroll
    for (int b = 0; b < NB; ++b)
      if ((vary_bytes >> b) & 1u) atomicAdd(&h[b][(uint32_t)(k >> (8 * b)) & 0xFF], 1u);   // (uniform: bytes that are the same in every key have no pass)
  }
  __syncthreads();
  for (int i = threadIdx.x; i < NB * 256; i += 256) {
    const uint32_t c = (&h[0][0])[i];
    if (c) atomicAdd(&ghist[i], (unsigned long long)c);
  }
}

template <typename K, int NT>
__global__ __launch_bounds__(NT) void sort_onesweep_kernel(const K* __restrict__ keys, const uint32_t* __restrict__ vals, int64_t n, int shift,
                                                           const unsigned long long* __restrict__ ghist, unsigned long long* __restrict__ status,
                                                           uint32_t* __restrict__ ctl, K* __restrict__ out_keys, uint32_t* __restrict__ out_vals) {
  constexpr int NW = NT / 64, TILE = NT * SORT_ITEMS;
  __shared__ K lkeys[TILE];
  __shared__ uint32_t lvals[TILE];
  __shared__ uint32_t wcount[NW][256];
  __shared__ uint32_t tile_off[256];
  __shared__ unsigned long long gpos[256];   // where the tile's keys of a digit go in the output
  __shared__ unsigned long long wave_tot64[4];
  __shared__ uint32_t wave_tot[4];
  __shared__ uint32_t tile_id;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) tile_id = atomicAdd(&ctl[0], 1u);
  if (tid < 256) {
#pragma unroll
    for (int w = 0; w < NW; ++w) wcount[w][tid] = 0;
  }
  __syncthreads();
  const uint32_t tile = tile_id;
  const int64_t tile_base = (int64_t)tile * TILE;
  const int64_t base = tile_base + (int64_t)wave * (64 * SORT_ITEMS);
  K key[SORT_ITEMS];
  uint32_t val[SORT_ITEMS];
  uint32_t lrank[SORT_ITEMS];
#pragma unroll
  for (int r = 0; r < SORT_ITEMS; ++r) {
    const int64_t i = base + r * 64 + lane;
    key[r] = i < n ? keys[i] : (K)~0ULL;
    val[r] = i < n ? vals[i] : 0;
  }
  volatile uint32_t* wc = wcount[wave];
#pragma unroll
  for (int r = 0; r < SORT_ITEMS; ++r) {
    const int64_t i = base + r * 64 + lane;
    const bool active = i < n;
    const uint32_t digit = (uint32_t)(key[r] >> shift) & 0xFF;
    uint64_t m = __ballot(active);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const uint64_t bal = __ballot((digit >> b) & 1);
      m &= ((digit >> b) & 1) ? bal : ~bal;
    }
    const uint32_t rank = __popcll(m & ((1ULL << lane) - 1));
    const uint32_t prev = wc[digit];
    __builtin_amdgcn_wave_barrier();
    if (active && rank == 0) wc[digit] = prev + __popcll(m);
    __builtin_amdgcn_wave_barrier();
    lrank[r] = prev + rank;
  }
  __syncthreads();
  {
    uint32_t cw[NW];
    uint32_t tot = 0, incl = 0;
    unsigned long long gh = 0, gincl = 0;
    if (tid < 256) {
#pragma unroll
      for (int w = 0; w < NW; ++w) { cw[w] = wcount[w][tid]; tot += cw[w]; }
      // this tile's count is public before anything else: the tiles behind it may already be looking
      unsigned long long* st = status + (size_t)tile * 256 + tid;
      if (tile == 0) __hip_atomic_store(st, OS_FLAG_PREFIX | (unsigned long long)tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else __hip_atomic_store(st, OS_FLAG_AGG | (unsigned long long)tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      incl = tot;
      gh = ghist[tid];
      gincl = gh;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(incl, d, 64);
        const unsigned long long go = __shfl_up(gincl, d, 64);
        if (lane >= d) { incl += o; gincl += go; }
      }
      if (lane == 63) { wave_tot[wave] = incl; wave_tot64[wave] = gincl; }
    }
    __syncthreads();
    if (tid < 256) {
      uint32_t wb = 0;
      unsigned long long gwb = 0;
      for (int w = 0; w < wave; ++w) { wb += wave_tot[w]; gwb += wave_tot64[w]; }
      uint32_t off = wb + incl - tot;
      tile_off[tid] = off;
#pragma unroll
      for (int w = 0; w < NW; ++w) { wcount[w][tid] = off; off += cw[w]; }
      // look back: keys of this digit in the tiles before this one
      unsigned long long before = 0;
      bool stalled = false;
      for (int64_t t = (int64_t)tile - 1; t >= 0; --t) {
        const unsigned long long* sp = status + (size_t)t * 256 + tid;
        unsigned long long v = __hip_atomic_load(sp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        uint32_t polls = 0;
        while ((v >> 62) == 0) {
          if ((++polls & 0x3FFu) == 0 && (polls >= (1u << 24) || __hip_atomic_load(&ctl[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) { stalled = true; break; }
          __builtin_amdgcn_s_sleep(1);
          v = __hip_atomic_load(sp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (stalled) break;
        before += v & OS_COUNT_MASK;
        if ((v >> 62) == 2) break;
      }
      if (stalled) __hip_atomic_store(&ctl[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else if (tile != 0) __hip_atomic_store(status + (size_t)tile * 256 + tid, OS_FLAG_PREFIX | (before + tot), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      gpos[tid] = (gwb + gincl - gh) + before;   // keys of smaller digits anywhere + keys of this digit in earlier tiles
    }
  }
  __syncthreads();
  if (__hip_atomic_load(&ctl[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;   // (a stalled look-back: the host reports it; nothing is written)
#pragma unroll
  for (int r = 0; r < SORT_ITEMS; ++r) {
    const int64_t i = base + r * 64 + lane;
    if (i < n) {
      const uint32_t digit = (uint32_t)(key[r] >> shift) & 0xFF;
      const uint32_t p = wcount[wave][digit] + lrank[r];
      lkeys[p] = key[r];
      lvals[p] = val[r];
    }
  }
  __syncthreads();
  const int tile_n = (int)((n - tile_base) < TILE ? (n - tile_base) : TILE);
#pragma unroll 4
  for (int j = tid; j < tile_n; j += NT) {
    const K k = lkeys[j];
    const uint32_t digit = (uint32_t)(k >> shift) & 0xFF;
    const uint64_t pos = gpos[digit] + (uint32_t)(j - tile_off[digit]);
    if (out_keys) out_keys[pos] = k;
    out_vals[pos] = lvals[j];
  }
}

// DataBlock::scatter for one value buffer: the radix scatter above with the destination index (< 256) as the digit and the
// column's elements as the payload — rows keep their order inside a destination (stable). The index is not written back.
template <typename V>
__global__ __launch_bounds__(256) void scatter_rows_kernel(const uint32_t* index, const V* src, int64_t n, const uint64_t* offs, int64_t ntiles,
                                                           V* out) {
  __shared__ V lvals[SORT_TILE];
  __shared__ uint8_t ldig[SORT_TILE];
  __shared__ uint32_t wcount[4][256];
  __shared__ uint32_t tile_off[256];
  __shared__ uint32_t wave_tot[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#pragma unroll
  for (int w = 0; w < 4; ++w) wcount[w][tid] = 0;
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * SORT_TILE + (int64_t)wave * (64 * SORT_ITEMS);
  uint32_t dig[SORT_ITEMS];
  V val[SORT_ITEMS];
  uint32_t lrank[SORT_ITEMS];
#pragma unroll
  for (int r = 0; r < SORT_ITEMS; ++r) {
    const int64_t i = base + r * 64 + lane;
    dig[r] = i < n ? (index[i] & 0xFF) : 0xFF;
    val[r] = i < n ? src[i] : V{};
  }
  volatile uint32_t* wc = wcount[wave];
#pragma unroll
  for (int r = 0; r < SORT_ITEMS; ++r) {
    const int64_t i = base + r * 64 + lane;
    const bool active = i < n;
    const uint32_t digit = dig[r];
    uint64_t m = __ballot(active);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const uint64_t bal = __ballot((digit >> b) & 1);
      m &= ((digit >> b) & 1) ? bal : ~bal;
    }
    const uint32_t rank = __popcll(m & ((1ULL << lane) - 1));
    const uint32_t prev = wc[digit];
    __builtin_amdgcn_wave_barrier();
    if (active && rank == 0) wc[digit] = prev + __popcll(m);
    __builtin_amdgcn_wave_barrier();
    lrank[r] = prev + rank;
  }
  __syncthreads();
  {
    const uint32_t c0 = wcount[0][tid], c1 = wcount[1][tid], c2 = wcount[2][tid], c3 = wcount[3][tid];
    const uint32_t tot = c0 + c1 + c2 + c3;
    uint32_t incl = tot;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t o = __shfl_up(incl, d, 64);
      if (lane >= d) incl += o;
    }
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    uint32_t wb = 0;
    for (int w = 0; w < wave; ++w) wb += wave_tot[w];
    const uint32_t off = wb + incl - tot;
    tile_off[tid] = off;
    wcount[0][tid] = off;
    wcount[1][tid] = off + c0;
    wcount[2][tid] = off + c0 + c1;
    wcount[3][tid] = off + c0 + c1 + c2;
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < SORT_ITEMS; ++r) {
    const int64_t i = base + r * 64 + lane;
    if (i < n) {
      const uint32_t p = wcount[wave][dig[r]] + lrank[r];
      lvals[p] = val[r];
      ldig[p] = (uint8_t)dig[r];
    }
  }
  __syncthreads();
  const int64_t tile_base = (int64_t)blockIdx.x * SORT_TILE;
  const int tile_n = (int)((n - tile_base) < SORT_TILE ? (n - tile_base) : SORT_TILE);
#pragma unroll 4
  for (int j = tid; j < tile_n; j += 256) {
    const uint32_t digit = ldig[j];
    out[offs[(int64_t)digit * ntiles + blockIdx.x] + (uint32_t)(j - tile_off[digit])] = lvals[j];
  }
}

// ---- LIMIT: radix select on the most significant sort key --------------------------------------
// count of images whose bits above `shift+8` equal those of `prefix`, per next byte
__global__ __launch_bounds__(256) void sort_select_hist_kernel(const uint64_t* enc, int64_t n, uint64_t prefix, int shift,
                                                               uint32_t* hist) {
  __shared__ uint32_t h[256];
  h[threadIdx.x] = 0;
  __syncthreads();
  const uint64_t himask = shift >= 56 ? 0ULL : (~0ULL << (shift + 8));
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const uint64_t e = enc[i];
    if ((e & himask) == (prefix & himask)) atomicAdd(&h[(e >> shift) & 0xFF], 1u);
  }
  __syncthreads();
  if (h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], h[threadIdx.x]);
}

// cand[0..count) = ascending row ids with enc <= threshold (wave-aggregated append keeps it cheap;
// the order of the list is irrelevant: the sort that follows is on keys, ties by row id need a stable
// base order, so the list is produced in ascending order by a scan instead)
__global__ __launch_bounds__(256) void sort_select_flag_kernel(const uint64_t* enc, int64_t n, uint64_t threshold,
                                                               uint32_t* cnt) {
  // per-tile (1024 rows) count of qualifying rows
  const int64_t base = (int64_t)blockIdx.x * 1024;
  uint32_t c = 0;
  for (int k = threadIdx.x; k < 1024; k += 256) {
    const int64_t i = base + k;
    if (i < n && enc[i] <= threshold) ++c;
  }
  c = (uint32_t)wave_sum_u64(c);
  __shared__ uint32_t part[4];
  if (lane_id() == 0) part[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) cnt[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}

__global__ __launch_bounds__(256) void sort_select_emit_kernel(const uint64_t* enc, int64_t n, uint64_t threshold,
                                                               const uint64_t* tile_off, uint32_t* cand) {
  const int64_t base = (int64_t)blockIdx.x * 1024;
  __shared__ uint32_t run;
  __shared__ uint32_t wt[4];
  if (threadIdx.x == 0) run = 0;
  __syncthreads();
  const uint64_t out0 = tile_off[blockIdx.x];
  for (int chunk = 0; chunk < 1024; chunk += 256) {
    const int64_t i = base + chunk + threadIdx.x;
    const bool q = i < n && enc[i] <= threshold;
    const uint64_t m = __ballot(q);
    if (lane_id() == 0) wt[threadIdx.x >> 6] = __popcll(m);
    __syncthreads();
    uint32_t wb = run;
    for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) wb += wt[w];
    if (q) cand[out0 + wb + __popcll(m & ((1ULL << lane_id()) - 1))] = (uint32_t)i;
    __syncthreads();
    if (threadIdx.x == 0) run += wt[0] + wt[1] + wt[2] + wt[3];
    __syncthreads();
  }
}


// ---- range partition of rows by sorted bounds (the distributed sort's scatter step, SURVEY §8e "sort") ----
// Reference: a sorted stream is cut at every bound — rows <= bound[i] (in sort order) belong to partition i, rows after the last
// bound to partition nbounds (sort_spill.rs:1008-1040 block_split_off_position / partition_point "first element that is
// greater than bound"; BoundBlockStream :740-1005) — and partition i travels to node i % n (sort_exchange_injector.rs
// SortBoundScatter / bound_scatter). On the device the rows need not be sorted first: partition(row) = number of bounds that
// sort strictly before the row, a binary search per row over the bounds' order-preserving images (the same images the
// radix sort uses, most significant first), which are staged in LDS.
struct BoundKeys {
  SortCol row[8];
  SortCol bnd[8];
  int nkeys;
  int nimg;            // images per row: per key an optional null flag, then the value parts from the most significant down
  uint8_t has_flag[8];
  uint8_t parts[8];    // <= 255 only for short keys; long strings carry their count in row[k].nparts
};

__device__ __forceinline__ int bound_key_parts(const BoundKeys& bk, int k) { return bk.row[k].nparts ? bk.row[k].nparts : bk.parts[k]; }

// image q of key k: q = 0 is the null flag when the key has one, then the value parts
__device__ __forceinline__ uint64_t bound_image(const SortCol& c, uint32_t row, bool has_flag, int parts, int q) {
  const bool valid = !c.validity || bit_get(c.validity, c.voff + row);
  if (has_flag) {
    if (q == 0) return (uint64_t)((valid ? 1 : 0) ^ (c.nulls_first ? 0 : 1));
    --q;
  }
  if (!valid) return 0;
  const uint64_t e = sort_encode(c, row, parts - 1 - q);
  return c.desc ? ~e : e;
}

__global__ __launch_bounds__(256) void sort_bound_encode_kernel(BoundKeys bk, int nbounds, uint64_t* img) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= nbounds) return;
  int at = 0;
  for (int k = 0; k < bk.nkeys; ++k) {
    const int parts = bound_key_parts(bk, k), nq = parts + (bk.has_flag[k] ? 1 : 0);
    for (int q = 0; q < nq; ++q) img[(size_t)j * bk.nimg + at++] = bound_image(bk.bnd[k], (uint32_t)j, bk.has_flag[k], parts, q);
  }
}

// does bound j sort strictly before the row? `first` = the row's first image (computed once per row)
__device__ __forceinline__ bool bound_before_row(const BoundKeys& bk, const uint64_t* img, int j, uint32_t row, uint64_t first) {
  const uint64_t* b = img + (size_t)j * bk.nimg;
  if (b[0] != first) return b[0] < first;
  int at = 0;
  for (int k = 0; k < bk.nkeys; ++k) {
    const int parts = bound_key_parts(bk, k), nq = parts + (bk.has_flag[k] ? 1 : 0);
    for (int q = 0; q < nq; ++q, ++at) {
      if (at == 0) continue;
      const uint64_t r = bound_image(bk.row[k], row, bk.has_flag[k], parts, q);
      if (b[at] != r) return b[at] < r;
    }
  }
  return false;
}

constexpr int BOUND_HIST_MAX = 2048;

template <bool IN_LDS>
__global__ __launch_bounds__(256) void sort_bound_partition_kernel(BoundKeys bk, int64_t n, int nbounds, const uint64_t* img_global,
                                                                   uint32_t* out_part, unsigned long long* counts) {
  extern __shared__ uint64_t bound_sh[];
  const int nhist = nbounds + 1 <= BOUND_HIST_MAX ? nbounds + 1 : 0;
  uint32_t* hist = (uint32_t*)bound_sh;
  uint64_t* img_lds = bound_sh + (nhist + 1) / 2;
  for (int i = threadIdx.x; i < nhist; i += blockDim.x) hist[i] = 0;
  if (IN_LDS)
    for (int i = threadIdx.x; i < nbounds * bk.nimg; i += blockDim.x) img_lds[i] = img_global[i];
  __syncthreads();
  const uint64_t* img = IN_LDS ? img_lds : img_global;
  const int parts0 = bound_key_parts(bk, 0);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const uint64_t first = bound_image(bk.row[0], (uint32_t)i, bk.has_flag[0], parts0, 0);
    int lo = 0, hi = nbounds;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (bound_before_row(bk, img, mid, (uint32_t)i, first)) lo = mid + 1; else hi = mid;
    }
    out_part[i] = (uint32_t)lo;
    if (nhist) atomicAdd(&hist[lo], 1u); else atomicAdd(&counts[lo], 1ULL);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < nhist; i += blockDim.x)
    if (hist[i]) atomicAdd(&counts[i], (unsigned long long)hist[i]);
}

// The common shape — ONE fixed-width key without NULLs (one image per row) and at most 63 bounds: the bounds sit in registers
// (uniform loads), a row's range is the branch-free count of bounds below its image, four rows per thread are in flight, and the
// rows per range are counted with one ballot per range (lane d accumulates range d) instead of LDS atomics.
constexpr int BOUND_SMALL_MAX = 63;
template <int NB>   // NB = bounds rounded up to 8, 16, 32 or 64 (padded with ~0: never below a row)
__global__ __launch_bounds__(256) void sort_bound_partition_small_kernel(BoundKeys bk, int64_t n, int nbounds, const uint64_t* img,
                                                                         uint32_t* out_part, unsigned long long* counts) {
  uint64_t b[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j) b[j] = j < nbounds ? img[j] : ~0ULL;
  const int parts0 = bound_key_parts(bk, 0);
  const int lane = lane_id();
  unsigned long long mine = 0;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t base = (int64_t)blockIdx.x * blockDim.x + threadIdx.x - lane; base < n; base += stride) {   // whole waves iterate together
    const int64_t i = base + lane;
    const bool live = i < n;
    const uint64_t x = live ? bound_image(bk.row[0], (uint32_t)i, false, parts0, 0) : 0;
    uint32_t p = 0;
#pragma unroll
    for (int j = 0; j < NB; ++j) p += b[j] < x ? 1u : 0u;
    if (live) out_part[i] = p;
    for (int d = 0; d <= nbounds; ++d) {
      const uint64_t m = __ballot(live && p == (uint32_t)d);
      if (lane == d) mine += (unsigned long long)__popcll(m);
    }
  }
  if (lane <= nbounds && mine) atomicAdd(&counts[lane], mine);
}

}  // namespace

extern "C" {

int32_t dbhip_sort_bound_partition(const dbhip_col* keys, const dbhip_col* bounds, const uint8_t* desc_host, const uint8_t* nulls_first_host,
                                   int32_t nkeys, int64_t n, int64_t nbounds, uint32_t* out_part, uint64_t* out_counts, void* stream) {
  DBHIP_REQUIRE(keys && nkeys >= 1 && nkeys <= 8, "dbhip_sort_bound_partition: 1..8 sort keys");
  DBHIP_REQUIRE(n >= 0 && n < 0xFFFFFFFFLL && nbounds >= 0 && nbounds < (1 << 24), "dbhip_sort_bound_partition: row / bound count out of range");
  DBHIP_REQUIRE(out_counts && (nbounds == 0 || bounds), "dbhip_sort_bound_partition: NULL buffer");
  hipStream_t s = resolve_stream(stream);
  DBHIP_CHECK(hipMemsetAsync(out_counts, 0, (size_t)(nbounds + 1) * 8, s));
  if (n == 0) return DBHIP_OK;
  DBHIP_REQUIRE(out_part, "dbhip_sort_bound_partition: NULL out");
  BoundKeys bk;
  memset(&bk, 0, sizeof(bk));
  bk.nkeys = nkeys;
  const int grid = grid_for(n, 256);
  for (int k = 0; k < nkeys; ++k) {
    const int t = keys[k].type;
    if (!(t >= DBHIP_T_BOOL && t <= DBHIP_T_STRING) || keys[k].is_scalar || (nbounds && (bounds[k].type != t || bounds[k].is_scalar))) {
      set_error("dbhip_sort_bound_partition: key %d: unsupported type %d, a scalar, or the bound column has another type", k, t);
      return DBHIP_ERR_UNSUPPORTED;
    }
    int nparts = 0;
    if (t == DBHIP_T_STRING) {   // the longest value of the rows AND the bounds decides the number of images, as in dbhip_sort_perm
      uint32_t* flag = (uint32_t*)scratch(64, 7, s);
      if (!flag) return DBHIP_ERR_HIP;
      DBHIP_CHECK(hipMemsetAsync(flag, 0, 4, s));
      hipLaunchKernelGGL(sort_max_len_kernel, dim3(grid), dim3(256), 0, s, (const uint32_t*)keys[k].data, n, flag);
      if (nbounds) hipLaunchKernelGGL(sort_max_len_kernel, dim3(grid_for(nbounds, 256)), dim3(256), 0, s, (const uint32_t*)bounds[k].data, nbounds, flag);
      uint32_t mx = 0;
      DBHIP_CHECK(hipMemcpyAsync(&mx, flag, 4, hipMemcpyDeviceToHost, s));
      DBHIP_CHECK(hipStreamSynchronize(s));
      if (mx > 12) {
        if (!keys[k].buffers || (nbounds && !bounds[k].buffers)) {
          set_error("dbhip_sort_bound_partition: string key %d holds values longer than 12 bytes but the rows or the bounds carry no data buffers", k);
          return DBHIP_ERR_INVALID;
        }
        if (mx > 4096) { set_error("dbhip_sort_bound_partition: string key %d holds a %u-byte value (> 4096: keep the CPU operator for this block)", k, mx); return DBHIP_ERR_UNSUPPORTED; }
        nparts = 1 + (int)((mx + 7) / 8);
      }
    }
    const uint8_t d = desc_host ? desc_host[k] : 0, nf = nulls_first_host ? nulls_first_host[k] : 0;
    bk.row[k] = SortCol{keys[k].data, keys[k].validity, keys[k].validity_offset, t, d, nf, nparts, keys[k].buffers};
    if (nbounds) bk.bnd[k] = SortCol{bounds[k].data, bounds[k].validity, bounds[k].validity_offset, t, d, nf, nparts, bounds[k].buffers};
    bk.has_flag[k] = (keys[k].validity || (nbounds && bounds[k].validity)) ? 1 : 0;
    bk.parts[k] = (uint8_t)((t == DBHIP_T_DEC128 || t == DBHIP_T_STRING) ? 2 : 1);
    bk.nimg += (nparts ? nparts : bk.parts[k]) + bk.has_flag[k];
  }
  const size_t img_bytes = (size_t)nbounds * bk.nimg * 8;
  uint64_t* img = (uint64_t*)scratch(img_bytes + 256, 7, s);
  if (!img) return DBHIP_ERR_HIP;
  if (nbounds) hipLaunchKernelGGL(sort_bound_encode_kernel, dim3((unsigned)ceil_div(nbounds, 256)), dim3(256), 0, s, bk, (int)nbounds, img);
  const int nhist = nbounds + 1 <= BOUND_HIST_MAX ? (int)nbounds + 1 : 0;
  const size_t hist_bytes = (size_t)((nhist + 1) / 2) * 8;
  const bool small = nkeys == 1 && bk.nimg == 1 && nbounds >= 1 && nbounds <= BOUND_SMALL_MAX && keys[0].type != DBHIP_T_STRING && keys[0].type != DBHIP_T_DEC128;
  if (small) {
    const dim3 g(grid_for(n, 256)), blk(256);
    if (nbounds <= 8) hipLaunchKernelGGL(sort_bound_partition_small_kernel<8>, g, blk, 0, s, bk, n, (int)nbounds, img, out_part, (unsigned long long*)out_counts);
    else if (nbounds <= 16) hipLaunchKernelGGL(sort_bound_partition_small_kernel<16>, g, blk, 0, s, bk, n, (int)nbounds, img, out_part, (unsigned long long*)out_counts);
    else if (nbounds <= 32) hipLaunchKernelGGL(sort_bound_partition_small_kernel<32>, g, blk, 0, s, bk, n, (int)nbounds, img, out_part, (unsigned long long*)out_counts);
    else hipLaunchKernelGGL(sort_bound_partition_small_kernel<64>, g, blk, 0, s, bk, n, (int)nbounds, img, out_part, (unsigned long long*)out_counts);
  } else if (img_bytes <= 32768)
    hipLaunchKernelGGL(sort_bound_partition_kernel<true>, dim3(grid), dim3(256), hist_bytes + img_bytes, s, bk, n, (int)nbounds, img, out_part,
                       (unsigned long long*)out_counts);
  else
    hipLaunchKernelGGL(sort_bound_partition_kernel<false>, dim3(grid), dim3(256), hist_bytes, s, bk, n, (int)nbounds, img, out_part,
                       (unsigned long long*)out_counts);
  DBHIP_LAUNCH_CHECK();
  DBHIP_CHECK(hipStreamSynchronize(s));  // scratch is reused by the next call
  return DBHIP_OK;
}

int32_t dbhip_scatter_block(const void* const* srcs_host, const int32_t* elem_sizes_host, int32_t ncols, const uint32_t* index, int64_t n,
                            uint32_t scatter_size, void* const* outs_host, void* stream) {
  DBHIP_REQUIRE(ncols >= 0 && n >= 0 && n < 0xFFFFFFFFLL && scatter_size >= 1, "dbhip_scatter_block: bad argument");
  if (n == 0 || ncols == 0) return DBHIP_OK;
  DBHIP_REQUIRE(srcs_host && elem_sizes_host && outs_host && index, "dbhip_scatter_block: NULL argument");
  bool direct = scatter_size <= 256;
  for (int c = 0; c < ncols; ++c) {
    const int es = elem_sizes_host[c];
    DBHIP_REQUIRE(es == 1 || es == 2 || es == 4 || es == 8 || es == 16, "dbhip_scatter_block: elem_size must be 1, 2, 4, 8 or 16");
    DBHIP_REQUIRE(srcs_host[c] && outs_host[c], "dbhip_scatter_block: NULL column");
    if (es == 16) direct = false;
  }
  hipStream_t s = resolve_stream(stream);
  if (!direct) {   // many destinations or 16-byte elements: the stable permutation of the rows + one gather
    uint32_t* perm = (uint32_t*)scratch((size_t)n * 4 + 64, 10, s);
    if (!perm) return DBHIP_ERR_HIP;
    dbhip_col key;
    memset(&key, 0, sizeof(key));
    key.type = DBHIP_T_U32;
    key.data = index;
    const uint8_t zero = 0;
    int32_t rc = dbhip_sort_perm(&key, &zero, &zero, 1, n, 0, perm, stream);
    if (rc) return rc;
    for (int c0 = 0; c0 < ncols; c0 += 8) {
      const int g = ncols - c0 < 8 ? ncols - c0 : 8;
      if ((rc = dbhip_take_block(srcs_host + c0, elem_sizes_host + c0, g, perm, n, outs_host + c0, stream))) return rc;
    }
    DBHIP_CHECK(hipStreamSynchronize(s));  // scratch is reused by the next call
    return DBHIP_OK;
  }
  const int64_t ntiles = ceil_div(n, SORT_TILE);
  const int64_t nh = 256 * ntiles;
  uint8_t* ws = (uint8_t*)scratch((size_t)nh * 4 + (size_t)nh * 8 + (size_t)(nh / SCAN_TILE + 2) * 8 + 256, 10, s);
  if (!ws) return DBHIP_ERR_HIP;
  uint64_t* offs = (uint64_t*)ws;
  uint64_t* blk = offs + nh;
  uint32_t* hist = (uint32_t*)(blk + nh / SCAN_TILE + 2);
  hipLaunchKernelGGL(sort_hist_kernel<uint32_t>, dim3((unsigned)ntiles), dim3(256), 0, s, index, n, 0, hist, ntiles);
  int32_t rc = dbscan::exclusive_scan_u32(hist, nh, blk, offs, s);
  if (rc) return rc;
  for (int c = 0; c < ncols; ++c) {
    switch (elem_sizes_host[c]) {
      case 1: hipLaunchKernelGGL(scatter_rows_kernel<uint8_t>, dim3((unsigned)ntiles), dim3(256), 0, s, index, (const uint8_t*)srcs_host[c], n, offs, ntiles, (uint8_t*)outs_host[c]); break;
      case 2: hipLaunchKernelGGL(scatter_rows_kernel<uint16_t>, dim3((unsigned)ntiles), dim3(256), 0, s, index, (const uint16_t*)srcs_host[c], n, offs, ntiles, (uint16_t*)outs_host[c]); break;
      case 4: hipLaunchKernelGGL(scatter_rows_kernel<uint32_t>, dim3((unsigned)ntiles), dim3(256), 0, s, index, (const uint32_t*)srcs_host[c], n, offs, ntiles, (uint32_t*)outs_host[c]); break;
      default: hipLaunchKernelGGL(scatter_rows_kernel<uint64_t>, dim3((unsigned)ntiles), dim3(256), 0, s, index, (const uint64_t*)srcs_host[c], n, offs, ntiles, (uint64_t*)outs_host[c]); break;
    }
  }
  DBHIP_LAUNCH_CHECK();
  DBHIP_CHECK(hipStreamSynchronize(s));  // scratch is reused by the next call
  return DBHIP_OK;
}

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
  const int64_t ntiles_max = ceil_div(n, SORT_TILE);
  const int64_t nh_max = 256 * ntiles_max;
  const int64_t nflag = ceil_div(n, 1024);
  // scratch: 2 key buffers, 2 perm buffers, hist, offsets, scan block sums, span (or/and), flags
  const int64_t nscan = (nh_max > nflag ? nh_max : nflag);
  size_t bytes = (size_t)n * 8 * 2 + (size_t)n * 4 * 2 + (size_t)nscan * 4 + (size_t)nscan * 8 + (size_t)(nscan / SCAN_TILE + 2) * 8 + 2048;
  uint8_t* ws = (uint8_t*)scratch(bytes + 64, 7, s);
  if (!ws) return DBHIP_ERR_HIP;
  uint32_t* long_flag = (uint32_t*)(ws + bytes);
  uint64_t* kb[2] = {(uint64_t*)ws, (uint64_t*)ws + n};
  uint64_t* offs = kb[1] + n;
  uint64_t* blk = offs + nscan;
  unsigned long long* span = (unsigned long long*)(blk + nscan / SCAN_TILE + 2);
  uint32_t* hist = (uint32_t*)(span + 4);
  uint32_t* pb[2] = {hist + nscan + 256, hist + nscan + 256 + n};
  int cur = 0;
  const int grid = grid_for(n, 256);
  // string keys: the longest value decides how many 8-byte images a value has (<= 12 bytes: the two images of the inline view)
  int str_parts[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int k = 0; k < nkeys; ++k) {
    if (keys[k].type != DBHIP_T_STRING) continue;
    DBHIP_CHECK(hipMemsetAsync(long_flag, 0, 4, s));
    hipLaunchKernelGGL(sort_max_len_kernel, dim3(grid), dim3(256), 0, s, (const uint32_t*)keys[k].data, n, long_flag);
    uint32_t mx = 0;
    DBHIP_CHECK(hipMemcpyAsync(&mx, long_flag, 4, hipMemcpyDeviceToHost, s));
    DBHIP_CHECK(hipStreamSynchronize(s));
    if (mx > 12) {
      if (!keys[k].buffers) { set_error("dbhip_sort_perm: string key %d holds values longer than 12 bytes but no data buffers", k); return DBHIP_ERR_INVALID; }
      if (mx > 4096) { set_error("dbhip_sort_perm: string key %d holds a %u-byte value (> 4096: keep the CPU operator for this block)", k, mx); return DBHIP_ERR_UNSUPPORTED; }
      str_parts[k] = 1 + (int)((mx + 7) / 8);
    }
  }
  hipLaunchKernelGGL(sort_iota_kernel, dim3(grid), dim3(256), 0, s, pb[cur], n);

  auto make_col = [&](int k) {
    return SortCol{keys[k].data, keys[k].validity, keys[k].validity_offset, keys[k].type,
                   desc_host ? desc_host[k] : 0, nulls_first_host ? nulls_first_host[k] : 0, str_parts[k], keys[k].buffers};
  };
  // enc = image of column c read through pb[cur][0..m); returns OR / AND of all images
  // `narrow`: 32-bit key images (columns of at most 4 bytes, null flags) in the same buffers
  auto encode = [&](const SortCol& c, int64_t m, int mode, bool narrow, uint64_t* v_or, uint64_t* v_and) -> int32_t {
    DBHIP_CHECK(hipMemsetAsync(&span[0], 0x00, 8, s));
    DBHIP_CHECK(hipMemsetAsync(&span[1], 0xFF, 8, s));
    if (narrow)
      hipLaunchKernelGGL(sort_encode_kernel<uint32_t>, dim3(grid_for(m, 256)), dim3(256), 0, s, c, pb[cur], m, mode, (uint32_t*)kb[cur], span);
    else
      hipLaunchKernelGGL(sort_encode_kernel<uint64_t>, dim3(grid_for(m, 256)), dim3(256), 0, s, c, pb[cur], m, mode, kb[cur], span);
    DBHIP_LAUNCH_CHECK();
    unsigned long long h[2];
    DBHIP_CHECK(hipMemcpyAsync(h, span, 16, hipMemcpyDeviceToHost, s));
    DBHIP_CHECK(hipStreamSynchronize(s));
    *v_or = h[0];
    *v_and = h[1];
    return DBHIP_OK;
  };

  int64_t m = n;  // rows being sorted (the candidates of the LIMIT select, else all rows)

  // ---- LIMIT (LimitType::LimitRows, sort_compare.rs:197-209 uses select_nth_unstable + sort):
  // radix select on the most significant key finds a threshold image T with at least `limit` rows
  // <= T; only those rows are sorted.
  if (limit > 0 && limit * 4 < n && n >= 65536 && !keys[0].validity) {
    SortCol c = make_col(0);
    const int top_part = c.nparts ? c.nparts - 1 : ((c.type == DBHIP_T_DEC128 || c.type == DBHIP_T_STRING) ? 1 : 0);
    uint64_t v_or, v_and;
    int32_t rc = encode(c, n, top_part, false, &v_or, &v_and);
    if (rc) return rc;
    const uint64_t vary = v_or ^ v_and;
    uint64_t prefix = v_and & ~vary;  // constant bits (exact in constant bytes; varying bytes are set below)
    for (int b = 0; b < 8; ++b)
      if ((vary >> (8 * b)) & 0xFF) prefix &= ~(0xFFULL << (8 * b));
    int64_t k_rem = limit, less_total = 0, bucket = n;
    uint64_t threshold = ~0ULL;
    bool decided = false;
    for (int b = 7; b >= 0 && !decided; --b) {
      if (((vary >> (8 * b)) & 0xFF) == 0) continue;
      DBHIP_CHECK(hipMemsetAsync(hist, 0, 256 * 4, s));
      hipLaunchKernelGGL(sort_select_hist_kernel, dim3(grid), dim3(256), 0, s, kb[cur], n, prefix, 8 * b, hist);
      DBHIP_LAUNCH_CHECK();
      uint32_t hh[256];
      DBHIP_CHECK(hipMemcpyAsync(hh, hist, sizeof(hh), hipMemcpyDeviceToHost, s));
      DBHIP_CHECK(hipStreamSynchronize(s));
      int64_t cum = 0;
      int d = 0;
      for (; d < 256; ++d) {
        if (cum + hh[d] >= k_rem) break;
        cum += hh[d];
      }
      if (d == 256) d = 255;  // (cannot happen: the bucket holds >= k_rem rows)
      prefix |= (uint64_t)d << (8 * b);
      k_rem -= cum;
      less_total += cum;
      bucket = hh[d];
      threshold = prefix | ((b == 0) ? 0ULL : ((1ULL << (8 * b)) - 1));
      // lower constant bytes are already part of prefix; lower varying bytes: take the whole bucket
      if (bucket <= (limit > 8192 ? limit : 8192)) decided = true;
    }
    if (threshold != ~0ULL) {
      // lower bits of the threshold: all ones in varying bytes below the last decided byte
      m = less_total + bucket;
      uint32_t* cnt = hist;
      hipLaunchKernelGGL(sort_select_flag_kernel, dim3((unsigned)nflag), dim3(256), 0, s, kb[cur], n, threshold, cnt);
      rc = dbscan::exclusive_scan_u32(cnt, nflag, blk, offs, s);
      if (rc) return rc;
      hipLaunchKernelGGL(sort_select_emit_kernel, dim3((unsigned)nflag), dim3(256), 0, s, kb[cur], n, threshold, offs, pb[cur ^ 1]);
      DBHIP_LAUNCH_CHECK();
      cur ^= 1;  // pb[cur] = ascending candidate row ids
    }
  }

  // 8192-key tiles of 512 threads (DBHIP_SORT_NT=256: the 4096-key tiles of rounds 1-3)
  static const int rs_nt = exp_env("DBHIP_SORT_NT") ? atoi(exp_env("DBHIP_SORT_NT")) : 512;
  const bool big_tiles = rs_nt == 512;
  const int64_t ntiles = ceil_div(m, big_tiles ? 512 * SORT_ITEMS : SORT_TILE);
  const int64_t nh = 256 * ntiles;
  // `final_vals`: this call holds the LAST pass of the whole sort — its permutation goes straight to the caller's buffer
  bool wrote_final = false;
  // onesweep (above) for sorts of 2^20 rows and more in 8192-key tiles; DBHIP_SORT_ONESWEEP=0: the histogram / scan / scatter passes
  static const bool onesweep_off = exp_env("DBHIP_SORT_ONESWEEP") && atoi(exp_env("DBHIP_SORT_ONESWEEP")) == 0;
  const bool onesweep = !onesweep_off && big_tiles && m >= (1 << 20);
  unsigned long long* os_ws = nullptr;   // [8][256] digit histograms | [8] x (ticket, stall flag) | [ntiles][256] status words (cleared per pass)
  const size_t os_words = (size_t)8 * 256 + 16 + (size_t)ntiles * 256;
  if (onesweep) {
    os_ws = (unsigned long long*)scratch(os_words * 8, 19, s);
    if (!os_ws) return DBHIP_ERR_HIP;
  }
  auto onesweep_passes = [&](int nbytes, uint64_t vary, bool narrow, uint32_t* final_vals) -> int32_t {
    int last_b = -1, npass = 0;
    uint32_t vary_bytes = 0;
    for (int b = 0; b < nbytes; ++b) if (((vary >> (8 * b)) & 0xFF) != 0) { last_b = b; vary_bytes |= 1u << b; ++npass; }
    if (npass == 0) return DBHIP_OK;
    DBHIP_POLL_CANCEL(s, "dbhip_sort_perm");
    DBHIP_CHECK(hipMemsetAsync(os_ws, 0, ((size_t)8 * 256 + 16) * 8, s));
    unsigned long long* ghist = os_ws;
    uint32_t* ctl = (uint32_t*)(os_ws + 8 * 256);            // per pass: ticket, stall flag
    unsigned long long* status = os_ws + 8 * 256 + 16;
    const int hgrid = (int)(ntiles < 2048 ? ntiles : 2048);
    if (narrow) hipLaunchKernelGGL((sort_onesweep_hist_kernel<uint32_t, 4>), dim3(hgrid), dim3(256), 0, s, (const uint32_t*)kb[cur], m, vary_bytes, ghist);
    else hipLaunchKernelGGL((sort_onesweep_hist_kernel<uint64_t, 8>), dim3(hgrid), dim3(256), 0, s, (const uint64_t*)kb[cur], m, vary_bytes, ghist);
    int pass = 0;
    for (int b = 0; b < nbytes; ++b) {
      if (!((vary_bytes >> b) & 1u)) continue;
      uint32_t* ov = (b == last_b && final_vals) ? final_vals : pb[cur ^ 1];
      // (one status array, cleared in stream order before every pass: 2 KiB per tile — 150 MB for 600 M keys — instead of that per pass)
      DBHIP_CHECK(hipMemsetAsync(status, 0, (size_t)ntiles * 256 * 8, s));
      if (narrow)
        hipLaunchKernelGGL((sort_onesweep_kernel<uint32_t, 512>), dim3((unsigned)ntiles), dim3(512), 0, s, (const uint32_t*)kb[cur], pb[cur], m, 8 * b, ghist + 256 * b,
                           status, ctl + 2 * pass, b == last_b ? (uint32_t*)nullptr : (uint32_t*)kb[cur ^ 1], ov);
      else
        hipLaunchKernelGGL((sort_onesweep_kernel<uint64_t, 512>), dim3((unsigned)ntiles), dim3(512), 0, s, (const uint64_t*)kb[cur], pb[cur], m, 8 * b, ghist + 256 * b,
                           status, ctl + 2 * pass, b == last_b ? (uint64_t*)nullptr : (uint64_t*)kb[cur ^ 1], ov);
      if (b == last_b && final_vals) wrote_final = true;
      cur ^= 1;
      ++pass;
    }
    DBHIP_LAUNCH_CHECK();
    uint32_t hctl[16];
    DBHIP_CHECK(hipMemcpyAsync(hctl, ctl, sizeof(hctl), hipMemcpyDeviceToHost, s));
    DBHIP_CHECK(hipStreamSynchronize(s));
    for (int p = 0; p < npass; ++p)
      if (hctl[2 * p + 1]) { set_error("dbhip_sort_perm: a tile's look-back got no answer within its bounded wait (pass %d); the output buffer is undefined, the call can be repeated", p); return DBHIP_ERR_HIP; }
    return DBHIP_OK;
  };
  auto radix_passes = [&](int nbytes, uint64_t vary, bool narrow, uint32_t* final_vals = nullptr) -> int32_t {
    // (64-bit images only: r05, 64 M keys — 0.535 ms per onesweep pass against 0.725 for histogram + scan + scatter; on 32-bit images
    //  the look-back pass takes 0.49 ms against 0.45 for the three kernels, whose scatter moves 16 B per key in 0.35 ms)
    if (onesweep && !narrow) return onesweep_passes(nbytes, vary, narrow, final_vals);
    int last_b = -1;
    for (int b = 0; b < nbytes; ++b) if (((vary >> (8 * b)) & 0xFF) != 0) last_b = b;
    for (int b = 0; b < nbytes; ++b) {
      if (((vary >> (8 * b)) & 0xFF) == 0) continue;  // every image has the same byte here
      DBHIP_POLL_CANCEL(s, "dbhip_sort_perm");
#define RS_PASS(K_, NT_, KEYS_, OUT_)                                                                                                     \
  do {                                                                                                                                    \
    hipLaunchKernelGGL((sort_hist_kernel<K_, NT_>), dim3((unsigned)ntiles), dim3(NT_), 0, s, (const K_*)(KEYS_), m, 8 * b, hist, ntiles);  \
    int32_t rc_ = dbscan::exclusive_scan_u32(hist, nh, blk, offs, s);                                                                     \
    if (rc_) return rc_;                                                                                                                  \
    hipLaunchKernelGGL((sort_scatter_kernel<K_, NT_>), dim3((unsigned)ntiles), dim3(NT_), 0, s, (const K_*)(KEYS_), pb[cur], m, 8 * b,    \
                       offs, ntiles, b == last_b ? (K_*)nullptr : (K_*)(OUT_), (b == last_b && final_vals) ? final_vals : pb[cur ^ 1]);   \
  } while (0)
      if (narrow) { if (big_tiles) RS_PASS(uint32_t, 512, kb[cur], kb[cur ^ 1]); else RS_PASS(uint32_t, 256, kb[cur], kb[cur ^ 1]); }
      else { if (big_tiles) RS_PASS(uint64_t, 512, kb[cur], kb[cur ^ 1]); else RS_PASS(uint64_t, 256, kb[cur], kb[cur ^ 1]); }
#undef RS_PASS
      if (b == last_b && final_vals) wrote_final = true;
      cur ^= 1;
    }
    DBHIP_LAUNCH_CHECK();
    return DBHIP_OK;
  };

  const int64_t mout = (limit > 0 && limit < m) ? limit : m;
  const bool direct_out = mout == m;   // (a LIMIT copies its first rows out of the scratch permutation instead)
  for (int k = nkeys - 1; k >= 0; --k) {
    SortCol c = make_col(k);
    const int parts = c.nparts ? c.nparts : ((c.type == DBHIP_T_DEC128 || c.type == DBHIP_T_STRING) ? 2 : 1);
    uint64_t v_or, v_and;
    const bool narrow = sort_key_bytes(c.type) <= 4;
    for (int part = 0; part < parts; ++part) {
      // the permutation is shared by both buffers of a pass: encode reads pb[cur], writes kb[cur]
      int32_t rc = encode(c, m, part, narrow, &v_or, &v_and);
      if (rc) return rc;
      const bool last_call = k == 0 && part == parts - 1 && !c.validity && direct_out;
      if ((rc = radix_passes(sort_key_bytes(c.type), v_or ^ v_and, narrow, last_call ? out_perm : nullptr))) return rc;
    }
    if (c.validity) {
      int32_t rc = encode(c, m, 2, true, &v_or, &v_and);
      if (rc) return rc;
      if ((rc = radix_passes(1, v_or ^ v_and, true, (k == 0 && direct_out) ? out_perm : nullptr))) return rc;
    }
  }
  if (!wrote_final) DBHIP_CHECK(hipMemcpyAsync(out_perm, pb[cur], (size_t)mout * 4, hipMemcpyDeviceToDevice, s));
  DBHIP_CHECK(hipStreamSynchronize(s));  // scratch is reused by the next call
  return DBHIP_OK;
}

// k-way merge of sorted runs (a16 merge row). Reference: Merger over a SortAlgorithm (HeapSort / LoserTreeSort,
// sorts/core/{merger.rs,algorithm.rs:33-61,loser_tree.rs}) pops the smallest cursor row of N sorted streams until
// `limit` rows are out; equal rows have no defined order between streams (cursor order is the row order only,
// algorithm.rs:215-223). Device: the runs sit back to back in the key columns (run r = rows
// [run_offsets[r], run_offsets[r+1])); the merged order is the STABLE order of all rows by key, i.e. ties come
// out by (run, position) — one of the orders the loser tree may produce. Computed with the radix machinery above
// (constant-byte passes skipped, LIMIT by radix select): a sequential loser tree has no data-parallel analogue,
// and a merge-path tree would read each row log2(N) times against <= 8 radix passes here.
int32_t dbhip_merge_sorted_perm(const dbhip_col* keys, const uint8_t* desc_host, const uint8_t* nulls_first_host,
                                int32_t nkeys, const int64_t* run_offsets_host, int32_t nruns, int64_t limit,
                                uint32_t* out_perm, void* stream) {
  DBHIP_REQUIRE(run_offsets_host && nruns >= 0, "dbhip_merge_sorted_perm: NULL run offsets");
  for (int r = 0; r < nruns; ++r)
    DBHIP_REQUIRE(run_offsets_host[r] <= run_offsets_host[r + 1], "dbhip_merge_sorted_perm: run offsets must ascend");
  const int64_t n = nruns ? run_offsets_host[nruns] : 0;
  DBHIP_REQUIRE(nruns == 0 || run_offsets_host[0] == 0, "dbhip_merge_sorted_perm: runs start at row 0");
  return dbhip_sort_perm(keys, desc_host, nulls_first_host, nkeys, n, limit, out_perm, stream);
}

}  // extern "C"

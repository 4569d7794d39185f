// k_hnsw.hip — HNSW vector index with u8 scalar quantisation (the reference's indexed ANN path).
//
// Reference: src/query/storages/common/index/src/hnsw_index/
//   hnsw.rs:62-374                                  HNSWIndex::{build, open, search, generate_scores}, cosine_preprocess
//   quantization/encoded_vectors_u8.rs:54-413       EncodedVectorsU8
//   graph_layers.rs:72-247, search_context.rs       search_entry / search_on_level / SearchContext
//   graph_layers_builder.rs:300-520                 link_new_point, the "not closer than base" heuristic
//   entry_points.rs:56-120
//
// Device design (gfx950): ONE WAVE PER QUERY (search) or PER NEW POINT (build). A graph walk is a chain of dependent
// gathers, so the unit of parallelism is the walk, not the row: the 64 lanes of a wave share one walk, score its neighbours
// cooperatively (a 768-byte u8 code row or a 3 KB f32 row is one coalesced wave load, four rows in flight per step) and lane 0
// keeps the two priority queues in LDS with exactly the sift order of std::collections::BinaryHeap, so that equal scores
// leave the queues in the reference's order (ScoredPointOffset orders by score only, common/types.rs:38-42). The visited set
// is an open-addressing table in global memory written by lane 0 only. Thousands of walks in flight hide the gather latency.
// Quantised vectors are kept as codes[n][actual_dim] (16-byte aligned rows) + offsets[n]; dbhip_hnsw_encoded writes the
// reference's interleaved storage layout.
//
// f32 arithmetic that must equal the reference's bit for bit (the quantiser, cosine_preprocess, score_point) uses the
// explicit round-to-nearest intrinsics in the reference's evaluation order: hipcc would otherwise contract a*b+c.
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <new>
#include <vector>

#include "dev_common.h"
#include "runtime.h"

using namespace dbhip;

namespace {

constexpr int HN_ALIGN = 16;          // encoded_vectors_u8.rs:33
constexpr int HN_MAX_M0 = 64;         // m0 = 2 m <= 64: a link list is one wave-wide load
constexpr int HN_MAX_EF = 256;        // ef = 4 * limit, limit <= 64
constexpr int HN_CCAP = 2048;         // candidates heap (LDS)
constexpr int HN_VCAP = 16384;        // visited table per wave (global), power of two
constexpr int HN_MAX_ADIM = 4096;
constexpr int HN_SEQ = 256;           // SINGLE_THREADED_HNSW_BUILD_THRESHOLD (hnsw.rs:53)

enum { D_DOT = 0, D_L1 = 1, D_L2 = 2 };

struct HnswView {   // by-value kernel argument
  int64_t n;
  int dim, adim, distance, m, m0;
  float alpha, offset, mult;
  const float* raw;            // original vectors (build only)
  const float* vlen;           // cosine: sqrt of the squared length, or 0 = leave as it is (cosine_preprocess)
  const uint8_t* codes;        // [n][adim]
  const float* voff;           // [n]
  uint32_t* links0;            // [n][m0]
  uint32_t* cnt0;              // [n]
  uint32_t* linksu;            // [n_upper_lists][m]
  uint32_t* cntu;              // [n_upper_lists]
  const int64_t* ufirst;       // [n] first upper list of the point (level 1), -1 = none
  const int32_t* level;        // [n]
  uint32_t* ready;             // [n]
  uint32_t* lock;              // [n]
  unsigned long long* entry;   // packed (level << 32 | ~idx), 0 = none
  int exact_scores;            // build scorer in the reference's sequential f32 order (the deterministic build)
};

struct dbhip_hnsw_impl {
  HnswView v;
  int small_search_off;   // a batch outgrew the LDS-resident search tables on THIS index: later batches take the general kernel directly
  int ef_construct;
  int64_t n_upper;
  std::vector<int32_t> level_host;
  std::vector<int64_t> ufirst_host;
  void* owned[12];
  int n_owned;
};

// ---------------------------------------------------------------------------------------------------------------------
// quantiser
// ---------------------------------------------------------------------------------------------------------------------
// correctly rounded f32 sqrt (a 53-bit sqrt rounded once more to 24 bits is the correctly rounded result: 53 >= 2 * 24 + 2)
__device__ __forceinline__ float hn_sqrt(float x) { return (float)sqrt((double)x); }
// cosine_preprocess (hnsw.rs:362-374): length = SEQUENTIAL f32 sum of x * x; one thread per vector keeps the order
__global__ __launch_bounds__(256) void hn_length_kernel(const float* data, int64_t n, int dim, float* vlen) {
  const int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (v >= n) return;
  const float* x = data + v * dim;
  float len = 0.0f;
  for (int i = 0; i < dim; ++i) len = __fadd_rn(len, __fmul_rn(x[i], x[i]));
  const bool keep = len < 1.1920929e-7f || fabsf(__fsub_rn(len, 1.0f)) <= 1.0e-6f;
  vlen[v] = keep ? 0.0f : hn_sqrt(len);
}
__device__ __forceinline__ float hn_value(const float* data, const float* vlen, int64_t v, int dim, int i) {
  const float x = data[v * dim + i];
  if (!vlen) return x;
  const float l = vlen[v];
  return l == 0.0f ? x : __fdiv_rn(x, l);
}
__device__ __forceinline__ uint32_t f32_order(float f) {   // monotone u32 image of a float
  const uint32_t b = __float_as_uint(f);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
// find_min_max_from_iter (quantile.rs:24-38) over the pre-processed values: mm[0] = min image, mm[1] = max image
__global__ __launch_bounds__(256) void hn_minmax_kernel(const float* data, const float* vlen, int64_t n, int dim, uint32_t* mm) {
  uint32_t lo = 0xFFFFFFFFu, hi = 0;
  const int64_t total = n * dim;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const float x = hn_value(data, vlen, e / dim, dim, (int)(e % dim));
    if (x != x) continue;   // NaN compares false both ways in the reference's fold
    const uint32_t o = f32_order(x);
    lo = o < lo ? o : lo;
    hi = o > hi ? o : hi;
  }
  for (int off = 32; off >= 1; off >>= 1) {
    const uint32_t l2 = __shfl_xor(lo, off, 64), h2 = __shfl_xor(hi, off, 64);
    lo = l2 < lo ? l2 : lo;
    hi = h2 > hi ? h2 : hi;
  }
  if ((threadIdx.x & 63) == 0) { atomicMin(&mm[0], lo); atomicMax(&mm[1], hi); }
}
__device__ __forceinline__ uint8_t hn_f32_to_u8(float i, float alpha, float offset) {   // encoded_vectors_u8.rs:243-246
  float x = __fdiv_rn(__fsub_rn(i, offset), alpha);
  if (x != x) return 0;
  x = x < 0.0f ? 0.0f : (x > 127.0f ? 127.0f : x);
  return (uint8_t)x;
}
// code sums are integers (exact in f32 below 2^24: dim <= 1040 for squares), summed as integers and converted once — equal to
// the reference's sequential f32 sum wherever that sum is exact
__device__ __forceinline__ float hn_offset_term(int distance, int adim, float alpha, float offset, uint32_t s1, uint32_t s2, bool with_dim) {
  if (distance == D_L1) return 0.0f;
  const float base = with_dim ? __fmul_rn(__fmul_rn((float)adim, offset), offset) : 0.0f;
  const float t = distance == D_DOT ? __fmul_rn(__fmul_rn((float)s1, alpha), offset) : __fmul_rn(__fmul_rn((float)s2, alpha), alpha);
  return with_dim ? __fadd_rn(base, t) : t;
}
// encode (:95-146): 16 lanes per vector
__global__ __launch_bounds__(256) void hn_encode_kernel(const float* data, const float* vlen, int64_t n, int dim, int adim, int distance,
                                                        float alpha, float offset, uint8_t* codes, float* voff) {
  const int sub = threadIdx.x & 15;
  const int64_t v = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 4;
  uint32_t s1 = 0, s2 = 0;
  if (v < n) {
    const float placeholder = distance == D_DOT ? 0.0f : offset;
    for (int i = sub; i < adim; i += 16) {
      const float x = i < dim ? hn_value(data, vlen, v, dim, i) : placeholder;
      const uint8_t c = hn_f32_to_u8(x, alpha, offset);
      codes[v * adim + i] = c;
      s1 += c;
      s2 += (uint32_t)c * c;
    }
  }
  for (int off = 8; off >= 1; off >>= 1) { s1 += __shfl_xor(s1, off, 64); s2 += __shfl_xor(s2, off, 64); }
  if (v < n && sub == 0) {
    const float vo = hn_offset_term(distance, adim, alpha, offset, s1, s2, true);
    voff[v] = distance == D_DOT ? vo : -vo;   // invert (hnsw.rs:77-80)
  }
}
__global__ __launch_bounds__(256) void hn_layout_kernel(const uint8_t* codes, const float* voff, int64_t n, int adim, uint8_t* out) {
  const int64_t rec = adim + 4;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n * rec; e += (int64_t)gridDim.x * 256) {
    const int64_t v = e / rec;
    const int b = (int)(e % rec);
    uint8_t x;
    if (b < 4) { const uint32_t w = __float_as_uint(voff[v]); x = (uint8_t)(w >> (8 * b)); }
    else x = codes[v * adim + (b - 4)];
    out[e] = x;
  }
}

// the inverse: the reference's storage layout (f32 offset + actual_dim codes per vector) -> codes[n][adim] + voff[n] (HNSWIndex::open)
__global__ __launch_bounds__(256) void hn_unlayout_kernel(const uint8_t* in, int64_t n, int adim, uint8_t* codes, float* voff) {
  const int64_t rec = adim + 4;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n * rec; e += (int64_t)gridDim.x * 256) {
    const int64_t v = e / rec;
    const int b = (int)(e % rec);
    if (b >= 4) codes[v * adim + (b - 4)] = in[e];
    else if (b == 0) {
      const uint32_t w = (uint32_t)in[e] | ((uint32_t)in[e + 1] << 8) | ((uint32_t)in[e + 2] << 16) | ((uint32_t)in[e + 3] << 24);
      voff[v] = __uint_as_float(w);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// per-wave query state (LDS) and the scorers
// ---------------------------------------------------------------------------------------------------------------------
struct SP { uint32_t idx; float score; };

struct WaveLds {
  SP nearest[HN_MAX_EF];       // FixedLengthPriorityQueue = BinaryHeap<Reverse<SP>>
  uint32_t qcodes[HN_MAX_ADIM / 4];
  int nn, nc, ef;
  float qoff;
  SP cur;
  int flag;                    // bit 0: candidate heap full, bit 1: visited table full (bits 2, 3: the same in the small-search kernel)
  uint32_t vis_count;
  int ccap;                    // capacity of `cand` in THIS kernel
  SP cand[HN_CCAP];            // BinaryHeap<SP> — LAST: the small-search kernel allocates the struct with a shorter heap
};
constexpr int HN_SMALL_CCAP = 1024;   // small search (ef <= 64): candidate heap and LDS-resident visited table
constexpr int HN_SMALL_VCAP = 2048;

__device__ __forceinline__ int lane() { return threadIdx.x & 63; }
// lane 0 writes queue state in LDS, every lane reads it afterwards: order the accesses for the compiler (the LDS unit serves
// one wave's operations in order)
#define WAVE_SYNC()                                        \
  do {                                                     \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); \
    __builtin_amdgcn_wave_barrier();                       \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); \
  } while (0)
// Orders the link words / counts a wave has just stored before the count / ready flag / lock release it stores next, for
// readers on OTHER CUs and XCDs: an agent-scope release (buffer_wbl2 sc1 + s_waitcnt vmcnt(0)). A workgroup-scope fence is
// nothing another CU can observe (MI355X_MICROARCH.md, inter-workgroup visibility); the explicit s_waitcnt is the guide's fix for
// the compiler dropping the wait behind buffer_wbl2 when it can prove the vmcnt scoreboard empty.
#define PUBLISH_RELEASE()                                   \
  do {                                                      \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");      \
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        \
  } while (0)

// OrderedFloat: NaN is the greatest value and equal to itself
__device__ __forceinline__ bool sp_le(float a, float b) {
  if (a != a) return b != b;
  if (b != b) return true;
  return a <= b;
}
__device__ __forceinline__ bool sp_lt(float a, float b) { return !sp_le(b, a); }

// ---- BinaryHeap<SP> (max-heap), lane 0 only; std's sift order (see oracle/hnsw_oracle.c) ----
__device__ void heap_sift_up(SP* d, int start, int pos) {
  const SP e = d[pos];
  while (pos > start) {
    const int parent = (pos - 1) / 2;
    if (sp_le(e.score, d[parent].score)) break;
    d[pos] = d[parent];
    pos = parent;
  }
  d[pos] = e;
}
__device__ void heap_push(SP* d, int* len, SP v) {
  d[*len] = v;
  heap_sift_up(d, 0, *len);
  ++*len;
}
__device__ bool heap_pop(SP* d, int* len, SP* out) {
  if (*len == 0) return false;
  --*len;
  SP item = d[*len];
  if (*len > 0) {
    const SP t = d[0];
    d[0] = item;
    item = t;
    // sift_down_to_bottom(0)
    const int end = *len;
    int pos = 0;
    const SP e = d[0];
    int child = 1;
    while (end >= 2 && child <= end - 2) {
      child += sp_le(d[child].score, d[child + 1].score) ? 1 : 0;
      d[pos] = d[child];
      pos = child;
      child = 2 * pos + 1;
    }
    if (child == end - 1) { d[pos] = d[child]; pos = child; }
    d[pos] = e;
    heap_sift_up(d, 0, pos);
  }
  *out = item;
  return true;
}
// sift_down_range + into_sorted_vec of the max-heap (ascending by score; equal scores end where std's heap sort puts them)
__device__ void heap_sift_down_range(SP* d, int pos, int end) {
  const SP e = d[pos];
  int child = 2 * pos + 1;
  while (end >= 2 && child <= end - 2) {
    child += sp_le(d[child].score, d[child + 1].score) ? 1 : 0;
    if (sp_le(d[child].score, e.score)) { d[pos] = e; return; }   // hole.element() >= hole.get(child)
    d[pos] = d[child];
    pos = child;
    child = 2 * pos + 1;
  }
  if (child == end - 1 && sp_lt(e.score, d[child].score)) { d[pos] = d[child]; pos = child; }
  d[pos] = e;
}
__device__ void heap_into_sorted(SP* d, int len) {
  int end = len;
  while (end > 1) {
    --end;
    const SP t = d[0]; d[0] = d[end]; d[end] = t;
    heap_sift_down_range(d, 0, end);
  }
}
// ---- BinaryHeap<Reverse<SP>>: the same with the order reversed ----
__device__ __forceinline__ bool r_le(float a, float b) { return sp_le(b, a); }
__device__ void rheap_sift_down_range(SP* d, int pos, int end) {
  const SP e = d[pos];
  int child = 2 * pos + 1;
  while (end >= 2 && child <= end - 2) {
    child += r_le(d[child].score, d[child + 1].score) ? 1 : 0;
    if (r_le(d[child].score, e.score)) { d[pos] = e; return; }
    d[pos] = d[child];
    pos = child;
    child = 2 * pos + 1;
  }
  if (child == end - 1 && !r_le(d[child].score, e.score)) { d[pos] = d[child]; pos = child; }
  d[pos] = e;
}
// FixedLengthPriorityQueue::push (:52-66): true + *removed when the queue was full
__device__ bool flpq_push(SP* d, int* len, int length, SP v, SP* removed) {
  if (*len < length) {
    int pos = (*len)++;
    d[pos] = v;
    const SP e = v;
    while (pos > 0) {
      const int parent = (pos - 1) / 2;
      if (r_le(e.score, d[parent].score)) break;
      d[pos] = d[parent];
      pos = parent;
    }
    d[pos] = e;
    return false;
  }
  if (sp_lt(d[0].score, v.score)) {
    const SP t = d[0];
    d[0] = v;
    v = t;
    rheap_sift_down_range(d, 0, *len);
  }
  *removed = v;
  return true;
}
__device__ void flpq_into_sorted(SP* d, int len) {   // descending by score
  int end = len;
  while (end > 1) {
    --end;
    const SP t = d[0]; d[0] = d[end]; d[end] = t;
    rheap_sift_down_range(d, 0, end);
  }
}
// SearchContext::process_candidate (search_context.rs:52-60), lane 0
__device__ void process_candidate(WaveLds* W, SP sp) {
  SP removed;
  const bool full = flpq_push(W->nearest, &W->nn, W->ef, sp, &removed);
  const bool was_added = !full || removed.idx != sp.idx;
  if (was_added) {
    if (W->nc >= W->ccap) { W->flag |= 1; return; }
    heap_push(W->cand, &W->nc, sp);
  }
}

// ---- visited table: open addressing, written by lane 0 only, read by every lane. VCAP = HN_VCAP: in global memory (the build and
// searches with ef > 64); VCAP = HN_SMALL_VCAP: in LDS (r03: a visited check is on the critical path of every step of the walk —
// one global round trip per step — and clearing 64 KB of global memory per query with scalar sc1 stores is not free either) ----
typedef volatile __attribute__((address_space(3))) uint32_t* HnLdsU32;
__device__ __forceinline__ uint32_t vis_hash(uint32_t id) { return (id * 2654435761u) >> 7; }
template <int VCAP>
__device__ __forceinline__ uint32_t vis_ld(const uint32_t* vis, uint32_t h) {
  if (VCAP == HN_SMALL_VCAP) return ((HnLdsU32)(uint32_t*)vis)[h];
  return __hip_atomic_load(&vis[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <int VCAP>
__device__ __forceinline__ void vis_st(uint32_t* vis, uint32_t h, uint32_t v) {
  if (VCAP == HN_SMALL_VCAP) ((HnLdsU32)vis)[h] = v;
  else __hip_atomic_store(&vis[h], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <int VCAP>
__device__ __forceinline__ bool vis_check(const uint32_t* vis, uint32_t id) {
  uint32_t h = vis_hash(id) & (VCAP - 1);
  for (int step = 0; step < VCAP; ++step) {
    const uint32_t x = vis_ld<VCAP>(vis, h);
    if (x == id + 1) return true;
    if (x == 0) return false;
    h = (h + 1) & (VCAP - 1);
  }
  return false;
}
template <int VCAP>
__device__ void vis_insert(uint32_t* vis, WaveLds* W, uint32_t id) {   // lane 0
  if (W->vis_count >= (VCAP == HN_SMALL_VCAP ? VCAP * 3 / 4 : VCAP / 2)) { W->flag |= 2; return; }
  uint32_t h = vis_hash(id) & (VCAP - 1);
  for (;;) {
    const uint32_t x = vis_ld<VCAP>(vis, h);
    if (x == id + 1) return;
    if (x == 0) break;
    h = (h + 1) & (VCAP - 1);
  }
  vis_st<VCAP>(vis, h, id + 1);
  ++W->vis_count;
}
template <int VCAP>
__device__ void vis_clear(uint32_t* vis) {
  for (int i = lane(); i < VCAP; i += 64) vis_st<VCAP>(vis, i, 0u);
}

// ---- link lists (atomic loads / stores: they change while other waves read them during a build) ----
__device__ __forceinline__ uint32_t* list_ptr(const HnswView& H, uint32_t p, int level, uint32_t** cnt) {
  if (level == 0) { *cnt = H.cnt0 + p; return H.links0 + (size_t)p * H.m0; }
  const int64_t l = H.ufirst[p] + (level - 1);
  *cnt = H.cntu + l;
  return H.linksu + (size_t)l * H.m;
}
__device__ __forceinline__ uint32_t ld32(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st32(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ float wave_sum_f(float v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
__device__ __forceinline__ uint32_t wave_sum_u(uint32_t v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// quantised score of up to 4 points against the wave's query (score_point_simple, :163-229): every lane gets all results
__device__ __forceinline__ void score_quant4(const HnswView& H, const WaveLds* W, const uint32_t (&id)[4], int cnt, float (&out)[4]) {
  const int D = H.adim >> 2;
  uint32_t acc[4] = {0, 0, 0, 0};
  for (int w = lane(); w < D; w += 64) {
    const uint32_t q = W->qcodes[w];
    uint32_t x[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) x[j] = j < cnt ? ((const uint32_t*)(H.codes + (size_t)id[j] * H.adim))[w] : 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (H.distance == D_L1) acc[j] += __builtin_amdgcn_sad_u8(q, x[j], 0u);
      else acc[j] = __builtin_amdgcn_udot4(q, x[j], acc[j], false);
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const uint32_t s = wave_sum_u(acc[j]);
    const float vo = j < cnt ? H.voff[id[j]] : 0.0f;
    out[j] = __fadd_rn(__fadd_rn(__fmul_rn(H.mult, (float)(int32_t)s), W->qoff), vo);
  }
}
// original-vector score (calculate_score, point_scorer.rs:133-174) of up to 4 pairs (a[j], b[j]) over the PRE-PROCESSED column
// (HNSWIndex::build normalises a cosine column first, hnsw.rs:150-157). Two forms:
//   exact (H.exact_scores, the deterministic build): lane j computes pair j alone, element by element in the reference's order
//     (Iterator::sum is a sequential f32 fold; every product / difference rounded on its own) — bit-identical scores;
//   wave (the concurrent build): the 64 lanes split the dimensions and the partial sums are combined by shuffles; for cosine the
//     raw dot product is divided by the two lengths afterwards. The reference's concurrent build is not reproducible either.
__device__ __forceinline__ void score_orig4(const HnswView& H, const uint32_t (&a)[4], const uint32_t (&b)[4], int cnt, float (&out)[4]) {
  if (H.exact_scores) {
    const int j = lane();
    float acc = 0.0f;
    if (j < cnt) {
      const uint32_t pa = j == 0 ? a[0] : j == 1 ? a[1] : j == 2 ? a[2] : a[3];
      const uint32_t pb = j == 0 ? b[0] : j == 1 ? b[1] : j == 2 ? b[2] : b[3];
      for (int i = 0; i < H.dim; ++i) {
        const float x = hn_value(H.raw, H.vlen, pa, H.dim, i), y = hn_value(H.raw, H.vlen, pb, H.dim, i);
        if (H.distance == D_DOT) acc = __fadd_rn(acc, __fmul_rn(x, y));
        else if (H.distance == D_L1) acc = __fadd_rn(acc, fabsf(__fsub_rn(x, y)));
        else { const float d = __fsub_rn(x, y); acc = __fadd_rn(acc, __fmul_rn(d, d)); }
      }
      if (H.distance != D_DOT) acc = -acc;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) out[q] = __shfl(acc, q, 64);
    return;
  }
  float acc[4] = {0, 0, 0, 0};
  for (int i = lane(); i < H.dim; i += 64) {
    float x[4], y[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      x[j] = j < cnt ? H.raw[(size_t)a[j] * H.dim + i] : 0.0f;
      y[j] = j < cnt ? H.raw[(size_t)b[j] * H.dim + i] : 0.0f;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (H.distance == D_DOT) acc[j] += x[j] * y[j];
      else if (H.distance == D_L1) acc[j] += fabsf(x[j] - y[j]);
      else { const float d = x[j] - y[j]; acc[j] += d * d; }
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float s = wave_sum_f(acc[j]);
    if (H.distance == D_DOT && H.vlen && j < cnt) {   // cosine: the column is normalised before the build
      const float la = H.vlen[a[j]], lb = H.vlen[b[j]];
      s = s / ((la == 0.0f ? 1.0f : la) * (lb == 0.0f ? 1.0f : lb));
    }
    out[j] = H.distance == D_DOT ? s : -s;
  }
}

template <bool QUANT>
__device__ __forceinline__ void score_points4(const HnswView& H, const WaveLds* W, uint32_t self, const uint32_t (&id)[4], int cnt, float (&out)[4]) {
  if (QUANT) score_quant4(H, W, id, cnt, out);
  else {
    const uint32_t a[4] = {self, self, self, self};
    score_orig4(H, a, id, cnt, out);
  }
}

// the k-th set bit (k = 0, 1, ...) of a 64-bit mask
__device__ __forceinline__ int nth_set(uint64_t m, int k) {
  for (int i = 0; i < k; ++i) m &= m - 1;
  return __ffsll((long long)m) - 1;
}

// _search_on_level (graph_layers.rs:72-108). QUANT: quantised scorer over the finished graph; else the build's scorer over
// the ready points (graph_layers_builder.rs:77-87)
template <bool QUANT, int VCAP = HN_VCAP>
__device__ void search_on_level(const HnswView& H, WaveLds* W, uint32_t* vis, uint32_t self, int level) {
  const int l = lane();
  for (;;) {
    if (l == 0) {
      SP c;
      W->cur.idx = 0xFFFFFFFFu;
      if (!(W->flag) && heap_pop(W->cand, &W->nc, &c)) {
        const float lb = W->nn >= W->ef ? W->nearest[0].score : -3.4028235e38f;   // lower_bound()
        if (!(c.score < lb)) W->cur = c;
      }
    }
    WAVE_SYNC();
    const uint32_t cidx = ((volatile SP*)&W->cur)->idx;
    if (cidx == 0xFFFFFFFFu) break;
    uint32_t* cnt_p;
    const uint32_t* links = list_ptr(H, cidx, level, &cnt_p);
    const int cnt = (int)ld32(cnt_p);
    uint32_t id = 0;
    bool ok = false;
    if (l < cnt) {
      id = ld32(links + l);
      ok = id < (uint32_t)H.n && (QUANT || ld32(H.ready + id) != 0) && !vis_check<VCAP>(vis, id);
    }
    const uint64_t mask = __ballot(ok);
    const int np = __popcll(mask);
    for (int b = 0; b < np; b += 4) {
      uint32_t ids[4] = {0, 0, 0, 0};
      const int c4 = np - b < 4 ? np - b : 4;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (j < c4) ids[j] = __shfl(id, nth_set(mask, b + j), 64);
      float sc[4];
      score_points4<QUANT>(H, W, self, ids, c4, sc);
      if (l == 0) {
        for (int j = 0; j < c4; ++j) {
          const SP sp = {ids[j], sc[j]};
          process_candidate(W, sp);
          vis_insert<VCAP>(vis, W, ids[j]);
        }
      }
      WAVE_SYNC();
    }
  }
}

// search_entry (graph_layers.rs:132-175): greedy, beam 1, levels top .. target + 1
template <bool QUANT>
__device__ SP search_entry(const HnswView& H, WaveLds* W, uint32_t self, uint32_t entry, int top_level, int target_level) {
  const int l = lane();
  SP cur;
  {
    const uint32_t ids[4] = {entry, 0, 0, 0};
    float sc[4];
    score_points4<QUANT>(H, W, self, ids, 1, sc);
    cur.idx = entry;
    cur.score = sc[0];
  }
  for (int level = top_level; level > target_level; --level) {
    bool changed = true;
    while (changed) {
      changed = false;
      uint32_t* cnt_p;
      const uint32_t* links = list_ptr(H, cur.idx, level, &cnt_p);
      const int cnt = (int)ld32(cnt_p);
      uint32_t id = 0;
      bool ok = false;
      if (l < cnt) {
        id = ld32(links + l);
        ok = id < (uint32_t)H.n && (QUANT || ld32(H.ready + id) != 0);
      }
      const uint64_t mask = __ballot(ok);
      const int np = __popcll(mask);
      for (int b = 0; b < np; b += 4) {
        uint32_t ids[4] = {0, 0, 0, 0};
        const int c4 = np - b < 4 ? np - b : 4;
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (j < c4) ids[j] = __shfl(id, nth_set(mask, b + j), 64);
        float sc[4];
        score_points4<QUANT>(H, W, self, ids, c4, sc);
        for (int j = 0; j < c4; ++j)
          if (sc[j] > cur.score) { changed = true; cur.idx = ids[j]; cur.score = sc[j]; }
      }
    }
  }
  return cur;
}

// ---------------------------------------------------------------------------------------------------------------------
// search kernel: one wave per query (persistent waves take queries from a counter)
// ---------------------------------------------------------------------------------------------------------------------
struct SearchArgs {
  const float* queries;     // [nq][dim] raw
  int nq, limit;
  uint32_t* out_ids;        // [nq][limit]
  float* out_dist;
  uint32_t* vis;            // [gridDim.x][HN_VCAP]
  unsigned int* next;       // query counter
  unsigned int* err;        // |= flags
};

__device__ __forceinline__ float hn_postprocess(int distance, float s) {   // hnsw.rs:317-343
  if (distance == D_L1) return fabsf(s);
  if (distance == D_L2) return hn_sqrt(fabsf(s));
  return fabsf(__fsub_rn(1.0f, s));
}

// encode_query (:317-366) into W->qcodes / W->qoff; the query is pre-processed first (preprocess_query, hnsw.rs:307-312)
__device__ void wave_encode_query(const HnswView& H, WaveLds* W, const float* q) {
  const int l = lane();
  float qlen = 0.0f;   // 0 = leave as it is
  if (H.distance == D_DOT) {
    float len = 0.0f;
    for (int i = 0; i < H.dim; ++i) len = __fadd_rn(len, __fmul_rn(q[i], q[i]));   // sequential, every lane the same
    const bool keep = len < 1.1920929e-7f || fabsf(__fsub_rn(len, 1.0f)) <= 1.0e-6f;
    qlen = keep ? 0.0f : hn_sqrt(len);
  }
  uint8_t* qc = (uint8_t*)W->qcodes;
  uint32_t s1 = 0, s2 = 0;
  const float placeholder = H.distance == D_DOT ? 0.0f : H.offset;
  for (int i = l; i < H.adim; i += 64) {
    float x = placeholder;
    if (i < H.dim) { x = q[i]; if (qlen != 0.0f) x = __fdiv_rn(x, qlen); }
    const uint8_t c = hn_f32_to_u8(x, H.alpha, H.offset);
    qc[i] = c;
    s1 += c;
    s2 += (uint32_t)c * c;
  }
  s1 = wave_sum_u(s1);
  s2 = wave_sum_u(s2);
  const float off = hn_offset_term(H.distance, H.adim, H.alpha, H.offset, s1, s2, false);
  if (l == 0) W->qoff = H.distance == D_DOT ? off : -off;
  WAVE_SYNC();
}

// SMALL (ef <= 64): the candidate heap holds HN_SMALL_CCAP entries and the visited table lives in LDS (HN_SMALL_VCAP slots) — the
// same LDS footprint per wave as the general kernel, whose visited table is in global memory. A walk that outgrows either sets
// flag bits 2 / 3 (instead of 0 / 1) and the host runs the whole batch again through the general kernel.
template <bool SMALL>
__global__ __launch_bounds__(64) void hn_search_kernel(HnswView H, SearchArgs A) {
  constexpr int VCAP = SMALL ? HN_SMALL_VCAP : HN_VCAP;
  constexpr int CCAP = SMALL ? HN_SMALL_CCAP : HN_CCAP;
  __shared__ __attribute__((aligned(16))) uint8_t wraw[offsetof(WaveLds, cand) + sizeof(SP) * CCAP];
  __shared__ uint32_t lvis[SMALL ? HN_SMALL_VCAP : 1];
  WaveLds* W = (WaveLds*)wraw;
  uint32_t* vis = SMALL ? lvis : A.vis + (size_t)blockIdx.x * HN_VCAP;
  const int l = lane();
  const unsigned long long ent = *H.entry;
  for (;;) {
    unsigned int qi = 0;
    if (l == 0) qi = atomicAdd(A.next, 1u);
    qi = __shfl(qi, 0, 64);
    if (qi >= (unsigned)A.nq) break;
    uint32_t* oid = A.out_ids + (size_t)qi * A.limit;
    float* od = A.out_dist + (size_t)qi * A.limit;
    if (ent == 0) {   // empty index
      for (int i = l; i < A.limit; i += 64) { oid[i] = 0xFFFFFFFFu; od[i] = __uint_as_float(0x7FC00000u); }
      continue;
    }
    const uint32_t entry = ~(uint32_t)(ent & 0xFFFFFFFFu);
    const int entry_level = (int)(ent >> 32) - 1;
    vis_clear<VCAP>(vis);
    if (l == 0) { W->nn = 0; W->nc = 0; W->flag = 0; W->vis_count = 0; W->ccap = CCAP; W->ef = A.limit * 4 > A.limit ? A.limit * 4 : A.limit; }
    WAVE_SYNC();
    wave_encode_query(H, W, A.queries + (size_t)qi * H.dim);
    const SP zero = search_entry<true>(H, W, 0, entry, entry_level, 0);
    if (l == 0) {
      vis_insert<VCAP>(vis, W, zero.idx);
      SP dummy;
      flpq_push(W->nearest, &W->nn, W->ef, zero, &dummy);   // SearchContext::new
      heap_push(W->cand, &W->nc, zero);
    }
    WAVE_SYNC();
    search_on_level<true, VCAP>(H, W, vis, 0, 0);
    if (l == 0) {
      flpq_into_sorted(W->nearest, W->nn);
      if (W->flag) atomicOr(A.err, (unsigned)(SMALL ? W->flag << 2 : W->flag));
    }
    WAVE_SYNC();
    const int nn = ((volatile WaveLds*)W)->nn;
    for (int i = l; i < A.limit; i += 64) {
      if (i < nn) { oid[i] = W->nearest[i].idx; od[i] = hn_postprocess(H.distance, W->nearest[i].score); }
      else { oid[i] = 0xFFFFFFFFu; od[i] = __uint_as_float(0x7FC00000u); }
    }
    WAVE_SYNC();
  }
}

// generate_scores (hnsw.rs:120-140): one workgroup per (query, slab of rows), a quarter wave per row
__global__ __launch_bounds__(256) void hn_scores_kernel(HnswView H, const float* queries, int nq, float* out) {
  __shared__ WaveLds Wm;   // only qcodes / qoff are used
  WaveLds* W = &Wm;
  const int qi = blockIdx.y;
  if (threadIdx.x < 64) wave_encode_query(H, W, queries + (size_t)qi * H.dim);
  __syncthreads();
  const int D = H.adim >> 2;
  const int sub = threadIdx.x & 15;
  for (int64_t v = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 4; v < H.n; v += (int64_t)gridDim.x * 16) {
    uint32_t acc = 0;
    const uint32_t* row = (const uint32_t*)(H.codes + (size_t)v * H.adim);
    for (int w = sub; w < D; w += 16) {
      if (H.distance == D_L1) acc += __builtin_amdgcn_sad_u8(W->qcodes[w], row[w], 0u);
      else acc = __builtin_amdgcn_udot4(W->qcodes[w], row[w], acc, false);
    }
    for (int off = 8; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if (sub == 0) {
      const float s = __fadd_rn(__fadd_rn(__fmul_rn(H.mult, (float)(int32_t)acc), W->qoff), H.voff[v]);
      out[(size_t)qi * H.n + v] = hn_postprocess(H.distance, s);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// build: one wave per new point (link_new_point, graph_layers_builder.rs:343-520)
// ---------------------------------------------------------------------------------------------------------------------
struct BuildArgs {
  int64_t p_begin, p_end;
  int ef_construct;
  int sequential;           // 1: ONE wave links [p_begin, p_end) one after the other (the reference's first 256 points)
  uint32_t* vis;            // [gridDim.x][HN_VCAP]
  unsigned int* next;       // point counter (relative to p_begin)
  unsigned int* err;
};

// select_candidate_with_heuristic_from_sorted (:300-327): cands (LDS) in descending score order; returns the count, the ids in sel
__device__ int select_heuristic(const HnswView& H, const SP* cands, int nc, int m, uint32_t* sel) {
  int k = 0;
  for (int i = 0; i < nc && k < m; ++i) {
    const SP c = cands[i];
    bool good = true;
    for (int b = 0; b < k && good; b += 4) {
      const int c4 = k - b < 4 ? k - b : 4;
      uint32_t a[4] = {c.idx, c.idx, c.idx, c.idx}, o[4] = {0, 0, 0, 0};
      for (int j = 0; j < c4; ++j) o[j] = sel[b + j];
      float sc[4];
      score_orig4(H, a, o, c4, sc);
      for (int j = 0; j < c4; ++j)
        if (sc[j] > c.score) { good = false; break; }
    }
    if (good) {
      if (lane() == 0) sel[k] = c.idx;
      WAVE_SYNC();
      ++k;
    }
  }
  return k;
}

struct BuildLds {
  WaveLds W;
  uint32_t sel[HN_MAX_M0];
  SP tmp[HN_MAX_M0 + 4];
  uint32_t sel2[HN_MAX_M0];
};

__device__ void link_point(const HnswView& H, BuildLds* B, uint32_t* vis, const BuildArgs& A, uint32_t p) {
  WaveLds* W = &B->W;
  const int l = lane();
  const int level = H.level[p];
  const unsigned long long ent = __hip_atomic_load(H.entry, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (ent != 0) {
    const uint32_t entry = ~(uint32_t)(ent & 0xFFFFFFFFu);
    const int entry_level = (int)(ent >> 32) - 1;
    if (l == 0) { W->flag = 0; W->ef = A.ef_construct; W->ccap = HN_CCAP; }
    WAVE_SYNC();
    SP level_entry;
    if (entry_level > level) level_entry = search_entry<false>(H, W, p, entry, entry_level, level);
    else {
      const uint32_t a[4] = {p, 0, 0, 0}, b[4] = {entry, 0, 0, 0};
      float sc[4];
      score_orig4(H, a, b, 1, sc);
      level_entry.idx = entry;
      level_entry.score = sc[0];
    }
    const int linking_level = level < entry_level ? level : entry_level;
    for (int cl = linking_level; cl >= 0; --cl) {
      // link_new_point_on_level (:418-462)
      vis_clear<HN_VCAP>(vis);
      if (l == 0) {
        W->nn = 0; W->nc = 0; W->vis_count = 0;
        vis_insert<HN_VCAP>(vis, W, level_entry.idx);
        SP dummy;
        flpq_push(W->nearest, &W->nn, W->ef, level_entry, &dummy);
        heap_push(W->cand, &W->nc, level_entry);
      }
      WAVE_SYNC();
      search_on_level<false>(H, W, vis, p, cl);
      if (l == 0) {
        // nearest.iter_unsorted().max(): the last of equal maxima
        SP best = W->nearest[0];
        for (int i = 1; i < W->nn; ++i)
          if (sp_le(best.score, W->nearest[i].score)) best = W->nearest[i];
        W->cur = best;
        flpq_into_sorted(W->nearest, W->nn);
      }
      WAVE_SYNC();
      level_entry.idx = ((volatile SP*)&W->cur)->idx;
      level_entry.score = ((volatile SP*)&W->cur)->score;
      const int nn = ((volatile WaveLds*)W)->nn;
      // link_with_heuristic (:464-520); a new point has no links yet
      const int level_m = cl == 0 ? H.m0 : H.m;
      const int ns = select_heuristic(H, W->nearest, nn, level_m, B->sel);
      uint32_t* mycnt;
      uint32_t* mine = list_ptr(H, p, cl, &mycnt);
      if (l < ns) st32(mine + l, B->sel[l]);
      PUBLISH_RELEASE();
      if (l == 0) st32(mycnt, (uint32_t)ns);
      for (int k = 0; k < ns; ++k) {
        const uint32_t other = B->sel[k];
        if (l == 0) {
          while (atomicCAS(&H.lock[other], 0u, 1u) != 0u) __builtin_amdgcn_s_sleep(2);
        }
        WAVE_SYNC();
        uint32_t* ocnt;
        uint32_t* ol = list_ptr(H, other, cl, &ocnt);
        const int on = (int)ld32(ocnt);
        if (on < level_m) {
          if (l == 0) {
            st32(ol + on, p);
            PUBLISH_RELEASE();
            st32(ocnt, (uint32_t)(on + 1));
          }
        } else {
          // candidates = {p} + other's links, scored against `other`, descending, through the heuristic
          const int nc = level_m + 1;
          for (int b = 0; b < nc; b += 4) {
            const int c4 = nc - b < 4 ? nc - b : 4;
            uint32_t a[4] = {0, 0, 0, 0}, o[4] = {other, other, other, other};
            for (int j = 0; j < c4; ++j) a[j] = (b + j == 0) ? p : ld32(ol + (b + j - 1));
            float sc[4];
            score_orig4(H, a, o, c4, sc);
            // candidates.push(..) in the reference's order: the new point first, then other's links (BinaryHeap::push = sift_up)
            if (l == 0)
              for (int j = 0; j < c4; ++j) { SP v; v.idx = a[j]; v.score = sc[j]; int len = b + j; heap_push(B->tmp, &len, v); }
            WAVE_SYNC();
          }
          // candidates.into_sorted_vec().into_iter().rev(): std's heap sort, then descending — equal scores (duplicate vectors,
          // a zero vector under cosine) come out in ITS order, not in insertion order
          if (l == 0) {
            heap_into_sorted(B->tmp, nc);
            for (int i = 0, j = nc - 1; i < j; ++i, --j) { const SP t = B->tmp[i]; B->tmp[i] = B->tmp[j]; B->tmp[j] = t; }
          }
          WAVE_SYNC();
          const int n2 = select_heuristic(H, B->tmp, nc, level_m, B->sel2);
          if (l < n2) st32(ol + l, B->sel2[l]);
          PUBLISH_RELEASE();
          if (l == 0) st32(ocnt, (uint32_t)n2);
        }
        PUBLISH_RELEASE();
        if (l == 0) __hip_atomic_store(&H.lock[other], 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        WAVE_SYNC();
      }
    }
    if (l == 0 && W->flag) atomicOr(A.err, (unsigned)W->flag);
  }
  PUBLISH_RELEASE();
  if (l == 0) {
    st32(H.ready + p, 1u);
    // entry_points.new_point (entry_points.rs:56-103): replaced only by a strictly higher level (the first one to get there
    // among concurrent inserts)
    const unsigned long long mine = ((unsigned long long)(level + 1) << 32) | (unsigned long long)(~p);
    unsigned long long cur = __hip_atomic_load(H.entry, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (cur == 0 || (cur >> 32) < (mine >> 32)) {
      const unsigned long long old = atomicCAS(H.entry, cur, mine);
      if (old == cur) break;
      cur = old;
    }
  }
  WAVE_SYNC();
}

__global__ __launch_bounds__(64) void hn_build_kernel(HnswView H, BuildArgs A) {
  __shared__ BuildLds B;
  uint32_t* vis = A.vis + (size_t)blockIdx.x * HN_VCAP;
  const int l = lane();
  if (A.sequential) {
    if (blockIdx.x != 0) return;
    for (int64_t p = A.p_begin; p < A.p_end; ++p) link_point(H, &B, vis, A, (uint32_t)p);
    return;
  }
  for (;;) {
    unsigned int i = 0;
    if (l == 0) i = atomicAdd(A.next, 1u);
    i = __shfl(i, 0, 64);
    if ((int64_t)i >= A.p_end - A.p_begin) break;
    link_point(H, &B, vis, A, (uint32_t)(A.p_begin + i));
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// host
// ---------------------------------------------------------------------------------------------------------------------
int dist_code(int32_t distance) {
  switch (distance) {
    case DBHIP_VEC_COSINE: return D_DOT;
    case DBHIP_VEC_L1: return D_L1;
    case DBHIP_VEC_L2: return D_L2;
    default: return -1;
  }
}

int32_t own(dbhip_hnsw_impl* h, size_t bytes, void** out) {
  int32_t rc = dbhip_alloc(bytes ? bytes : 16, out);
  if (rc) return rc;
  h->owned[h->n_owned++] = *out;
  return DBHIP_OK;
}

uint64_t splitmix64(uint64_t* s) {
  uint64_t z = (*s += 0x9E3779B97F4A7C15ULL);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}

void destroy_impl(dbhip_hnsw_impl* h) {
  if (!h) return;
  for (int i = 0; i < h->n_owned; ++i) (void)dbhip_free(h->owned[i]);
  delete h;
}

// owns a half-built index: any early return (DBHIP_CHECK, DBHIP_REQUIRE, a validation error) frees the device buffers
struct ImplGuard {
  dbhip_hnsw_impl* h;
  explicit ImplGuard(dbhip_hnsw_impl* p) : h(p) {}
  ~ImplGuard() { destroy_impl(h); }
  dbhip_hnsw_impl* release() { dbhip_hnsw_impl* p = h; h = nullptr; return p; }
  ImplGuard(const ImplGuard&) = delete;
  ImplGuard& operator=(const ImplGuard&) = delete;
};

// quantiser + graph storage for `levels`
// `encoded` (HNSWIndex::open): the quantised vectors in the reference's storage layout with their metadata, instead of `vectors`
struct EncodedIn { const uint8_t* data; float alpha, offset, multiplier; };
int32_t create_common(const float* vectors, int64_t n, int32_t dim, int32_t distance, int32_t m, const int32_t* levels,
                      hipStream_t s, dbhip_hnsw_impl** out, const EncodedIn* encoded = nullptr) {
  const int dc = dist_code(distance);
  if (dc < 0) { set_error("dbhip_hnsw: distance must be cosine, l1 or l2 (the reference's index option)"); return DBHIP_ERR_UNSUPPORTED; }
  DBHIP_REQUIRE(n >= 0 && n < 0x7FFFFFF0LL && dim > 0 && m >= 1 && 2 * m <= HN_MAX_M0, "dbhip_hnsw: bad n / dim / m (m <= 32)");
  DBHIP_REQUIRE(n == 0 || vectors || (encoded && encoded->data), "dbhip_hnsw: NULL vectors");
  const int adim = dim + (HN_ALIGN - dim % HN_ALIGN) % HN_ALIGN;
  if (adim > HN_MAX_ADIM) { set_error("dbhip_hnsw: dim > %d", HN_MAX_ADIM); return DBHIP_ERR_UNSUPPORTED; }
  dbhip_hnsw_impl* h = new (std::nothrow) dbhip_hnsw_impl();
  if (!h) return DBHIP_ERR_HIP;
  ImplGuard guard(h);
  h->n_owned = 0;
  h->small_search_off = 0;
  HnswView& V = h->v;
  memset(&V, 0, sizeof(V));
  V.n = n; V.dim = dim; V.adim = adim; V.distance = dc; V.m = m; V.m0 = 2 * m;
  V.raw = vectors;
  h->level_host.assign(levels, levels + n);
  h->ufirst_host.assign((size_t)n, -1);
  int64_t nu = 0;
  for (int64_t i = 0; i < n; ++i) {
    if (levels[i] < 0 || levels[i] > 60) { set_error("dbhip_hnsw: bad level"); return DBHIP_ERR_INVALID; }
    if (levels[i] > 0) { h->ufirst_host[i] = nu; nu += levels[i]; }
  }
  h->n_upper = nu;
  int32_t rc;
  void *codes, *voff, *vlen = nullptr, *l0, *c0, *lu, *cu, *uf, *lv, *rd, *lk, *en;
  const size_t nn = (size_t)(n > 0 ? n : 1);
  if ((rc = own(h, nn * adim, &codes)) || (rc = own(h, nn * 4, &voff)) || (rc = own(h, nn * V.m0 * 4, &l0)) || (rc = own(h, nn * 4, &c0)) ||
      (rc = own(h, (size_t)(nu > 0 ? nu : 1) * m * 4, &lu)) || (rc = own(h, (size_t)(nu > 0 ? nu : 1) * 4, &cu)) || (rc = own(h, nn * 8, &uf)) ||
      (rc = own(h, nn * 4, &lv)) || (rc = own(h, nn * 4, &rd)) || (rc = own(h, nn * 4, &lk)) || (rc = own(h, 16, &en))) return rc;
  if (dc == D_DOT && !encoded && (rc = own(h, nn * 4, &vlen))) return rc;
  V.codes = (uint8_t*)codes; V.voff = (float*)voff; V.vlen = (float*)vlen; V.links0 = (uint32_t*)l0; V.cnt0 = (uint32_t*)c0;
  V.linksu = (uint32_t*)lu; V.cntu = (uint32_t*)cu; V.ufirst = (int64_t*)uf; V.level = (int32_t*)lv; V.ready = (uint32_t*)rd;
  V.lock = (uint32_t*)lk; V.entry = (unsigned long long*)en;
  DBHIP_CHECK(hipMemsetAsync(c0, 0, nn * 4, s));
  DBHIP_CHECK(hipMemsetAsync(l0, 0, nn * (size_t)V.m0 * 4, s));  // a reader that races a publish sees 0 (a valid id), never garbage
  DBHIP_CHECK(hipMemsetAsync(cu, 0, (size_t)(nu > 0 ? nu : 1) * 4, s));
  DBHIP_CHECK(hipMemsetAsync(rd, 0, nn * 4, s));
  DBHIP_CHECK(hipMemsetAsync(lk, 0, nn * 4, s));
  DBHIP_CHECK(hipMemsetAsync(en, 0, 16, s));
  if (n > 0) {
    DBHIP_CHECK(hipMemcpyAsync(lv, h->level_host.data(), (size_t)n * 4, hipMemcpyHostToDevice, s));
    DBHIP_CHECK(hipMemcpyAsync(uf, h->ufirst_host.data(), (size_t)n * 8, hipMemcpyHostToDevice, s));
    if (encoded) {   // EncodedVectorsU8::load (encoded_vectors_u8.rs:312-325): the stored codes and metadata as they are
      V.alpha = encoded->alpha; V.offset = encoded->offset; V.mult = encoded->multiplier;
      hipLaunchKernelGGL(hn_unlayout_kernel, dim3(grid_for(n * (adim + 4), 256)), dim3(256), 0, s, encoded->data, n, adim, (uint8_t*)codes, (float*)voff);
      DBHIP_LAUNCH_CHECK();
      *out = guard.release();
      return DBHIP_OK;
    }
    // ---- EncodedVectorsU8::encode over the pre-processed vectors (hnsw.rs:150-157,262-283) ----
    if (dc == D_DOT) hipLaunchKernelGGL(hn_length_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, s, vectors, n, dim, (float*)vlen);
    uint32_t* mm = (uint32_t*)scratch(8, 3, s);
    const uint32_t init[2] = {0xFFFFFFFFu, 0u};
    DBHIP_CHECK(hipMemcpyAsync(mm, init, 8, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(hn_minmax_kernel, dim3(grid_for(n * dim, 256)), dim3(256), 0, s, vectors, (const float*)vlen, n, dim, mm);
    DBHIP_LAUNCH_CHECK();
    uint32_t got[2];
    DBHIP_CHECK(hipMemcpyAsync(got, mm, 8, hipMemcpyDeviceToHost, s));
    DBHIP_CHECK(hipStreamSynchronize(s));
    auto unorder = [](uint32_t o) { const uint32_t b = (o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o; float f; memcpy(&f, &b, 4); return f; };
    // (no finite value at all: the reference's fold leaves (f32::MAX, f32::MIN))
    const float mn = got[0] == 0xFFFFFFFFu ? 3.4028235e38f : unorder(got[0]);
    const float mx = got[1] == 0u ? -3.4028235e38f : unorder(got[1]);
    // alpha_offset_from_min_max (:237-241) and the multiplier (:150-159), in f32 as the reference computes them
    volatile float alpha = (mx - mn) / 127.0f;
    V.alpha = alpha;
    V.offset = mn;
    volatile float a2 = V.alpha * V.alpha;
    float mult = dc == D_DOT ? (float)a2 : dc == D_L1 ? V.alpha : (float)(volatile float)(-2.0f * V.alpha) * V.alpha;
    V.mult = dc == D_DOT ? mult : -mult;
    hipLaunchKernelGGL(hn_encode_kernel, dim3((unsigned)ceil_div(n * 16, 256)), dim3(256), 0, s, vectors, (const float*)vlen, n, dim, adim, dc,
                       V.alpha, V.offset, (uint8_t*)codes, (float*)voff);
    DBHIP_LAUNCH_CHECK();
  }
  *out = guard.release();
  return DBHIP_OK;
}

int search_grid(int64_t work) {
  int g = 2048;   // persistent waves: 8 per CU
  if (work < g) g = (int)(work > 0 ? work : 1);
  return g;
}

}  // namespace

struct dbhip_hnsw { dbhip_hnsw_impl impl; };

namespace {
// links every point of a fresh index: `deterministic` = ONE wave links the points one after the other with the exact scorer
int32_t build_graph(const float* vectors_dev, int64_t n, int32_t dim, int32_t distance, int32_t m, int32_t ef_construct,
                    const int32_t* levels, bool deterministic, dbhip_hnsw** out, hipStream_t s) {
  dbhip_hnsw_impl* h = nullptr;
  int32_t rc = create_common(vectors_dev, n, dim, distance, m, levels, s, &h);
  if (rc) return rc;
  ImplGuard guard(h);
  h->ef_construct = ef_construct;
  h->v.exact_scores = deterministic ? 1 : 0;
  if (n > 0) {
    const int grid = search_grid(n);
    uint32_t* vis = nullptr;
    unsigned int* ctl = nullptr;
    if ((rc = dbhip_alloc((size_t)grid * HN_VCAP * 4, (void**)&vis)) || (rc = dbhip_alloc(16, (void**)&ctl))) { if (vis) dbhip_free(vis); return rc; }
    auto fail = [&](int32_t code) { dbhip_free(vis); dbhip_free(ctl); return code; };
    if (hipMemsetAsync(ctl, 0, 16, s) != hipSuccess) return fail(DBHIP_ERR_HIP);
    BuildArgs A;
    A.ef_construct = ef_construct; A.vis = vis; A.next = ctl; A.err = ctl + 1;
    // the first 256 points one after the other (hnsw.rs:196-235), then concurrently in growing launches so that early waves
    // are not blind to each other: [256, 512), [512, 1024), ... doubling up to 1 M points per launch
    int64_t done = 0;
    {
      A.p_begin = 0; A.p_end = (n < HN_SEQ || deterministic) ? n : HN_SEQ; A.sequential = 1;
      hipLaunchKernelGGL(hn_build_kernel, dim3(1), dim3(64), 0, s, h->v, A);
      done = A.p_end;
    }
    while (done < n) {
      int64_t step = done;
      if (step > (1 << 20)) step = 1 << 20;
      A.p_begin = done; A.p_end = done + step < n ? done + step : n; A.sequential = 0;
      if (hipMemsetAsync(ctl, 0, 4, s) != hipSuccess) return fail(DBHIP_ERR_HIP);
      const int g = search_grid(A.p_end - A.p_begin);
      hipLaunchKernelGGL(hn_build_kernel, dim3(g), dim3(64), 0, s, h->v, A);
      done = A.p_end;
    }
    if (hipGetLastError() != hipSuccess) return fail(DBHIP_ERR_HIP);
    unsigned int hc[2] = {0, 0};
    if (hipMemcpyAsync(hc, ctl, 8, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) return fail(DBHIP_ERR_HIP);
    dbhip_free(vis); dbhip_free(ctl);
    if (hc[1]) {
      set_error("dbhip_hnsw_build: a walk outgrew its candidate heap / visited table (flags %u)", hc[1]);
      return DBHIP_ERR_CAPACITY;
    }
  }
  h->v.raw = nullptr;   // the caller's vectors are not kept (search uses the codes)
  h->v.exact_scores = 0;
  *out = (dbhip_hnsw*)guard.release();
  return DBHIP_OK;
}
}  // namespace

extern "C" {

int32_t dbhip_hnsw_build(const float* vectors_dev, int64_t n, int32_t dim, int32_t distance, int32_t m, int32_t ef_construct,
                         uint64_t seed, dbhip_hnsw** out, void* stream) {
  DBHIP_REQUIRE(out && ef_construct >= 1 && ef_construct <= HN_MAX_EF, "dbhip_hnsw_build: bad argument (ef_construct <= 256)");
  hipStream_t s = resolve_stream(stream);
  // get_random_layer (graph_layers_builder.rs:246-255): round(-ln(u) * 1 / ln(max(m, 2))), u uniform in [0, 1)
  std::vector<int32_t> levels((size_t)(n > 0 ? n : 0));
  const double level_factor = 1.0 / log((double)(m > 2 ? m : 2));
  uint64_t st = seed;
  for (int64_t i = 0; i < n; ++i) {
    double u = (double)(splitmix64(&st) >> 11) * (1.0 / 9007199254740992.0);
    if (u <= 0.0) u = 1.0 / 9007199254740992.0;
    levels[(size_t)i] = (int32_t)llround(-log(u) * level_factor);
  }
  return build_graph(vectors_dev, n, dim, distance, m, ef_construct, levels.data(), false, out, s);
}

int32_t dbhip_hnsw_build_sequential(const float* vectors_dev, int64_t n, int32_t dim, int32_t distance, int32_t m, int32_t ef_construct,
                                    const int32_t* levels_host, dbhip_hnsw** out, void* stream) {
  DBHIP_REQUIRE(out && ef_construct >= 1 && ef_construct <= HN_MAX_EF && (levels_host || n == 0), "dbhip_hnsw_build_sequential: bad argument (ef_construct <= 256, levels given)");
  return build_graph(vectors_dev, n, dim, distance, m, ef_construct, levels_host, true, out, resolve_stream(stream));
}

}  // extern "C"

namespace {
// copies a given graph into a fresh index (lists in point-major, level-minor order)
int32_t install_graph(dbhip_hnsw_impl* h, int64_t n, int32_t m, const int32_t* levels_host, const uint32_t* links_host, const int32_t* nlinks_host,
                      uint32_t entry_point, int32_t entry_level, hipStream_t s) {
  const int m0 = 2 * m;
  std::vector<uint32_t> l0((size_t)n * m0, 0), c0((size_t)n, 0), lu((size_t)h->n_upper * m, 0), cu((size_t)h->n_upper, 0), rd((size_t)n, 1);
  int64_t list = 0, off = 0;
  for (int64_t p = 0; p < n; ++p)
    for (int lv = 0; lv <= levels_host[p]; ++lv, ++list) {
      const int c = nlinks_host[list];
      if (c < 0 || c > (lv == 0 ? m0 : m)) { set_error("dbhip_hnsw: a list longer than m / m0"); return DBHIP_ERR_INVALID; }
      for (int i = 0; i < c; ++i) {
        const uint32_t x = links_host[off + i];
        if (x >= (uint64_t)n || levels_host[x] < lv) { set_error("dbhip_hnsw: link out of range or to a point below the list's level"); return DBHIP_ERR_INVALID; }
        if (lv == 0) l0[(size_t)p * m0 + i] = x;
        else lu[(size_t)(h->ufirst_host[p] + lv - 1) * m + i] = x;
      }
      if (lv == 0) c0[p] = c; else cu[h->ufirst_host[p] + lv - 1] = c;
      off += c;
    }
  if (n > 0) {
    DBHIP_CHECK(hipMemcpyAsync(h->v.links0, l0.data(), l0.size() * 4, hipMemcpyHostToDevice, s));
    DBHIP_CHECK(hipMemcpyAsync(h->v.cnt0, c0.data(), c0.size() * 4, hipMemcpyHostToDevice, s));
    if (h->n_upper) {
      DBHIP_CHECK(hipMemcpyAsync(h->v.linksu, lu.data(), lu.size() * 4, hipMemcpyHostToDevice, s));
      DBHIP_CHECK(hipMemcpyAsync(h->v.cntu, cu.data(), cu.size() * 4, hipMemcpyHostToDevice, s));
    }
    DBHIP_CHECK(hipMemcpyAsync(h->v.ready, rd.data(), rd.size() * 4, hipMemcpyHostToDevice, s));
    if (entry_point >= (uint64_t)n || entry_level < 0 || entry_level > levels_host[entry_point]) { set_error("dbhip_hnsw: bad entry point"); return DBHIP_ERR_INVALID; }
    const unsigned long long ent = ((unsigned long long)(entry_level + 1) << 32) | (unsigned long long)(~entry_point);
    DBHIP_CHECK(hipMemcpyAsync(h->v.entry, &ent, 8, hipMemcpyHostToDevice, s));
    DBHIP_CHECK(hipStreamSynchronize(s));
  }
  h->v.raw = nullptr;
  h->ef_construct = 0;
  return DBHIP_OK;
}
}  // namespace

extern "C" {

int32_t dbhip_hnsw_from_graph(const float* vectors_dev, int64_t n, int32_t dim, int32_t distance, int32_t m,
                              const int32_t* levels_host, const uint32_t* links_host, const int32_t* nlinks_host,
                              uint32_t entry_point, int32_t entry_level, dbhip_hnsw** out, void* stream) {
  DBHIP_REQUIRE(out && (n == 0 || (levels_host && links_host && nlinks_host)), "dbhip_hnsw_from_graph: bad argument");
  hipStream_t s = resolve_stream(stream);
  dbhip_hnsw_impl* h = nullptr;
  int32_t rc = create_common(vectors_dev, n, dim, distance, m, levels_host, s, &h);
  if (rc) return rc;
  ImplGuard guard(h);
  if ((rc = install_graph(h, n, m, levels_host, links_host, nlinks_host, entry_point, entry_level, s))) return rc;
  *out = (dbhip_hnsw*)guard.release();
  return DBHIP_OK;
}

int32_t dbhip_hnsw_open(const uint8_t* encoded_dev, float alpha, float offset, float multiplier, int64_t n, int32_t dim, int32_t distance,
                        int32_t m, const int32_t* levels_host, const uint32_t* links_host, const int32_t* nlinks_host, uint32_t entry_point,
                        int32_t entry_level, dbhip_hnsw** out, void* stream) {
  DBHIP_REQUIRE(out && (n == 0 || (encoded_dev && levels_host && links_host && nlinks_host)), "dbhip_hnsw_open: bad argument");
  hipStream_t s = resolve_stream(stream);
  dbhip_hnsw_impl* h = nullptr;
  const EncodedIn enc{encoded_dev, alpha, offset, multiplier};
  int32_t rc = create_common(nullptr, n, dim, distance, m, levels_host, s, &h, &enc);
  if (rc) return rc;
  ImplGuard guard(h);
  if ((rc = install_graph(h, n, m, levels_host, links_host, nlinks_host, entry_point, entry_level, s))) return rc;
  *out = (dbhip_hnsw*)guard.release();
  return DBHIP_OK;
}

int32_t dbhip_hnsw_export_graph(dbhip_hnsw* hh, int32_t* levels_host, uint32_t* links_host, int32_t* nlinks_host,
                                int64_t* out_n_lists_host, uint32_t* out_entry_point_host, int32_t* out_entry_level_host, void* stream) {
  DBHIP_REQUIRE(hh, "dbhip_hnsw_export_graph: NULL index");
  dbhip_hnsw_impl* h = (dbhip_hnsw_impl*)hh;
  hipStream_t s = resolve_stream(stream);
  const int64_t n = h->v.n;
  const int m = h->v.m, m0 = h->v.m0;
  std::vector<uint32_t> l0((size_t)n * m0), c0((size_t)n), lu((size_t)h->n_upper * m), cu((size_t)h->n_upper);
  unsigned long long ent = 0;
  if (n > 0) {
    DBHIP_CHECK(hipMemcpyAsync(l0.data(), h->v.links0, l0.size() * 4, hipMemcpyDeviceToHost, s));
    DBHIP_CHECK(hipMemcpyAsync(c0.data(), h->v.cnt0, c0.size() * 4, hipMemcpyDeviceToHost, s));
    if (h->n_upper) {
      DBHIP_CHECK(hipMemcpyAsync(lu.data(), h->v.linksu, lu.size() * 4, hipMemcpyDeviceToHost, s));
      DBHIP_CHECK(hipMemcpyAsync(cu.data(), h->v.cntu, cu.size() * 4, hipMemcpyDeviceToHost, s));
    }
  }
  DBHIP_CHECK(hipMemcpyAsync(&ent, h->v.entry, 8, hipMemcpyDeviceToHost, s));
  DBHIP_CHECK(hipStreamSynchronize(s));
  int64_t list = 0, off = 0;
  for (int64_t p = 0; p < n; ++p) {
    if (levels_host) levels_host[p] = h->level_host[p];
    for (int lv = 0; lv <= h->level_host[p]; ++lv, ++list) {
      const int c = lv == 0 ? (int)c0[p] : (int)cu[h->ufirst_host[p] + lv - 1];
      if (nlinks_host) nlinks_host[list] = c;
      if (links_host)
        for (int i = 0; i < c; ++i) links_host[off + i] = lv == 0 ? l0[(size_t)p * m0 + i] : lu[(size_t)(h->ufirst_host[p] + lv - 1) * m + i];
      off += c;
    }
  }
  if (out_n_lists_host) *out_n_lists_host = list;
  if (out_entry_point_host) *out_entry_point_host = ent ? ~(uint32_t)(ent & 0xFFFFFFFFu) : 0xFFFFFFFFu;
  if (out_entry_level_host) *out_entry_level_host = ent ? (int32_t)(ent >> 32) - 1 : -1;
  return DBHIP_OK;
}

int32_t dbhip_hnsw_search(dbhip_hnsw* hh, const float* queries_dev, int32_t nq, int32_t limit, uint32_t* out_ids_dev,
                          float* out_dist_dev, void* stream) {
  DBHIP_REQUIRE(hh && nq >= 0 && limit >= 1 && limit <= 64, "dbhip_hnsw_search: bad argument (limit <= 64)");
  if (nq == 0) return DBHIP_OK;
  DBHIP_REQUIRE(queries_dev && out_ids_dev && out_dist_dev, "dbhip_hnsw_search: NULL buffer");
  dbhip_hnsw_impl* h = (dbhip_hnsw_impl*)hh;
  hipStream_t s = resolve_stream(stream);
  const int grid = search_grid(nq);
  uint32_t* vis = (uint32_t*)scratch((size_t)grid * HN_VCAP * 4 + 64, 4, s);
  if (!vis) return DBHIP_ERR_HIP;
  unsigned int* ctl = (unsigned int*)(vis + (size_t)grid * HN_VCAP);
  DBHIP_CHECK(hipMemsetAsync(ctl, 0, 16, s));
  SearchArgs A;
  A.queries = queries_dev; A.nq = nq; A.limit = limit; A.out_ids = out_ids_dev; A.out_dist = out_dist_dev; A.vis = vis;
  A.next = ctl; A.err = ctl + 1;
  static const bool small_off = exp_env("DBHIP_HNSW_SMALL") && atoi(exp_env("DBHIP_HNSW_SMALL")) == 0;
  const bool small = limit * 4 <= 64 && !small_off && !h->small_search_off;
  unsigned int hc[2];
  kernel_timer_start(s);
  if (small) hipLaunchKernelGGL(hn_search_kernel<true>, dim3(grid), dim3(64), 0, s, h->v, A);
  else hipLaunchKernelGGL(hn_search_kernel<false>, dim3(grid), dim3(64), 0, s, h->v, A);
  kernel_timer_stop(s);
  DBHIP_LAUNCH_CHECK();
  DBHIP_CHECK(hipMemcpyAsync(hc, ctl, 8, hipMemcpyDeviceToHost, s));
  DBHIP_CHECK(hipStreamSynchronize(s));
  if (small && (hc[1] & 12u)) {   // some walk outgrew the LDS-resident tables: the whole batch again through the general kernel, and
    h->small_search_off = 1;       // this index (its data: i.i.d. vectors make every walk long) stays on it from now on
    DBHIP_CHECK(hipMemsetAsync(ctl, 0, 16, s));
    hipLaunchKernelGGL(hn_search_kernel<false>, dim3(grid), dim3(64), 0, s, h->v, A);
    DBHIP_LAUNCH_CHECK();
    DBHIP_CHECK(hipMemcpyAsync(hc, ctl, 8, hipMemcpyDeviceToHost, s));
    DBHIP_CHECK(hipStreamSynchronize(s));
  }
  if (hc[1]) {
    set_error("dbhip_hnsw_search: a walk outgrew its candidate heap / visited table (flags %u)", hc[1]);
    return DBHIP_ERR_CAPACITY;
  }
  return DBHIP_OK;
}

int32_t dbhip_hnsw_scores(dbhip_hnsw* hh, const float* queries_dev, int32_t nq, float* out_dev, void* stream) {
  DBHIP_REQUIRE(hh && nq >= 0, "dbhip_hnsw_scores: bad argument");
  dbhip_hnsw_impl* h = (dbhip_hnsw_impl*)hh;
  if (nq == 0 || h->v.n == 0) return DBHIP_OK;
  DBHIP_REQUIRE(queries_dev && out_dev, "dbhip_hnsw_scores: NULL buffer");
  hipStream_t s = resolve_stream(stream);
  int gx = (int)ceil_div(h->v.n, 16);
  if (gx > 1024) gx = 1024;
  hipLaunchKernelGGL(hn_scores_kernel, dim3(gx, nq), dim3(256), 0, s, h->v, queries_dev, nq, out_dev);
  DBHIP_LAUNCH_CHECK();
  return DBHIP_OK;
}

int32_t dbhip_hnsw_encoded(dbhip_hnsw* hh, void* out_dev, void* stream) {
  DBHIP_REQUIRE(hh, "dbhip_hnsw_encoded: NULL index");
  dbhip_hnsw_impl* h = (dbhip_hnsw_impl*)hh;
  if (h->v.n == 0) return DBHIP_OK;
  DBHIP_REQUIRE(out_dev, "dbhip_hnsw_encoded: NULL buffer");
  hipStream_t s = resolve_stream(stream);
  hipLaunchKernelGGL(hn_layout_kernel, dim3(grid_for(h->v.n * (h->v.adim + 4), 256)), dim3(256), 0, s, h->v.codes, h->v.voff, h->v.n, h->v.adim,
                     (uint8_t*)out_dev);
  DBHIP_LAUNCH_CHECK();
  return DBHIP_OK;
}

int32_t dbhip_hnsw_meta(dbhip_hnsw* hh, float* alpha_host, float* offset_host, float* multiplier_host, int32_t* actual_dim_host) {
  DBHIP_REQUIRE(hh, "dbhip_hnsw_meta: NULL index");
  dbhip_hnsw_impl* h = (dbhip_hnsw_impl*)hh;
  if (alpha_host) *alpha_host = h->v.alpha;
  if (offset_host) *offset_host = h->v.offset;
  if (multiplier_host) *multiplier_host = h->v.mult;
  if (actual_dim_host) *actual_dim_host = h->v.adim;
  return DBHIP_OK;
}

int32_t dbhip_hnsw_destroy(dbhip_hnsw* hh) {
  destroy_impl((dbhip_hnsw_impl*)hh);
  return DBHIP_OK;
}

}  // extern "C"

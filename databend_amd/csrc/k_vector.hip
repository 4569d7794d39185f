// k_vector.hip — exact f32 vector distances and top-k.
//
// Reference: src/common/vector/src/distance.rs:19-95
//   cosine = 1 - sum(a*b) / (sqrt(sum(a*a)) * sqrt(sum(b*b)))      l2 = sqrt(sum((a-b)^2))
//   dot    = sum(a*b)                                              l1 = sum(|a-b|)
// driven one query at a time by functions/src/scalars/vector.rs:497-560 over a flat row-major
// VectorColumn::Float32 (types/vector.rs:377-380), then ORDER BY .. LIMIT k
// (kernels/sort_compare.rs:197-209). Here queries are batched:
//   dot / cosine : C[q][i] = sum_k Q[q][k] * B[i][k] on v_mfma_f32_32x32x2_f32 (exact f32, a k-ordered
//                  fmaf chain; MFMA-bound once the query batch is >~ 40), row norms
//                  from one extra streaming pass;
//   l2 / l1      : same LDS tiling on the VALU (the difference form cannot be a GEMM without
//                  cancellation error).
//   top-k        : distances are produced chunk by chunk into scratch and reduced to the k best
//                  (dist, row id) per query; the n x nq matrix is never materialised.
// f32 results depend on summation order; parity with the reference (ndarray's 8-lane unrolled
// sum) is a tolerance, not bit equality — see DESIGN.md.
#include <mutex>
#include "dev_common.h"
#include "runtime.h"

#include <math.h>
#include <stdlib.h>

#include <new>
#include <vector>

using namespace dbhip;

struct dbhip_vec_index {
  int metric;
  const float* base;   // borrowed: the f32 column (re-read for exact re-scoring)
  int64_t n;
  int dim, dpad;
  uint16_t* bh;        // [n][dpad] bf16
  float* rowA; float* rowX; float* rowY;
};

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int BK = 32;       // k-depth of one LDS tile
constexpr int LDK = BK + 1;  // padded leading dimension: conflict-free ds_read_b32 down a column

// ---------------------------------------------------------------------------
// row norms: out[i] = sqrt(sum_k x[i][k]^2)   (one wave per row)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void row_norm_kernel(const float* __restrict__ x, int64_t n, int dim,
                                                       float* __restrict__ out) {
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t r = wave; r < n; r += nwaves) {
    const float* p = x + r * dim;
    float s = 0.f;
    for (int k = lane_id(); k < dim; k += 64) s = fmaf(p[k], p[k], s);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
    if (lane_id() == 0) out[r] = sqrtf(s);
  }
}

// ---------------------------------------------------------------------------
// MFMA tile kernel for dot / cosine. One workgroup (4 waves) owns a TQ x TI tile of the
// (query, base row) score matrix and walks k in steps of BK = 32:
//   * the next k-tile of both operands is prefetched from HBM into registers (16-B loads, one full
//     128-B line per row per step) while the current one is consumed from LDS by the matrix pipe,
//   * LDS rows are padded to 33 floats so the MFMA operand reads (lane = row) are conflict-free,
//   * every wave holds MQ x MI accumulators of v_mfma_f32_32x32x2_f32:
//       A operand (lane l): Q[q + (l&31)][k + (l>>5)]     B operand: Base[i + (l&31)][k + (l>>5)]
//       C/D reg r: col = l&31 (base row), row = (r&3) + 8*(r>>2) + 4*(l>>5) (query),
//   * the base-row norms of the cosine metric are accumulated from the very registers that stage
//     the B tile (no second pass over the base column).
// Tile shapes: 128 x 128 (2 x 2 waves of 64 x 64) for large query batches, 64 x 256 and 32 x 256
// (waves side by side along the base rows) for small ones; f32 MFMA is so slow (64 cycles per
// instruction per SIMD) that the 32-row variant still streams the base at the HBM rate.
// blockIdx -> tile: the q-tiles of one base tile run back to back on ONE XCD (workgroup b lands on
// XCD b % 8), so the base tile is fetched from HBM once and re-read from that XCD's L2.
// Epilogue MODE_WRITE stores the scores; MODE_FILTER appends (score, row id) to the query's
// candidate list when the score is not worse than the query's threshold tau (top-k, see below).
// ---------------------------------------------------------------------------
constexpr int MODE_WRITE = 0, MODE_FILTER = 1;

struct DotArgs {
  const float* base;      // [n][dim]
  const float* queries;   // [nq][dim]
  const float* qnorm;     // [nq]   (cosine)
  int64_t n;
  int dim, nq;
  int vec4;               // dim % 4 == 0 and both pointers 16-byte aligned
  int n_qtiles;
  int64_t n_itiles;
  float* out;             // MODE_WRITE: [nq][out_ld]
  int64_t out_ld;
  const float* tau;       // MODE_FILTER: threshold of query q = tau[q * tau_stride]
  int64_t tau_stride;
  float* cand_d;          // [nq][cand_cap]
  uint32_t* cand_i;
  uint32_t* cand_cnt;     // [nq]
  uint32_t cand_cap;
  uint32_t row_origin;    // added to the row index in candidate ids
};

// Branch-free staging load of elements k..k+3 of a row (`row` points at the row start and is always
// a valid row; elements at or beyond dim read as zero). VEC4: dim % 4 == 0 and 16-byte aligned rows.
template <bool VEC4>
__device__ __forceinline__ float4 load4(const float* row, int k, int dim) {
  float4 v;
  if (VEC4) {
    const int kk = k < dim ? k : dim - 4;
    v = *(const float4*)(row + kk);
    if (k >= dim) v = make_float4(0.f, 0.f, 0.f, 0.f);
  } else {
    const int last = dim - 1;
    v.x = row[k + 0 < dim ? k + 0 : last];
    v.y = row[k + 1 < dim ? k + 1 : last];
    v.z = row[k + 2 < dim ? k + 2 : last];
    v.w = row[k + 3 < dim ? k + 3 : last];
    if (k + 0 >= dim) v.x = 0.f;
    if (k + 1 >= dim) v.y = 0.f;
    if (k + 2 >= dim) v.z = 0.f;
    if (k + 3 >= dim) v.w = 0.f;
  }
  return v;
}

template <int TQ, int TI, int WQ, bool COSINE, int MODE, bool VEC4>
__global__ __launch_bounds__(256, (TI == 256 && TQ == 64) ? 2 : 3) void dot_tile_kernel(DotArgs A) {
  constexpr int WI = 4 / WQ;
  constexpr int MQ = TQ / WQ / 32, MI = TI / WI / 32;
  constexpr int A_F4 = TQ / 32, B_F4 = TI / 32;  // float4 staged per thread and k-tile
  __shared__ float As[TQ * LDK];
  __shared__ float Bs[TI * LDK];
  __shared__ float Bn[TI];
  __shared__ float Qn[TQ];
  __shared__ float Tau[TQ];

  // XCD-aware tile mapping
  const int64_t slot = blockIdx.x >> 3;
  const int xcd = blockIdx.x & 7;
  const int qt = (int)(slot % A.n_qtiles);
  const int64_t it = (slot / A.n_qtiles) * 8 + xcd;
  if (it >= A.n_itiles) return;
  const int64_t i0 = it * TI;
  const int q0 = qt * TQ;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wq = (wave / WI) * (TQ / WQ), wi = (wave % WI) * (TI / WI);
  const int dim = A.dim;

  f32x16 acc[MQ][MI];
#pragma unroll
  for (int a = 0; a < MQ; ++a)
#pragma unroll
    for (int b = 0; b < MI; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  float4 ra[A_F4], rb[B_F4];
  float bsq[B_F4];
#pragma unroll
  for (int j = 0; j < B_F4; ++j) bsq[j] = 0.f;
  // thread t stages rows (t >> 3) + 32 j of each operand, k offset (t & 7) * 4 of the k-tile.
  // Rows past the end of either operand are clamped to the last valid row: their products land in
  // outputs that are never stored, so no predication is needed on the loads.
  const int kc = (tid & 7) * 4;
  const int r0 = tid >> 3;
  const float* atile = A.queries + (int64_t)q0 * dim;
  const float* btile = A.base + i0 * dim;
  const int alast = (A.nq - q0 < TQ ? A.nq - q0 : TQ) - 1;
  const int blast = (int)(A.n - i0 < TI ? A.n - i0 : TI) - 1;
  uint32_t aoff[A_F4], boff[B_F4];
#pragma unroll
  for (int j = 0; j < A_F4; ++j) aoff[j] = (uint32_t)(r0 + 32 * j < alast ? r0 + 32 * j : alast) * (uint32_t)dim;
#pragma unroll
  for (int j = 0; j < B_F4; ++j) boff[j] = (uint32_t)(r0 + 32 * j < blast ? r0 + 32 * j : blast) * (uint32_t)dim;

  auto gload = [&](int k0) {
#pragma unroll
    for (int j = 0; j < A_F4; ++j) ra[j] = load4<VEC4>(atile + aoff[j], k0 + kc, dim);
#pragma unroll
    for (int j = 0; j < B_F4; ++j) rb[j] = load4<VEC4>(btile + boff[j], k0 + kc, dim);
  };
  auto lds_store = [&]() {
#pragma unroll
    for (int j = 0; j < A_F4; ++j) {
      float* d = As + (r0 + 32 * j) * LDK + kc;  // A_F4 * 32 == TQ: always a tile row
      d[0] = ra[j].x; d[1] = ra[j].y; d[2] = ra[j].z; d[3] = ra[j].w;
    }
#pragma unroll
    for (int j = 0; j < B_F4; ++j) {
      float* d = Bs + (r0 + 32 * j) * LDK + kc;
      d[0] = rb[j].x; d[1] = rb[j].y; d[2] = rb[j].z; d[3] = rb[j].w;
      if (COSINE) bsq[j] = fmaf(rb[j].x, rb[j].x, fmaf(rb[j].y, rb[j].y, fmaf(rb[j].z, rb[j].z, fmaf(rb[j].w, rb[j].w, bsq[j]))));
    }
  };

  gload(0);
  for (int k0 = 0; k0 < dim; k0 += BK) {
    __syncthreads();  // every wave is done reading the previous tile
    lds_store();
    __syncthreads();
    if (k0 + BK < dim) gload(k0 + BK);  // in flight while the matrix pipe works on this tile
#pragma unroll
    for (int kk = 0; kk < BK; kk += 2) {
      const int kl = kk + (lane >> 5);
      float a[MQ], b[MI];
#pragma unroll
      for (int x = 0; x < MQ; ++x) a[x] = As[(wq + x * 32 + (lane & 31)) * LDK + kl];
#pragma unroll
      for (int y = 0; y < MI; ++y) b[y] = Bs[(wi + y * 32 + (lane & 31)) * LDK + kl];
#pragma unroll
      for (int x = 0; x < MQ; ++x)
#pragma unroll
        for (int y = 0; y < MI; ++y) acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[x], b[y], acc[x][y], 0, 0, 0);
    }
  }

  // per-tile side data for the epilogue
  if (COSINE) {
#pragma unroll
    for (int j = 0; j < B_F4; ++j) {
      float s = bsq[j];
      s += __shfl_xor(s, 1, 64);
      s += __shfl_xor(s, 2, 64);
      s += __shfl_xor(s, 4, 64);
      if ((tid & 7) == 0) Bn[(tid + 256 * j) >> 3] = sqrtf(s);
    }
    if (tid < TQ) Qn[tid] = (q0 + tid < A.nq) ? A.qnorm[q0 + tid] : 1.f;
  }
  if (MODE == MODE_FILTER && tid < TQ) Tau[tid] = (q0 + tid < A.nq) ? A.tau[(int64_t)(q0 + tid) * A.tau_stride] : -INFINITY;
  __syncthreads();

#pragma unroll
  for (int x = 0; x < MQ; ++x)
#pragma unroll
    for (int y = 0; y < MI; ++y) {
      const int il = wi + y * 32 + (lane & 31);
      const int64_t i = i0 + il;
      const float bn = COSINE ? Bn[il] : 1.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ql = wq + x * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const int q = q0 + ql;
        float v = acc[x][y][r];
        if (COSINE) v = 1.0f - v / (Qn[ql] * bn);
        if (MODE == MODE_WRITE) {
          if (q < A.nq && i < A.n) A.out[(int64_t)q * A.out_ld + i] = v;
        } else {
          // NaN scores and scores not above tau stay in the race (ordering: cand_less)
          if (q < A.nq && i < A.n && !(v > Tau[ql])) {
            const uint32_t s = atomicAdd(&A.cand_cnt[q], 1u);
            if (s < A.cand_cap) {
              A.cand_d[(int64_t)q * A.cand_cap + s] = v;
              A.cand_i[(int64_t)q * A.cand_cap + s] = A.row_origin + (uint32_t)i;
            }
          }
        }
      }
    }
}

// ---------------------------------------------------------------------------
// VALU tile kernel for the difference metrics: 64 x 64 outputs per block, 4 x 4 per thread
// ---------------------------------------------------------------------------
template <bool L1>
__global__ __launch_bounds__(256) void diff_valu_kernel(const float* __restrict__ base, int64_t n, int dim,
                                                        const float* __restrict__ queries, int nq,
                                                        float* __restrict__ out, int64_t out_ld) {
  __shared__ float As[64 * LDK];
  __shared__ float Bs[64 * LDK];
  const int tid = threadIdx.x;
  const int tq = (tid >> 4) * 4, ti = (tid & 15) * 4;
  const int64_t i0 = (int64_t)blockIdx.x * 64;
  const int q0 = blockIdx.y * 64;
  float acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = 0.f;
  for (int k0 = 0; k0 < dim; k0 += BK) {
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      int e = t * 256 + tid;
      int row = e >> 5, kk = e & 31;
      int k = k0 + kk;
      int q = q0 + row;
      int64_t i = i0 + row;
      As[row * LDK + kk] = (q < nq && k < dim) ? queries[(int64_t)q * dim + k] : 0.f;
      Bs[row * LDK + kk] = (i < n && k < dim) ? base[i * dim + k] : 0.f;
    }
    __syncthreads();
#pragma unroll 8
    for (int kk = 0; kk < BK; ++kk) {
      float a[4], b[4];
#pragma unroll
      for (int x = 0; x < 4; ++x) { a[x] = As[(tq + x) * LDK + kk]; b[x] = Bs[(ti + x) * LDK + kk]; }
#pragma unroll
      for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int y = 0; y < 4; ++y) {
          float d = a[x] - b[y];
          acc[x][y] = L1 ? acc[x][y] + fabsf(d) : fmaf(d, d, acc[x][y]);
        }
    }
    __syncthreads();
  }
#pragma unroll
  for (int x = 0; x < 4; ++x)
#pragma unroll
    for (int y = 0; y < 4; ++y) {
      int q = q0 + tq + x;
      int64_t i = i0 + ti + y;
      if (q < nq && i < n) out[(int64_t)q * out_ld + i] = L1 ? acc[x][y] : sqrtf(acc[x][y]);
    }
}

// ---------------------------------------------------------------------------
// top-k selection (ORDER BY distance LIMIT k, kernels/sort_compare.rs:197-209): order is ascending
// distance, NaN last (taken as +inf), ties by lower row id. Segment-parallel and multi-stage:
//   stage kernel  one workgroup per (segment of SEG scores, query): k best of the segment, sorted
//                 1. every thread takes SEG/256 scores into registers and finds its own minimum
//                 2. T = k-th smallest of the 256 thread minima (rank counting in LDS): at least k
//                    scores are <= T and at most k * SEG/256 can be
//                 3. scores <= T are appended to a small LDS list
//                 4. rank counting inside the list writes the k best in order
//   The per-segment lists of one query form the input of the next stage until one list is left.
// Cost per score: one load, ~3 compares. No per-thread insertion sort, no serial scan per query.
// ---------------------------------------------------------------------------
constexpr int KMAX = 16;
constexpr int SEL_PER_THREAD = 32;
constexpr int SEL_SEG = 256 * SEL_PER_THREAD;  // scores per workgroup
constexpr int SEL_LIST = SEL_PER_THREAD * KMAX;

__device__ __forceinline__ bool cand_less(float d1, uint32_t i1, float d2, uint32_t i2) {
  // ascending distance, NaN last, ties by lower row id
  bool n1 = d1 != d1, n2 = d2 != d2;
  if (n1 != n2) return n2;
  if (!n1 && d1 != d2) return d1 < d2;
  return i1 < i2;
}
// strict total order on (distance, id, position): position only separates the (+inf, 0xFFFFFFFF) fillers
__device__ __forceinline__ bool pair_less(float d1, uint32_t i1, int p1, float d2, uint32_t i2, int p2) {
  if (d1 != d2) return d1 < d2;
  if (i1 != i2) return i1 < i2;
  return p1 < p2;
}

struct SelArgs {
  const float* d;         // [nq][ld] scores
  const uint32_t* ids;    // NULL: id of element i is row_origin + i; else [nq][ld]
  const uint32_t* counts; // NULL: every row holds n elements; else row q holds min(counts[q], ld)
  int64_t ld, n;
  uint32_t row_origin;
  int k;
  int nseg;               // segments per query = gridDim.x (without the carry block)
  const float* prev_d;    // optional carried list [nq][k] (the running best), merged as one more segment
  const uint32_t* prev_i;
  float* out_d;           // [nq][out_ld], segment s writes [s*k, s*k+k)
  uint32_t* out_i;
  int64_t out_ld;
};

__global__ __launch_bounds__(256) void select_stage_kernel(SelArgs A) {
  __shared__ float md[256];
  __shared__ uint32_t mi[256];
  __shared__ float Td;
  __shared__ uint32_t Ti;
  __shared__ float ld_[SEL_LIST];
  __shared__ uint32_t li_[SEL_LIST];
  __shared__ uint32_t lcount;
  const int seg = blockIdx.x, q = blockIdx.y, tid = threadIdx.x, k = A.k;
  float* od = A.out_d + (int64_t)q * A.out_ld + (int64_t)seg * k;
  uint32_t* oi = A.out_i + (int64_t)q * A.out_ld + (int64_t)seg * k;
  if (seg == A.nseg) {  // carry block: copies the running best into the extra segment slot
    if (tid < k) { od[tid] = A.prev_d[(int64_t)q * k + tid]; oi[tid] = A.prev_i[(int64_t)q * k + tid]; }
    return;
  }
  int64_t n = A.n;
  if (A.counts) n = A.counts[q] < (uint64_t)A.ld ? (int64_t)A.counts[q] : A.ld;
  const float* d = A.d + (int64_t)q * A.ld;
  const uint32_t* ids = A.ids ? A.ids + (int64_t)q * A.ld : nullptr;
  const int64_t base = (int64_t)seg * SEL_SEG;

  float v[SEL_PER_THREAD];
  uint32_t id[SEL_PER_THREAD];
  float bd = INFINITY;
  uint32_t bi = 0xFFFFFFFFu;
#pragma unroll
  for (int j = 0; j < SEL_PER_THREAD; ++j) {
    const int64_t p = base + j * 256 + tid;
    const bool in = p < n;
    float x = in ? d[p] : INFINITY;
    x = (x != x) ? INFINITY : x;
    const uint32_t y = in ? (ids ? ids[p] : (uint32_t)(A.row_origin + p)) : 0xFFFFFFFFu;
    v[j] = x; id[j] = y;
    if (x < bd || (x == bd && y < bi)) { bd = x; bi = y; }
  }
  md[tid] = bd; mi[tid] = bi;
  if (tid == 0) lcount = 0;
  __syncthreads();
  {  // rank of this thread's minimum among the 256 minima
    int rank = 0;
    for (int t = 0; t < 256; ++t) rank += pair_less(md[t], mi[t], t, bd, bi, tid) ? 1 : 0;
    if (rank == k - 1) { Td = bd; Ti = bi; }
  }
  __syncthreads();
  const float td = Td;
  const uint32_t ti = Ti;
#pragma unroll
  for (int j = 0; j < SEL_PER_THREAD; ++j) {
    const bool pass = id[j] != 0xFFFFFFFFu && (v[j] < td || (v[j] == td && id[j] <= ti));
    if (pass) {
      const uint32_t s = atomicAdd(&lcount, 1u);
      if (s < (uint32_t)SEL_LIST) { ld_[s] = v[j]; li_[s] = id[j]; }
    }
  }
  __syncthreads();
  const int L = (int)(lcount < (uint32_t)SEL_LIST ? lcount : (uint32_t)SEL_LIST);
  for (int e = tid; e < L; e += 256) {
    const float x = ld_[e];
    const uint32_t y = li_[e];
    int rank = 0;
    for (int t = 0; t < L; ++t) rank += pair_less(ld_[t], li_[t], t, x, y, e) ? 1 : 0;
    if (rank < k) { od[rank] = x; oi[rank] = y; }
  }
  if (tid >= L && tid < k) { od[tid] = INFINITY; oi[tid] = 0xFFFFFFFFu; }
}

// k best of every row of `d` (optionally merged with the running best in out_d/out_i) -> out_d/out_i
int32_t select_topk(const float* d, const uint32_t* ids, const uint32_t* counts, int64_t ld, int64_t n,
                    uint32_t row_origin, int nq, int k, bool have_prev, float* out_d, uint32_t* out_i, hipStream_t s) {
  int64_t nseg = ceil_div(n > 0 ? n : 1, SEL_SEG);
  bool carry = have_prev;
  int flip = 0;
  for (;;) {
    const int64_t slots = nseg + (carry ? 1 : 0);
    const bool last = slots == 1;
    float* sd = out_d;
    uint32_t* si = out_i;
    int64_t out_ld = k;
    if (!last) {
      uint8_t* ws = (uint8_t*)scratch((size_t)nq * slots * k * 8, 9 + flip, s);
      if (!ws) return DBHIP_ERR_HIP;
      sd = (float*)ws;
      si = (uint32_t*)(ws + (size_t)nq * slots * k * 4);
      out_ld = slots * k;
    }
    SelArgs A{};
    A.d = d; A.ids = ids; A.counts = counts; A.ld = ld; A.n = n; A.row_origin = row_origin; A.k = k;
    A.nseg = (int)nseg; A.prev_d = out_d; A.prev_i = out_i; A.out_d = sd; A.out_i = si; A.out_ld = out_ld;
    hipLaunchKernelGGL(select_stage_kernel, dim3((unsigned)slots, (unsigned)nq), dim3(256), 0, s, A);
    DBHIP_LAUNCH_CHECK();
    if (last) return DBHIP_OK;
    // next stage reads the lists just written
    d = sd; ids = si; counts = nullptr; ld = out_ld; n = out_ld; row_origin = 0;
    nseg = ceil_div(n, SEL_SEG);
    carry = false;
    flip ^= 1;
  }
}

// ---------------------------------------------------------------------------
// u8-quantised scoring (cpp/avx2.c:45-139): one wave per base row, v_dot4-style packed dot
// ---------------------------------------------------------------------------
// Fast path (dim % 16 == 0, 16-byte aligned base): a QUARTER wave (16 lanes x 16 B) walks one row, four rows per
// wave at a time, the query staged once per workgroup in LDS; every lane has 16-byte loads in flight (the one-wave-
// per-row kernel below moves 4 bytes per lane per load and reaches 0.38 of the HBM rate).
__global__ __launch_bounds__(256) void score_u8_q16_kernel(const uint8_t* __restrict__ q, const uint8_t* __restrict__ base,
                                                           int64_t n, int dim, int is_l1, float* __restrict__ out) {
  extern __shared__ uint32_t qs[];  // dim / 4 words
  for (int k = threadIdx.x; k < dim / 4; k += 256) qs[k] = ((const uint32_t*)q)[k];
  __syncthreads();
  const int sub = threadIdx.x & 15;
  const int64_t qw = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;  // quarter-wave id
  const int64_t nqw = ((int64_t)gridDim.x * blockDim.x) >> 4;
  const int64_t n_pad = (n + 3) & ~3LL;  // the 4 quarter-waves of a wave stay convergent for the shuffles
  for (int64_t r = qw; r < n_pad; r += nqw) {
    uint32_t s = 0;
    if (r < n) {
      const uint8_t* v = base + r * dim;
      for (int k = sub * 16; k < dim; k += 256) {
        const u32x4 b = *(const u32x4*)(v + k);
        const u32x4 a = *(const u32x4*)(qs + (k >> 2));
        if (is_l1) {
          s += __builtin_amdgcn_sad_u8(a.x, b.x, 0u) + __builtin_amdgcn_sad_u8(a.y, b.y, 0u) + __builtin_amdgcn_sad_u8(a.z, b.z, 0u) +
               __builtin_amdgcn_sad_u8(a.w, b.w, 0u);
        } else {
          s = __builtin_amdgcn_udot4(a.x, b.x, s, false);
          s = __builtin_amdgcn_udot4(a.y, b.y, s, false);
          s = __builtin_amdgcn_udot4(a.z, b.z, s, false);
          s = __builtin_amdgcn_udot4(a.w, b.w, s, false);
        }
      }
    }
    s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64); s += __shfl_xor(s, 8, 64);
    if (sub == 0 && r < n) out[r] = (float)s;
  }
}

__global__ __launch_bounds__(256) void score_u8_kernel(const uint8_t* __restrict__ q, const uint8_t* __restrict__ base,
                                                       int64_t n, int dim, int is_l1, float* __restrict__ out) {
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t r = wave; r < n; r += nwaves) {
    const uint8_t* v = base + r * dim;
    uint32_t s = 0;
    for (int k = lane_id() * 4; k < dim; k += 256) {
      uint32_t a = 0, b = 0;
      if (k + 4 <= dim) { a = *(const uint32_t*)(q + k); b = *(const uint32_t*)(v + k); }
      else for (int t = 0; k + t < dim; ++t) { a |= (uint32_t)q[k + t] << (8 * t); b |= (uint32_t)v[k + t] << (8 * t); }
      if (is_l1) s += __builtin_amdgcn_sad_u8(a, b, 0u);
      else s = __builtin_amdgcn_udot4(a, b, s, false);
    }
    s = (uint32_t)wave_sum_u64(s);
    if (lane_id() == 0) out[r] = (float)s;
  }
}

template <int TQ, int TI, int WQ, bool VEC4>
void launch_dot_shape(bool cosine, int mode, DotArgs& A, hipStream_t s) {
  A.n_qtiles = (int)ceil_div(A.nq, TQ);
  A.n_itiles = ceil_div(A.n, TI);
  const int64_t blocks = ceil_div(A.n_itiles, 8) * 8 * A.n_qtiles;
  dim3 grid((unsigned)blocks), block(256);
  if (cosine) {
    if (mode == MODE_WRITE) hipLaunchKernelGGL((dot_tile_kernel<TQ, TI, WQ, true, MODE_WRITE, VEC4>), grid, block, 0, s, A);
    else hipLaunchKernelGGL((dot_tile_kernel<TQ, TI, WQ, true, MODE_FILTER, VEC4>), grid, block, 0, s, A);
  } else {
    if (mode == MODE_WRITE) hipLaunchKernelGGL((dot_tile_kernel<TQ, TI, WQ, false, MODE_WRITE, VEC4>), grid, block, 0, s, A);
    else hipLaunchKernelGGL((dot_tile_kernel<TQ, TI, WQ, false, MODE_FILTER, VEC4>), grid, block, 0, s, A);
  }
}

// dot / cosine scores of base rows [0, n) against nq queries; tile shape by query-batch size
int32_t launch_dot(bool cosine, int mode, DotArgs A, hipStream_t s) {
  if (A.n <= 0 || A.nq <= 0) return DBHIP_OK;
  A.vec4 = (A.dim % 4 == 0) && A.dim >= 4 && (((uintptr_t)A.base | (uintptr_t)A.queries) % 16 == 0);
  if (A.vec4) {
    if (A.nq <= 32) launch_dot_shape<32, 256, 1, true>(cosine, mode, A, s);
    else if (A.nq <= 64) launch_dot_shape<64, 256, 1, true>(cosine, mode, A, s);
    else launch_dot_shape<128, 128, 2, true>(cosine, mode, A, s);
  } else {  // odd dims / unaligned columns: scalar staging loads, one shape
    launch_dot_shape<128, 128, 2, false>(cosine, mode, A, s);
  }
  DBHIP_LAUNCH_CHECK();
  return DBHIP_OK;
}

int32_t launch_distance(int metric, const float* base, int64_t n, int dim, const float* queries, int nq,
                        const float* qnorm, float* out, int64_t out_ld, hipStream_t s) {
  if (metric == DBHIP_VEC_DOT || metric == DBHIP_VEC_COSINE) {
    DotArgs A{};
    A.base = base; A.queries = queries; A.qnorm = qnorm; A.n = n; A.dim = dim; A.nq = nq;
    A.out = out; A.out_ld = out_ld;
    return launch_dot(metric == DBHIP_VEC_COSINE, MODE_WRITE, A, s);
  }
  dim3 grid((unsigned)ceil_div(n, 64), (unsigned)ceil_div(nq, 64));
  if (metric == DBHIP_VEC_L1)
    hipLaunchKernelGGL(diff_valu_kernel<true>, grid, dim3(256), 0, s, base, n, dim, queries, nq, out, out_ld);
  else
    hipLaunchKernelGGL(diff_valu_kernel<false>, grid, dim3(256), 0, s, base, n, dim, queries, nq, out, out_ld);
  DBHIP_LAUNCH_CHECK();
  return DBHIP_OK;
}

constexpr uint32_t CAND_CAP = 4096;  // candidates kept per query by the filtered pass

// chunked materialising path over base rows [lo, hi): always exact
int32_t exact_topk_range(int metric, const float* base, int64_t lo, int64_t hi, int dim, const float* queries, int nq,
                         int k, const float* qnorm, bool have_prev, uint32_t* out_idx, float* out_dist, hipStream_t s) {
  int64_t chunk = (int64_t)(1LL << 28) / nq;
  chunk = chunk < 4096 ? 4096 : chunk;
  chunk = (chunk / 256) * 256;
  if (chunk > hi - lo) chunk = ceil_div(hi - lo > 0 ? hi - lo : 1, 256) * 256;
  float* dist = (float*)scratch((size_t)chunk * nq * 4, 6, s);
  if (!dist) return DBHIP_ERR_HIP;
  bool first = !have_prev;
  for (int64_t c0 = lo; c0 < hi || first; c0 += chunk) {
    int64_t cn = hi - c0 < chunk ? hi - c0 : chunk;
    int32_t rc = DBHIP_OK;
    if (cn > 0) {
      rc = launch_distance(metric, base + c0 * dim, cn, dim, queries, nq, qnorm, dist, chunk, s);
      if (rc) return rc;
    }
    rc = select_topk(dist, nullptr, nullptr, chunk, cn > 0 ? cn : 0, (uint32_t)c0, nq, k, !first, out_dist, out_idx, s);
    if (rc) return rc;
    first = false;
    if (hi <= lo) break;
  }
  return DBHIP_OK;
}

// rows of the exactly scored sample: its k-th best is an upper bound (tau) of the final k-th best
int64_t sample_rows(int64_t n) {
  if (n <= 65536) return n;
  int64_t S = n / 64 > 65536 ? n / 64 : 65536;
  S = ceil_div(S, 256) * 256;
  return S > n ? n : S;
}

// Exact top-k of one query batch (nq queries, all on the device).
//   1. sample   : rows [0, S) are scored into scratch and reduced to their exact top-k; the k-th
//                 best of the sample is an upper bound (tau) of the query's final k-th best.
//   2. filter   : rows [S, n) are scored by the MFMA kernel whose epilogue appends only scores
//                 <= tau to the query's candidate list — the n x nq matrix never exists.
//   3. reduce   : sample top-k + candidates -> final top-k (ties by lower row id).
// If a candidate list overflows (adversarial row order) the remaining rows are redone with the
// chunked materialising path, which is always exact.
int32_t topk_batch(int metric, const float* base, int64_t n, int dim, const float* queries, int nq, int k,
                   const float* qnorm, uint32_t* out_idx, float* out_dist, hipStream_t s) {
  const bool gemm = metric == DBHIP_VEC_DOT || metric == DBHIP_VEC_COSINE;
  const int64_t S = gemm ? sample_rows(n) : n;
  int32_t rc = exact_topk_range(metric, base, 0, S, dim, queries, nq, k, qnorm, false, out_idx, out_dist, s);
  if (rc || S >= n) return rc;

  uint8_t* ws = (uint8_t*)scratch((size_t)nq * CAND_CAP * 8 + (size_t)nq * 4 + 64, 8, s);
  if (!ws) return DBHIP_ERR_HIP;
  uint32_t* cnt = (uint32_t*)ws;
  float* cand_d = (float*)(ws + (((size_t)nq * 4 + 63) & ~(size_t)63));
  uint32_t* cand_i = (uint32_t*)(cand_d + (size_t)nq * CAND_CAP);
  DBHIP_CHECK(hipMemsetAsync(cnt, 0, (size_t)nq * 4, s));
  DotArgs A{};
  A.base = base + S * dim; A.queries = queries; A.qnorm = qnorm; A.n = n - S; A.dim = dim; A.nq = nq;
  A.tau = out_dist + (k - 1); A.tau_stride = k;
  A.cand_d = cand_d; A.cand_i = cand_i; A.cand_cnt = cnt; A.cand_cap = CAND_CAP; A.row_origin = (uint32_t)S;
  rc = launch_dot(metric == DBHIP_VEC_COSINE, MODE_FILTER, A, s);
  if (rc) return rc;
  static thread_local std::vector<uint32_t> hcnt;
  hcnt.resize(nq);
  DBHIP_CHECK(hipMemcpyAsync(hcnt.data(), cnt, (size_t)nq * 4, hipMemcpyDeviceToHost, s));
  DBHIP_CHECK(hipStreamSynchronize(s));
  bool overflow = false;
  for (int q = 0; q < nq; ++q) overflow |= hcnt[q] > CAND_CAP;
  if (overflow) return exact_topk_range(metric, base, S, n, dim, queries, nq, k, qnorm, true, out_idx, out_dist, s);
  return select_topk(cand_d, cand_i, cnt, CAND_CAP, CAND_CAP, 0u, nq, k, true, out_dist, out_idx, s);
}

// ---------------------------------------------------------------------------
// Exact ANN index: bf16 pre-filter with a rigorous error bound + exact f32 re-scoring.
//
// The reference answers ORDER BY distance LIMIT k through an HNSW graph over u8-quantised vectors
// (HNSWIndex::{build, search}, hnsw_index/hnsw.rs:62-315): approximate, recall < 1. On MI355X a
// brute-force scan on the bf16 matrix pipe (16x the f32 MFMA rate) that can only OVER-select, followed
// by exact re-scoring of the few survivors, returns the exact top-k (recall 1.0) at a query rate the
// pointer-chasing graph walk cannot reach.
//   build   Bh = bf16(B) (round to nearest even), per row: ||b||, ||bh||, ||b - bh||
//   search  q = qh + ql, b = bh + bl:  q.b = qh.bh + (ql.bh + qh.bl + ql.bl)
//           |q.b - qh.bh| <= ||ql|| ||bh|| + ||qh|| ||bl|| + ||ql|| ||bl||      (Cauchy-Schwarz)
//           plus the f32 accumulation slack of the matrix pipe, c = dim * 2^-23 + 2e-4 (relative to
//           ||qh|| ||bh||). The tile kernel computes qh.bh on v_mfma_f32_32x32x16_bf16 and keeps a row iff the
//           LOWER bound of its distance is <= tau (the exact k-th best of a sample, as above);
//           survivors are re-scored exactly in f32 and reduced with the sample's top-k.
// LDS rows are padded to 72 bf16 (144 B): the 16 lanes of a ds_read_b128 group hit 16 distinct 16-B
// slots (9 * row mod 16 is a bijection), the staging ds_write_b128 of 8 lanes covers one 128-B row.
// ---------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int HBK = 64;  // bf16 elements per k-tile (one 128-B line per row)
constexpr int HLD = 72;  // padded LDS row stride in bf16 elements

struct HArgs {
  const uint16_t* base;     // [n][dpad] bf16, rows [row_origin ..) of the index
  const uint16_t* queries;  // [nq][dpad] bf16
  const float* rowA; const float* rowX; const float* rowY;  // [n] per-row coefficients (see vec_to_bf16_kernel)
  const float* qn; const float* qh; const float* qe;        // [nq] ||q||, ||qh||(1+), ||q - qh||(1+)
  const float* tau;
  int64_t tau_stride;
  int64_t n;
  int dpad, nq, n_qtiles;
  int64_t n_itiles;
  float c;                  // accumulation + rounding slack
  uint32_t* cand_i;         // [nq][cand_cap]
  uint32_t* cand_cnt;
  uint32_t cand_cap, row_origin;
  const float* qcB; const float* qcG; const float* qcT;   // bf16_filter256_kernel: per query B', G', T' (bf16_qcoef_kernel), padded to whole q-tiles
};

// the three per-query numbers of the bound test (see bf16_filter_kernel's epilogue), once per search range instead of once per tile
template <int METRIC>
__global__ __launch_bounds__(256) void bf16_qcoef_kernel(HArgs A, int nq_pad, float* qcB, float* qcG, float* qcT) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= nq_pad) return;
  const int q = t < A.nq ? t : A.nq - 1;
  const float n_ = A.qn[q], h_ = A.qh[q], e_ = A.qe[q];
  const float tau = t < A.nq ? A.tau[(int64_t)q * A.tau_stride] : -INFINITY;
  qcB[t] = e_ + A.c * h_; qcG[t] = h_ + e_;
  qcT[t] = METRIC == 0 ? (1.0f - tau) * n_ : (METRIC == 1 ? -tau : 0.5f * (0.99999f * n_ * n_ - 1.00001f * tau * tau));
}

// TQ = 128: 4 waves (2 x 2), TQ = 256: 8 waves (4 x 2); every wave owns 64 x 64 scores. The taller tile reads the
// base tile (L2 -> LDS) half as often per query and amortises the LDS stores over twice the matrix work.
// METRIC: 0 cosine, 1 dot, 2 l2 (the squared distance ||q||^2 + ||b||^2 - 2 q.b is bounded from BELOW with the upper
// bound of q.b; the per-row term -||b||^2 / 2 rides in rowA, see vec_to_bf16_kernel mode 3)
template <int METRIC, int TQ>
__global__ __launch_bounds__(TQ * 2) __attribute__((amdgpu_waves_per_eu(TQ == 256 ? 4 : 3))) void bf16_filter_kernel(HArgs A) {
  constexpr int TI = 128;
  constexpr int NT = TQ * 2;          // threads
  constexpr int RS = NT / 8;          // rows staged per pass (8 lanes cover one 128-byte row segment)
  constexpr int AJ = TQ / RS, BJ = TI / RS;
  __shared__ __attribute__((aligned(16))) uint16_t As[TQ * HLD];
  __shared__ __attribute__((aligned(16))) uint16_t Bs[TI * HLD];
  __shared__ __attribute__((aligned(16))) float rA[TI], rX[TI], rY[TI], rZ[TI], qB[TQ], qG[TQ], Tau[TQ];

  const int64_t slot = blockIdx.x >> 3;
  const int xcd = blockIdx.x & 7;
  const int qt = (int)(slot % A.n_qtiles);
  const int64_t it = (slot / A.n_qtiles) * 8 + xcd;
  if (it >= A.n_itiles) return;
  const int64_t i0 = it * TI;
  const int q0 = qt * TQ;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wq = (wave >> 1) * 64, wi = (wave & 1) * 64;
  const int dpad = A.dpad;

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  // thread t stages rows (t >> 3) + RS j, 16-byte piece (t & 7) of the 128-byte k-tile row segment
  const int kc = (tid & 7) * 8;
  const int r0 = tid >> 3;
  const uint16_t* atile = A.queries + (int64_t)q0 * dpad;
  const uint16_t* btile = A.base + i0 * dpad;
  const int alast = (A.nq - q0 < TQ ? A.nq - q0 : TQ) - 1;
  const int blast = (int)(A.n - i0 < TI ? A.n - i0 : TI) - 1;
  uint32_t aoff[AJ], boff[BJ];
#pragma unroll
  for (int j = 0; j < AJ; ++j) aoff[j] = (uint32_t)(r0 + RS * j < alast ? r0 + RS * j : alast) * (uint32_t)dpad + kc;
#pragma unroll
  for (int j = 0; j < BJ; ++j) boff[j] = (uint32_t)(r0 + RS * j < blast ? r0 + RS * j : blast) * (uint32_t)dpad + kc;
  u32x4 ra[AJ], rb[BJ];
  auto gload = [&](int k0) {
#pragma unroll
    for (int j = 0; j < AJ; ++j) ra[j] = *(const u32x4*)(atile + aoff[j] + k0);
#pragma unroll
    for (int j = 0; j < BJ; ++j) rb[j] = *(const u32x4*)(btile + boff[j] + k0);
  };
  auto lds_store = [&]() {
#pragma unroll
    for (int j = 0; j < AJ; ++j) *(u32x4*)(As + (r0 + RS * j) * HLD + kc) = ra[j];
#pragma unroll
    for (int j = 0; j < BJ; ++j) *(u32x4*)(Bs + (r0 + RS * j) * HLD + kc) = rb[j];
  };

  gload(0);
  for (int k0 = 0; k0 < dpad; k0 += HBK) {
    __syncthreads();
    lds_store();
    __syncthreads();
    if (k0 + HBK < dpad) gload(k0 + HBK);
#pragma unroll
    for (int kk = 0; kk < HBK; kk += 16) {
      const int kl = kk + (lane >> 5) * 8;
      const u32x4 a0 = *(const u32x4*)(As + (wq + (lane & 31)) * HLD + kl);
      const u32x4 a1 = *(const u32x4*)(As + (wq + 32 + (lane & 31)) * HLD + kl);
      const u32x4 b0 = *(const u32x4*)(Bs + (wi + (lane & 31)) * HLD + kl);
      const u32x4 b1 = *(const u32x4*)(Bs + (wi + 32 + (lane & 31)) * HLD + kl);
      const bf16x8 fa0 = __builtin_bit_cast(bf16x8, a0), fa1 = __builtin_bit_cast(bf16x8, a1);
      const bf16x8 fb0 = __builtin_bit_cast(bf16x8, b0), fb1 = __builtin_bit_cast(bf16x8, b1);
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa0, fb0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa0, fb1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa1, fb0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa1, fb1, acc[1][1], 0, 0, 0);
    }
  }

  // Epilogue. A row survives unless the LOWER bound of its distance exceeds tau:
  //   cosine  1 - (v a + x B' + y G') / ||q|| > tau     <=>   v a + x B' + y G' < (1 - tau) ||q||      (a = 1/||b||)
  //   dot     v - (x B' + y G') > tau                   <=>  -v   + x B' + y G' < -tau
  //   l2      ||q||^2 + ||b||^2 - 2 (v + x B' + y G') > tau^2  <=>  v + x B' + y G' - ||b||^2/2 < (||q||^2 - tau^2)/2
  // so per query three numbers (B', G', T') and per base row three (a, x, y). The 16 accumulator rows of a lane map to
  // 16 queries that only depend on lane >> 5: their coefficients are fetched with 12 ds_read_b128 per 32-query slab
  // (not 4 scalar LDS reads per score) and the test itself is three VALU operations.
  if (tid < TI) {
    const int64_t i = i0 + tid < A.n ? i0 + tid : A.n - 1;
    rA[tid] = METRIC == 0 ? A.rowA[i] : (METRIC == 1 ? -1.0f : 1.0f);
    rZ[tid] = METRIC == 2 ? A.rowA[i] : 0.0f;
    rX[tid] = A.rowX[i]; rY[tid] = A.rowY[i];
  } else if (tid < TI + TQ) {
    const int t = tid - TI;
    const int q = q0 + t < A.nq ? q0 + t : A.nq - 1;
    const float n_ = A.qn[q], h_ = A.qh[q], e_ = A.qe[q];
    const float tau = (q0 + t < A.nq) ? A.tau[(int64_t)q * A.tau_stride] : -INFINITY;
    // (both sides of the cosine test are multiplied by ||q||: B' = ||ql|| + c ||qh||, G' = ||qh|| + ||ql||)
    qB[t] = e_ + A.c * h_; qG[t] = h_ + e_;
    // l2: discard iff ||q||^2 + ||b||^2 - 2 (v + E) > tau^2  <=>  v + E - ||b||^2/2 < (||q||^2 - tau^2) / 2, with 1e-5 relative
    // slack on both squared norms for their own f32 rounding
    Tau[t] = METRIC == 0 ? (1.0f - tau) * n_ : (METRIC == 1 ? -tau : 0.5f * (0.99999f * n_ * n_ - 1.00001f * tau * tau));
  }
  __syncthreads();

#pragma unroll
  for (int x = 0; x < 2; ++x) {
    float cB[16], cG[16], cT[16];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int qb = wq + x * 32 + 8 * j + 4 * (lane >> 5);
      const float4 vb = *(const float4*)(qB + qb), vg = *(const float4*)(qG + qb), vt = *(const float4*)(Tau + qb);
      cB[4 * j + 0] = vb.x; cB[4 * j + 1] = vb.y; cB[4 * j + 2] = vb.z; cB[4 * j + 3] = vb.w;
      cG[4 * j + 0] = vg.x; cG[4 * j + 1] = vg.y; cG[4 * j + 2] = vg.z; cG[4 * j + 3] = vg.w;
      cT[4 * j + 0] = vt.x; cT[4 * j + 1] = vt.y; cT[4 * j + 2] = vt.z; cT[4 * j + 3] = vt.w;
    }
#pragma unroll
    for (int y = 0; y < 2; ++y) {
      const int il = wi + y * 32 + (lane & 31);
      const int64_t i = i0 + il;
      const float a_ = rA[il], x_ = rX[il], y_ = rY[il], z_ = rZ[il];
      const bool row_ok = i < A.n;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float f = fmaf(acc[x][y][r], a_, fmaf(x_, cB[r], y_ * cG[r])) + z_;
        if (!(f < cT[r]) && row_ok) {  // NaN bounds stay in the race
          const int q = q0 + wq + x * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          if (q < A.nq) {
            const uint32_t s = atomicAdd(&A.cand_cnt[q], 1u);
            if (s < A.cand_cap) A.cand_i[(int64_t)q * A.cand_cap + s] = A.row_origin + (uint32_t)i;
          }
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Round 4: the 256 x 256 tile, 8-phase schedule (cdna_hip_programming.md "The 256^2 8-phase template") for the bf16 filter.
//
// Eight waves (2 query halves x 4 base quarters), each owning 128 queries x 64 base rows = 8 x 4 fragments of
// v_mfma_f32_16x16x32_bf16; BK = 64; the operands of a k-tile are staged with global_load_lds (16 B per lane, no VGPR staging, no
// LDS stores by the VALU) into a double-buffered 128 KB LDS image, in four UNITS of 128 rows x 128 B per k-tile:
//     A0 / A1 = the first / second 64 query rows of both query halves,   B0 / B1 = the first / second 32 base rows of all four quarters
// — a unit is exactly what one phase's prefetch moves (2 x 512 lanes x 16 B), and every wave reads a unit in ONE phase:
//     phase 1  reads A0        MFMA quadrant (A0, B0)       phase 3  reads A1             quadrant (A1, B1)
//     phase 2  reads B1        quadrant (A0, B1)             phase 4  reads B0 of tile t+1  quadrant (A1, B0)   (B0 stays in registers)
// (phases 5-8: the same on the other buffer). So a unit is dead one phase after it was needed and is restaged two phases later; every
// unit is issued six phases before the phase that reads it, and ONE counted wait per phase, s_waitcnt vmcnt(10) (five units = 10 loads
// may stay in flight), retires exactly the unit the NEXT phase reads; the barrier that follows publishes it (RAW) and fences the
// restaging of the unit read two phases ago (WAR).
// Never vmcnt(0) in the loop. LDS rows are 128 B with the 16-byte slot XOR-swizzled by (row >> 1) & 7: the 16 lanes of a
// ds_read_b128 group (16 consecutive rows, one slot) land on 16 distinct 16-byte positions of the 256-byte bank row; global_load_lds
// writes linearly, so the swizzle is applied to the per-lane GLOBAL address. Fragment layout (A and B alike): lane l holds row l & 15,
// k = 8 (l >> 4) .. + 8; C: column l & 15, rows 4 (l >> 4) + j.
// The epilogue is the bound test of bf16_filter_kernel on the 16 x 16 fragment layout.
typedef float f32x4v __attribute__((ext_vector_type(4)));

// V = 1 (round 6): a phase's fragment reads are split at the k half. The k = 0 half is read before the phase's first barrier as before;
// the k = 1 half is issued right AFTER the cluster's first MFMA pair and lands behind the k = 0 MFMAs — the matrix pipe starts after 4 (2)
// ds_read_b128 instead of 8 (4), and half of every phase's LDS latency disappears behind MFMA issue. (The compiler waits lgkmcnt(0)
// before the first MFMA that uses a pending fragment and does not count: reads issued before that MFMA would be waited for as well,
// hence the placement behind it, pinned with sched_barrier.) V = 0: rounds 4-5.
template <int METRIC, int V = 1>
__global__ __launch_bounds__(512) void bf16_filter256_kernel(HArgs A) {
  extern __shared__ __attribute__((aligned(1024))) uint8_t fsm[];   // 2 buffers x 4 units x 16 KB
  constexpr int TQ = 256, TI = 256;
  // One output tile (256 queries x 256 base rows) per workgroup; the q-tiles of one base tile run back to back on one XCD (workgroup
  // b -> XCD b mod 8), so the base tile is fetched from HBM about once. (r04q tried the workgroup PERSISTENT over the query tiles of a
  // base tile — the pipeline filled once, the bound test of a finished tile overlapping the next tile's loads: with the test inside
  // the loop the kernel needs more than its 256 registers, spills into the k-loop and runs at 4.65 ms instead of 3.65 ms.)
  const int64_t slot = blockIdx.x >> 3;
  const int xcd = blockIdx.x & 7;
  const int qt = (int)(slot % A.n_qtiles);
  const int64_t it = (slot / A.n_qtiles) * 8 + xcd;
  if (it >= A.n_itiles) return;
  const int64_t i0 = it * TI;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 2, wc = wave & 3;
  const int dpad = A.dpad;
  const int nkt = dpad / HBK;   // k-tiles per output tile (even: checked by the host)
  const int gtot = nkt;

  // ---- staging: this thread's two 16-byte pieces of every unit ----
  // piece j of a unit: unit row u = 64 j + (tid >> 3), physical slot tid & 7 = logical slot ^ ((u >> 1) & 7)
  // (the query image is padded to whole 256-row tiles with zero rows: no clamp on that side)
  const int blast = (int)(A.n - i0 < TI ? A.n - i0 : TI) - 1;
  uint32_t goff[4][2];   // [unit][piece]: element offset of the piece's first bf16 inside the query / base tile at k-tile 0
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int u = 64 * j + (tid >> 3);
    const int sl = (tid & 7) ^ ((u >> 1) & 7);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int ra = (u >> 6) * 128 + s * 64 + (u & 63);      // A_s: both query halves' rows [64 s, 64 s + 64)
      int rb = (u >> 5) * 64 + s * 32 + (u & 31);             // B_s: all four base quarters' rows [32 s, 32 s + 32)
      rb = rb < blast ? rb : blast;
      goff[s][j] = (uint32_t)ra * (uint32_t)dpad + sl * 8;
      goff[2 + s][j] = (uint32_t)rb * (uint32_t)dpad + sl * 8;
    }
  }
  const uint16_t* btile = A.base + i0 * dpad;
  const int64_t qtile_elems = (int64_t)TQ * dpad;
  // unit `un` (0 A0, 1 A1, 2 B0, 3 B1) of global k-tile `g` into buffer `buf`; k-tiles past the end re-load the last one (the slot
  // is dead by construction, the data is never read) so that the load count per phase stays uniform
  auto stage = [&](int un, int buf, int g) {
    g = g < gtot ? g : gtot - 1;
    const int k0 = g * HBK;
    const uint16_t* src = un < 2 ? A.queries + qt * qtile_elems : btile;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const uint32_t go = un == 0 ? goff[0][j] : (un == 1 ? goff[1][j] : (un == 2 ? goff[2][j] : goff[3][j]));
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + go + k0),
                                       (__attribute__((address_space(3))) void*)(fsm + buf * 65536 + un * 16384 + j * 8192 + wave * 1024), 16, 0, 0);
    }
  };
  // ---- fragment reads ----
  const int fr = lane & 15, fg = lane >> 4;
  auto frag_off = [&](int u, int sl) { return u * 128 + ((sl ^ ((u >> 1) & 7)) << 4); };
  // [k half] of fragment 0; fragment m lies m * 16 rows = m * 2048 bytes further with the SAME swizzle term ((u >> 1) & 7 does not see
  // multiples of 16 rows): an immediate offset of the ds_read instead of a register per fragment (round 6: the registers this frees are
  // what the split-half schedule needs)
  uint32_t aoff0[2], boff0[2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
    aoff0[kk] = (uint32_t)frag_off(wr * 64 + fr, kk * 4 + fg);
    boff0[kk] = (uint32_t)frag_off(wc * 32 + fr, kk * 4 + fg);
  }
#define aoffs(m, kk) (aoff0[kk] + (uint32_t)(m) * 2048u)
#define boffs(nn, kk) (boff0[kk] + (uint32_t)(nn) * 2048u)
  bf16x8 fa[4][2], fb0[2][2], fb1[2][2];
  auto read_a = [&](int buf, int un, int half = 0) {
    const uint8_t* base = fsm + buf * 65536 + un * 16384;
    if (V == 1) {
#pragma unroll
      for (int m = 0; m < 4; ++m) fa[m][half] = __builtin_bit_cast(bf16x8, *(const u32x4*)(base + aoffs(m, half)));
      return;
    }
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) fa[m][kk] = __builtin_bit_cast(bf16x8, *(const u32x4*)(base + aoffs(m, kk)));
  };
  auto read_b = [&](int buf, int un, bf16x8 (&fb)[2][2], int half = -1) {
    const uint8_t* base = fsm + buf * 65536 + un * 16384;
    if (V == 1 && half >= 0) {
#pragma unroll
      for (int nn = 0; nn < 2; ++nn) fb[nn][half] = __builtin_bit_cast(bf16x8, *(const u32x4*)(base + boffs(nn, half)));
      return;
    }
#pragma unroll
    for (int nn = 0; nn < 2; ++nn)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) fb[nn][kk] = __builtin_bit_cast(bf16x8, *(const u32x4*)(base + boffs(nn, kk)));
  };
  f32x4v acc[8][4];
#pragma unroll
  for (int m = 0; m < 8; ++m)
#pragma unroll
    for (int nn = 0; nn < 4; ++nn) acc[m][nn] = f32x4v{0.f, 0.f, 0.f, 0.f};
  // 16 MFMAs: quadrant (sa, sb) of the wave's 8 x 4 fragments
  // `fresh` (V = 1): what this phase's own reads deliver — 1: the A fragments of unit (rbuf, run), 2: the B fragments of unit (rbuf, run)
  // into `fb`, 0: nothing the cluster uses (the phase that reads B0 of the NEXT k-tile works from registers and reads both halves up front)
  auto mfma_quadrant = [&](int sa, int sb, bf16x8 (&fb)[2][2], int fresh = 0, int rbuf = 0, int run = 0) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int m = 0; m < 4; ++m) {
#pragma unroll
        for (int nn = 0; nn < 2; ++nn)
          acc[sa * 4 + m][sb * 2 + nn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[m][kk], fb[nn][kk], acc[sa * 4 + m][sb * 2 + nn], 0, 0, 0);
        if (V == 1 && fresh != 0 && kk == 0 && m == 0) {   // the k = 1 half, behind the first MFMA pair
          __builtin_amdgcn_sched_barrier(0);
          if (fresh == 1) read_a(rbuf, run, 1); else read_b(rbuf, run, fb, 1);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    __builtin_amdgcn_s_setprio(0);
  };

  // Per phase: [LDS reads of this phase's operands | this phase's unit prefetch | counted wait] BARRIER [16 MFMAs] BARRIER. The second
  // barrier exists for the STAGGER: the four waves of the second query half run one barrier behind the first half's (one extra
  // barrier before the loop, one for the first half after it), so while one half issues its MFMAs the other half — the other wave of
  // every SIMD: wave w and w + 4 share SIMD w & 3 — does its LDS reads and prefetches.
  // RAW: a unit is waited for before the first barrier of phase p by every wave that issued a piece of it and read in phase p + 1,
  // i.e. at least two barriers later for either half. WAR: a unit is restaged two phases after its last read.
#define F256_MEM_SYNC()                              \
  asm volatile("s_waitcnt vmcnt(10)" ::: "memory");  \
  __builtin_amdgcn_s_barrier();                      \
  if (V == 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define F256_END_SYNC() __builtin_amdgcn_s_barrier()

  // LDS reads per phase are BALANCED (8 / 4 / 8 / 4 ds_read_b128 per wave): B0 of the NEXT k-tile is read in phase 4, whose own MFMAs
  // work from registers, into the B register set that B1 has just left. Per k-tile g:
  //     phase 1  reads A0(g)     MFMA (A0, B0)         issue A1(g+1)        phase 3  reads A1(g)     MFMA (A1, B1)   issue A0(g+2)
  //     phase 2  reads B1(g)     MFMA (A0, B1)         issue B0(g+2)        phase 4  reads B0(g+1)   MFMA (A1, B0)   issue B1(g+2)
  // every unit is issued six phases before the phase that reads it and two after the last read of the unit it replaces; the one
  // counted wait per phase is vmcnt(10): five units may stay in flight, the sixth-youngest — the one the next phase reads — has landed.
  // B0 lives in set X on even k-tiles and in Y on odd ones.
  bf16x8 (&bx)[2][2] = fb0;
  bf16x8 (&by)[2][2] = fb1;
  stage(2, 0, 0); stage(0, 0, 0); stage(3, 0, 0); stage(1, 0, 0); stage(2, 1, 1); stage(0, 1, 1); stage(3, 1, 1);
  asm volatile("s_waitcnt vmcnt(12)" ::: "memory");   // B0(0) has landed
  __builtin_amdgcn_s_barrier();
  read_b(0, 2, bx);
  asm volatile("s_waitcnt vmcnt(10)" ::: "memory");   // A0(0) has landed
  __builtin_amdgcn_s_barrier();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if (wr == 1) __builtin_amdgcn_s_barrier();          // the second half starts one barrier behind
  {
    for (int t = 0; t < nkt; t += 2) {
      const int g = t;
      // ---- k-tile g in buffer 0 (B0 in X, B1 in Y) ----
      read_a(0, 0);
      stage(1, 1, g + 1);
      F256_MEM_SYNC();
      mfma_quadrant(0, 0, bx, 1, 0, 0);
      F256_END_SYNC();
      read_b(0, 3, by, V == 1 ? 0 : -1);
      stage(2, 0, g + 2);
      F256_MEM_SYNC();
      mfma_quadrant(0, 1, by, 2, 0, 3);
      F256_END_SYNC();
      read_a(0, 1);
      stage(0, 0, g + 2);
      F256_MEM_SYNC();
      mfma_quadrant(1, 1, by, 1, 0, 1);
      F256_END_SYNC();
      read_b(1, 2, by);            // B0(g + 1) into Y (B1(g) has had its last use)
      stage(3, 0, g + 2);
      F256_MEM_SYNC();
      mfma_quadrant(1, 0, bx, 0);
      F256_END_SYNC();
      // ---- k-tile g + 1 in buffer 1 (B0 in Y, B1 in X) ----
      read_a(1, 0);
      stage(1, 0, g + 2);
      F256_MEM_SYNC();
      mfma_quadrant(0, 0, by, 1, 1, 0);
      F256_END_SYNC();
      read_b(1, 3, bx, V == 1 ? 0 : -1);
      stage(2, 1, g + 3);
      F256_MEM_SYNC();
      mfma_quadrant(0, 1, bx, 2, 1, 3);
      F256_END_SYNC();
      read_a(1, 1);
      stage(0, 1, g + 3);
      F256_MEM_SYNC();
      mfma_quadrant(1, 1, bx, 1, 1, 1);
      F256_END_SYNC();
      read_b(0, 2, bx);            // B0(g + 2) into X
      stage(3, 1, g + 3);
      F256_MEM_SYNC();
      mfma_quadrant(1, 0, by, 0);
      F256_END_SYNC();
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();          // (the barrier the second half's last phase pairs with)
    // ---- the bound test of bf16_filter_kernel; the per-query numbers come precomputed (bf16_qcoef_kernel), the per-row ones from
    // the index ----
    const int q0 = qt * TQ;
    // (the base rows' coefficients come from the L2 every output tile rather than living in 16 registers through the k-loop: with
    // them the kernel needed 256 registers + scratch, r04q)
    float ba[4], bx_[4], by_[4], bz[4];
    bool bok[4];
#pragma unroll
    for (int nn = 0; nn < 4; ++nn) {
      const int64_t i = i0 + wc * 64 + nn * 16 + fr;
      bok[nn] = i < A.n;
      const int64_t ic = bok[nn] ? i : A.n - 1;
      ba[nn] = METRIC == 0 ? A.rowA[ic] : (METRIC == 1 ? -1.0f : 1.0f);
      bz[nn] = METRIC == 2 ? A.rowA[ic] : 0.0f;
      bx_[nn] = A.rowX[ic]; by_[nn] = A.rowY[ic];
    }
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      const int ql = q0 + wr * 128 + m * 16 + fg * 4;   // this lane's 4 queries of fragment row m
      const float4 vb = *(const float4*)(A.qcB + ql), vg = *(const float4*)(A.qcG + ql), vt = *(const float4*)(A.qcT + ql);
      const float cB[4] = {vb.x, vb.y, vb.z, vb.w}, cG[4] = {vg.x, vg.y, vg.z, vg.w}, cT[4] = {vt.x, vt.y, vt.z, vt.w};
#pragma unroll
      for (int nn = 0; nn < 4; ++nn) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float f = fmaf(acc[m][nn][j], ba[nn], fmaf(bx_[nn], cB[j], by_[nn] * cG[j])) + bz[nn];
          if (!(f < cT[j]) && bok[nn]) {  // NaN bounds stay in the race
            const int q = ql + j;
            if (q < A.nq) {
              const uint32_t sidx = atomicAdd(&A.cand_cnt[q], 1u);
              if (sidx < A.cand_cap) A.cand_i[(int64_t)q * A.cand_cap + sidx] = A.row_origin + (uint32_t)(i0 + wc * 64 + nn * 16 + fr);
            }
          }
        }
      }
    }
  }
#undef F256_MEM_SYNC
#undef F256_END_SYNC
#undef aoffs
#undef boffs
}

// One wave per row: out[row][0..dpad) = bf16(x[row]) (RNE, zero padded) and the row's coefficients.
//   mode 0 (base rows, cosine): A = 1/||b||, X = ||bh||/||b|| (1+1e-4), Y = ||b-bh||/||b|| (1+1e-4)
//   mode 1 (base rows, dot)   : A = 1,       X = ||bh|| (1+1e-4),       Y = ||b-bh|| (1+1e-4)
//   mode 2 (queries)          : A = ||q||,   X = ||qh|| (1+1e-4),       Y = ||q-qh|| (1+1e-4)
//   mode 3 (base rows, l2)    : A = -||b||^2 / 2 (1-1e-5), X = ||bh|| (1+1e-4), Y = ||b-bh|| (1+1e-4)
__global__ __launch_bounds__(256) void vec_to_bf16_kernel(const float* __restrict__ x, int64_t n, int dim, int dpad,
                                                          int mode, uint16_t* __restrict__ out, float* __restrict__ oA,
                                                          float* __restrict__ oX, float* __restrict__ oY) {
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t r = wave; r < n; r += nwaves) {
    const float* p = x + r * dim;
    uint16_t* o = out + r * dpad;
    float s = 0.f, sh = 0.f, se = 0.f;
    for (int k = lane_id(); k < dpad; k += 64) {
      uint16_t hb = 0;
      if (k < dim) {
        const float v = p[k];
        uint32_t b = __float_as_uint(v);
        if (v != v) b |= 0x00400000u;                              // quiet NaN survives the truncation
        else b += 0x7FFFu + ((b >> 16) & 1u);                      // round to nearest even
        hb = (uint16_t)(b >> 16);
        const float h = __uint_as_float((uint32_t)hb << 16);
        const float d = v - h;
        s = fmaf(v, v, s); sh = fmaf(h, h, sh); se = fmaf(d, d, se);
      }
      o[k] = hb;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      s += __shfl_xor(s, off, 64); sh += __shfl_xor(sh, off, 64); se += __shfl_xor(se, off, 64);
    }
    if (lane_id() == 0) {
      const float nb = sqrtf(s), nh = sqrtf(sh) * 1.0001f, ne = sqrtf(se) * 1.0001f;
      if (mode == 0) { oA[r] = 1.0f / nb; oX[r] = nh / nb; oY[r] = ne / nb; }
      else if (mode == 1) { oA[r] = 1.0f; oX[r] = nh; oY[r] = ne; }
      else if (mode == 3) { oA[r] = -0.5f * 0.99999f * s; oX[r] = nh; oY[r] = ne; }
      else { oA[r] = nb; oX[r] = nh; oY[r] = ne; }
    }
  }
}

// exact f32 distance of every surviving (query, row) pair: one wave per pair
__global__ __launch_bounds__(256) void rescore_kernel(int metric, const float* __restrict__ base, int dim,
                                                      const float* __restrict__ queries, const float* __restrict__ qnorm,
                                                      const uint32_t* __restrict__ cand_i, const uint32_t* __restrict__ cand_cnt,
                                                      uint32_t cap, float* __restrict__ cand_d) {
  const int q = blockIdx.y;
  const uint32_t j = blockIdx.x * 4 + (threadIdx.x >> 6);
  const uint32_t cnt = cand_cnt[q] < cap ? cand_cnt[q] : cap;
  if (j >= cnt) return;
  const uint32_t i = cand_i[(int64_t)q * cap + j];
  const float* b = base + (int64_t)i * dim;
  const float* qv = queries + (int64_t)q * dim;
  float dot = 0.f, bsq = 0.f;
  if (metric == DBHIP_VEC_L2) {  // the difference form of the reference (distance.rs:65-80), never the expanded one
    for (int k = lane_id(); k < dim; k += 64) {
      const float d = qv[k] - b[k];
      dot = fmaf(d, d, dot);
    }
  } else {
    for (int k = lane_id(); k < dim; k += 64) {
      const float bv = b[k];
      dot = fmaf(qv[k], bv, dot);
      bsq = fmaf(bv, bv, bsq);
    }
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) { dot += __shfl_xor(dot, off, 64); bsq += __shfl_xor(bsq, off, 64); }
  if (lane_id() == 0)
    cand_d[(int64_t)q * cap + j] = metric == DBHIP_VEC_COSINE ? 1.0f - dot / (qnorm[q] * sqrtf(bsq)) : (metric == DBHIP_VEC_L2 ? sqrtf(dot) : dot);
}


// one query batch through the index (see the block comment above bf16_filter_kernel). Three levels, each
// tightening tau for the next: exact f32 scan of the first S0 = 8192 rows; bf16 filter + exact re-score of
// [S0, S1) (S1 = sample_rows(n)); bf16 filter + exact re-score of [S1, n). Only S0 rows ever run at the f32 rate.
int32_t index_search_batch(dbhip_vec_index* ix, const float* queries, int nq, int k, const float* qnorm,
                           uint32_t* out_idx, float* out_dist, hipStream_t s) {
  const int64_t n = ix->n;
  const int dim = ix->dim, dpad = ix->dpad;
  const bool cosine = ix->metric == DBHIP_VEC_COSINE;
  const bool l2 = ix->metric == DBHIP_VEC_L2;
  const int64_t S0 = n < 8192 ? n : 8192;
  int32_t rc = exact_topk_range(ix->metric, ix->base, 0, S0, dim, queries, nq, k, qnorm, false, out_idx, out_dist, s);
  if (rc || S0 >= n) return rc;

  // query side: bf16 image + norms
  // (the bf16 image of the queries is padded with zero rows to whole 256-row tiles: the 8-phase filter kernel stages them unclamped)
  const int nq_pad = (int)ceil_div(nq, 256) * 256;
  const size_t qh_bytes = (((size_t)nq_pad * dpad * 2) + 255) & ~(size_t)255;
  // (qA / qX / qY hold nq floats each in slots of nq4 = nq rounded up to 4: the coefficient arrays behind them are read as float4
  //  by bf16_filter256_kernel and have to start on 16 bytes whatever nq is)
  const size_t nq4 = ((size_t)nq + 3) & ~(size_t)3;
  uint8_t* qws = (uint8_t*)scratch(qh_bytes + nq4 * 12 + (size_t)nq_pad * 12 + 256, 11, s);
  if (!qws) return DBHIP_ERR_HIP;
  uint16_t* qh = (uint16_t*)qws;
  float* qA = (float*)(qws + qh_bytes);
  float* qX = qA + nq4;
  float* qY = qX + nq4;
  float* qcB = qY + nq4;
  float* qcG = qcB + nq_pad;
  float* qcT = qcG + nq_pad;
  if (nq_pad > nq) DBHIP_CHECK(hipMemsetAsync(qh + (size_t)nq * dpad, 0, (size_t)(nq_pad - nq) * dpad * 2, s));
  hipLaunchKernelGGL(vec_to_bf16_kernel, dim3(grid_for((int64_t)nq * 64, 256)), dim3(256), 0, s, queries, (int64_t)nq, dim,
                     dpad, 2, qh, qA, qX, qY);

  uint8_t* ws = (uint8_t*)scratch((size_t)nq * CAND_CAP * 8 + (size_t)nq * 4 + 64, 8, s);
  if (!ws) return DBHIP_ERR_HIP;
  uint32_t* cnt = (uint32_t*)ws;
  float* cand_d = (float*)(ws + (((size_t)nq * 4 + 63) & ~(size_t)63));
  uint32_t* cand_i = (uint32_t*)(cand_d + (size_t)nq * CAND_CAP);
  static thread_local std::vector<uint32_t> hcnt;
  hcnt.resize(nq);

  auto filter_range = [&](int64_t lo, int64_t hi) -> int32_t {
    DBHIP_POLL_CANCEL(s, "dbhip_vec_index_search");
    DBHIP_CHECK(hipMemsetAsync(cnt, 0, (size_t)nq * 4, s));
    HArgs A{};
    A.base = ix->bh + lo * dpad; A.queries = qh;
    A.rowA = ix->rowA + lo; A.rowX = ix->rowX + lo; A.rowY = ix->rowY + lo;
    A.qn = qA; A.qh = qX; A.qe = qY;
    A.tau = out_dist + (k - 1); A.tau_stride = k;
    A.n = hi - lo; A.dpad = dpad; A.nq = nq;
    const bool tall = nq > 128;  // 256-query tiles once there are enough queries to fill them
    // the 256 x 256 8-phase kernel (round 4) when the k-tiles pair up and the range fills the chip; DBHIP_BF16_256=0: the r01 kernel
    static const bool f256_off = exp_env("DBHIP_BF16_256") && atoi(exp_env("DBHIP_BF16_256")) == 0;
    const bool f256 = tall && !f256_off && (dpad / HBK) % 2 == 0 && dpad >= 128 && A.n >= 256;
    A.n_qtiles = (int)ceil_div(nq, tall ? 256 : 128);
    A.n_itiles = ceil_div(A.n, f256 ? 256 : 128);
    A.c = (float)dim * 1.1920929e-07f + 2e-4f;
    A.cand_i = cand_i; A.cand_cnt = cnt; A.cand_cap = CAND_CAP; A.row_origin = (uint32_t)lo;
    const int64_t blocks = ceil_div(A.n_itiles, 8) * 8 * A.n_qtiles;
    if (f256) {
      static std::once_flag raised_once;
      static hipError_t raised_err = hipSuccess;
      std::call_once(raised_once, [] {
        raised_err = hipFuncSetAttribute((const void*)bf16_filter256_kernel<0, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
        if (raised_err == hipSuccess) raised_err = hipFuncSetAttribute((const void*)bf16_filter256_kernel<1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
        if (raised_err == hipSuccess) raised_err = hipFuncSetAttribute((const void*)bf16_filter256_kernel<2, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
        if (raised_err == hipSuccess) raised_err = hipFuncSetAttribute((const void*)bf16_filter256_kernel<0, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
      });
      DBHIP_CHECK(raised_err);
      A.qcB = qcB; A.qcG = qcG; A.qcT = qcT;
      const dim3 cg((unsigned)ceil_div(nq_pad, 256)), fg((unsigned)blocks);
      if (l2) {
        hipLaunchKernelGGL((bf16_qcoef_kernel<2>), cg, dim3(256), 0, s, A, nq_pad, qcB, qcG, qcT);
        hipLaunchKernelGGL((bf16_filter256_kernel<2>), fg, dim3(512), 131072, s, A);
      } else if (cosine) {
        hipLaunchKernelGGL((bf16_qcoef_kernel<0>), cg, dim3(256), 0, s, A, nq_pad, qcB, qcG, qcT);
        // (experiments build: DBHIP_BF16_V=0 runs the rounds-4-5 schedule of the cosine kernel for a same-process A/B)
        static const bool v0 = exp_env("DBHIP_BF16_V") && atoi(exp_env("DBHIP_BF16_V")) == 0;
        if (v0) hipLaunchKernelGGL((bf16_filter256_kernel<0, 0>), fg, dim3(512), 131072, s, A);
        else hipLaunchKernelGGL((bf16_filter256_kernel<0, 1>), fg, dim3(512), 131072, s, A);
      } else {
        hipLaunchKernelGGL((bf16_qcoef_kernel<1>), cg, dim3(256), 0, s, A, nq_pad, qcB, qcG, qcT);
        hipLaunchKernelGGL((bf16_filter256_kernel<1>), fg, dim3(512), 131072, s, A);
      }
    } else if (l2) {
      if (tall) hipLaunchKernelGGL((bf16_filter_kernel<2, 256>), dim3((unsigned)blocks), dim3(512), 0, s, A);
      else hipLaunchKernelGGL((bf16_filter_kernel<2, 128>), dim3((unsigned)blocks), dim3(256), 0, s, A);
    } else if (tall) {
      if (cosine) hipLaunchKernelGGL((bf16_filter_kernel<0, 256>), dim3((unsigned)blocks), dim3(512), 0, s, A);
      else hipLaunchKernelGGL((bf16_filter_kernel<1, 256>), dim3((unsigned)blocks), dim3(512), 0, s, A);
    } else {
      if (cosine) hipLaunchKernelGGL((bf16_filter_kernel<0, 128>), dim3((unsigned)blocks), dim3(256), 0, s, A);
      else hipLaunchKernelGGL((bf16_filter_kernel<1, 128>), dim3((unsigned)blocks), dim3(256), 0, s, A);
    }
    DBHIP_LAUNCH_CHECK();
    DBHIP_CHECK(hipMemcpyAsync(hcnt.data(), cnt, (size_t)nq * 4, hipMemcpyDeviceToHost, s));
    DBHIP_CHECK(hipStreamSynchronize(s));
    uint32_t maxc = 0;
    for (int q = 0; q < nq; ++q) maxc = hcnt[q] > maxc ? hcnt[q] : maxc;
    if (maxc > CAND_CAP)  // the bound could not separate enough rows: the exact scan is always right
      return exact_topk_range(ix->metric, ix->base, lo, hi, dim, queries, nq, k, qnorm, true, out_idx, out_dist, s);
    if (maxc == 0) return DBHIP_OK;
    hipLaunchKernelGGL(rescore_kernel, dim3((unsigned)ceil_div(maxc, 4), (unsigned)nq), dim3(256), 0, s, ix->metric, ix->base,
                       dim, queries, qnorm, cand_i, cnt, CAND_CAP, cand_d);
    DBHIP_LAUNCH_CHECK();
    return select_topk(cand_d, cand_i, cnt, CAND_CAP, CAND_CAP, 0u, nq, k, true, out_dist, out_idx, s);
  };

  const int64_t S1 = sample_rows(n);
  if (S1 > S0 && S1 < n) {
    if ((rc = filter_range(S0, S1))) return rc;
    return filter_range(S1, n);
  }
  return filter_range(S0, n);
}

// Row by row: out[i] = distance(lhs[i], rhs[i]) — the column-vs-column form of the vector scalar functions (scalars/vector.rs:59-260
// Array(Float32 / Float64), :490-560 Vector(Float32 / Int8)). HBM bound (two rows of dim elements per result): G lanes per row
// (G = the power of two that gives every lane ~4 elements, at most a wave), coalesced element-strided loads, a shuffle reduction per
// group. T = element type in memory, F = arithmetic type (f32 for f32 / i8 rows, f64 for f64 rows: the *_64 functions of distance.rs).
template <typename T, typename F>
__global__ __launch_bounds__(256) void vec_rows_kernel(int metric, const T* __restrict__ lhs, int lhs_scalar, const T* __restrict__ rhs, int rhs_scalar,
                                                       int64_t n, int dim, int gshift, F* __restrict__ out) {
  const int G = 1 << gshift;
  const int lane = threadIdx.x & (G - 1);
  const int64_t groups_per_block = 256 >> gshift;
  for (int64_t row = (int64_t)blockIdx.x * groups_per_block + (threadIdx.x >> gshift); row < ((n + groups_per_block - 1) / groups_per_block) * groups_per_block;
       row += (int64_t)gridDim.x * groups_per_block) {
    const bool live = row < n;
    const int64_t r = live ? row : 0;
    const T* a = lhs + (lhs_scalar ? 0 : r * dim);
    const T* b = rhs ? rhs + (rhs_scalar ? 0 : r * dim) : a;
    F s0 = 0, s1 = 0, s2 = 0;
    for (int k = lane; k < dim; k += G) {
      const F x = (F)a[k], y = (F)b[k];
      switch (metric) {   // (uniform)
        case 0: s0 += x * y; s1 += x * x; s2 += y * y; break;
        case 1: { const F d = x - y; s0 += d * d; } break;
        case 2: s0 += x * y; break;
        case 3: { const F d = x - y; s0 += d < 0 ? -d : d; } break;
        default: s0 += x * x; break;
      }
    }
    for (int off = G >> 1; off >= 1; off >>= 1) {
      s0 += __shfl_xor(s0, off, 64);
      if (metric == 0) { s1 += __shfl_xor(s1, off, 64); s2 += __shfl_xor(s2, off, 64); }
    }
    if (live && lane == 0) {
      F res;
      switch (metric) {
        case 0: res = (F)1 - s0 / (sqrt(s1) * sqrt(s2)); break;
        case 1: res = sqrt(s0); break;
        case 4: res = sqrt(s0); break;
        default: res = s0; break;
      }
      out[row] = res;
    }
  }
}

}  // namespace

extern "C" {

int32_t dbhip_vec_index_build(int32_t metric, const float* base, int64_t n, int32_t dim, dbhip_vec_index** out_host,
                              void* stream) {
  DBHIP_REQUIRE(out_host, "dbhip_vec_index_build: NULL out");
  if (metric != DBHIP_VEC_COSINE && metric != DBHIP_VEC_DOT && metric != DBHIP_VEC_L2) {
    set_error("dbhip_vec_index_build: metric %d has no bf16 pre-filter (use dbhip_vec_topk)", metric);
    return DBHIP_ERR_UNSUPPORTED;
  }
  DBHIP_REQUIRE(dim > 0 && n >= 0 && n < 0xFFFFFFFFLL && (base || n == 0), "dbhip_vec_index_build: bad shape");
  dbhip_vec_index* ix = new (std::nothrow) dbhip_vec_index();
  DBHIP_REQUIRE(ix, "dbhip_vec_index_build: out of host memory");
  ix->metric = metric; ix->base = base; ix->n = n; ix->dim = dim;
  ix->dpad = (int)ceil_div(dim, HBK) * HBK;
  ix->bh = nullptr; ix->rowA = ix->rowX = ix->rowY = nullptr;
  const int64_t rows = n > 0 ? n : 1;
  int32_t rc;
  if ((rc = dbhip_alloc((size_t)rows * ix->dpad * 2, (void**)&ix->bh)) || (rc = dbhip_alloc((size_t)rows * 4, (void**)&ix->rowA)) ||
      (rc = dbhip_alloc((size_t)rows * 4, (void**)&ix->rowX)) || (rc = dbhip_alloc((size_t)rows * 4, (void**)&ix->rowY))) {
    dbhip_vec_index_destroy(ix);
    return rc;
  }
  if (n > 0) {
    hipStream_t s = resolve_stream(stream);
    hipLaunchKernelGGL(vec_to_bf16_kernel, dim3(grid_for(n * 64, 256)), dim3(256), 0, s, base, n, dim, ix->dpad,
                       metric == DBHIP_VEC_COSINE ? 0 : (metric == DBHIP_VEC_L2 ? 3 : 1), ix->bh, ix->rowA, ix->rowX, ix->rowY);
    DBHIP_LAUNCH_CHECK();
  }
  *out_host = ix;
  return DBHIP_OK;
}

int32_t dbhip_vec_index_search(dbhip_vec_index* ix, const float* queries, int32_t nq, int32_t k, uint32_t* out_idx,
                               float* out_dist, void* stream) {
  DBHIP_REQUIRE(ix && nq >= 0 && k >= 1, "dbhip_vec_index_search: bad argument");
  if (k > KMAX) {
    set_error("dbhip_vec_index_search: k=%d > %d", k, KMAX);
    return DBHIP_ERR_UNSUPPORTED;
  }
  if (nq == 0) return DBHIP_OK;
  DBHIP_REQUIRE(queries && out_idx && out_dist, "dbhip_vec_index_search: NULL argument");
  hipStream_t s = resolve_stream(stream);
  float* qnorm = nullptr;
  if (ix->metric == DBHIP_VEC_COSINE) {
    qnorm = (float*)scratch((size_t)(nq + 1) * 4, 5, s);
    if (!qnorm) return DBHIP_ERR_HIP;
    hipLaunchKernelGGL(row_norm_kernel, dim3(grid_for((int64_t)nq * 64, 256)), dim3(256), 0, s, queries, (int64_t)nq, ix->dim, qnorm);
  }
  const int QB = 2048;
  kernel_timer_start(s);
  for (int qb = 0; qb < nq; qb += QB) {
    DBHIP_POLL_CANCEL(s, "dbhip_vec_index_search");
    const int bn = nq - qb < QB ? nq - qb : QB;
    int32_t rc = index_search_batch(ix, queries + (int64_t)qb * ix->dim, bn, k, qnorm ? qnorm + qb : nullptr,
                                    out_idx + (int64_t)qb * k, out_dist + (int64_t)qb * k, s);
    if (rc) return rc;
  }
  kernel_timer_stop(s);
  return DBHIP_OK;
}

int32_t dbhip_vec_index_destroy(dbhip_vec_index* ix) {
  if (!ix) return DBHIP_OK;
  (void)hipDeviceSynchronize();
  if (ix->bh) (void)dbhip_free(ix->bh);
  if (ix->rowA) (void)dbhip_free(ix->rowA);
  if (ix->rowX) (void)dbhip_free(ix->rowX);
  if (ix->rowY) (void)dbhip_free(ix->rowY);
  delete ix;
  return DBHIP_OK;
}

int32_t dbhip_vec_distance(int32_t metric, const float* base, int64_t n, int32_t dim, const float* queries,
                           int32_t nq, float* out, void* stream) {
  DBHIP_REQUIRE(metric >= DBHIP_VEC_COSINE && metric <= DBHIP_VEC_L1, "dbhip_vec_distance: bad metric");
  DBHIP_REQUIRE(dim > 0 && nq >= 0 && n >= 0, "dbhip_vec_distance: bad shape");
  if (n == 0 || nq == 0) return DBHIP_OK;
  DBHIP_REQUIRE(base && queries && out, "dbhip_vec_distance: NULL argument");
  hipStream_t s = resolve_stream(stream);
  float* qnorm = nullptr;
  if (metric == DBHIP_VEC_COSINE) {
    qnorm = (float*)scratch((size_t)nq * 4, 5, s);
    if (!qnorm) return DBHIP_ERR_HIP;
    hipLaunchKernelGGL(row_norm_kernel, dim3(grid_for((int64_t)nq * 64, 256)), dim3(256), 0, s, queries, (int64_t)nq, dim, qnorm);
  }
  kernel_timer_start(s);
  int32_t rc = launch_distance(metric, base, n, dim, queries, nq, qnorm, out, n, s);
  kernel_timer_stop(s);
  return rc;
}

int32_t dbhip_vec_distance_rows(int32_t metric, int32_t elem_type, const void* lhs, int32_t lhs_is_scalar, const void* rhs, int32_t rhs_is_scalar,
                                int64_t n, int32_t dim, void* out, void* stream) {
  DBHIP_REQUIRE(metric >= DBHIP_VEC_COSINE && metric <= DBHIP_VEC_NORM, "dbhip_vec_distance_rows: bad metric");
  DBHIP_REQUIRE(dim > 0 && n >= 0, "dbhip_vec_distance_rows: bad shape");
  if (elem_type != DBHIP_T_F32 && elem_type != DBHIP_T_F64 && elem_type != DBHIP_T_I8) {
    set_error("dbhip_vec_distance_rows: element type %d (Float32, Float64 and Int8 vectors exist in the reference)", elem_type);
    return DBHIP_ERR_UNSUPPORTED;
  }
  if (n == 0) return DBHIP_OK;
  DBHIP_REQUIRE(lhs && out && (rhs || metric == DBHIP_VEC_NORM), "dbhip_vec_distance_rows: NULL argument");
  hipStream_t s = resolve_stream(stream);
  int gshift = 0;
  while (gshift < 6 && (4 << gshift) < dim) ++gshift;   // ~4 elements per lane, at most one wave per row
  const int64_t groups = 256 >> gshift;
  const dim3 grid((unsigned)grid_for(ceil_div(n, groups) * 256, 256)), blk(256);
  kernel_timer_start(s);
  if (elem_type == DBHIP_T_F32)
    hipLaunchKernelGGL((vec_rows_kernel<float, float>), grid, blk, 0, s, metric, (const float*)lhs, lhs_is_scalar, (const float*)rhs, rhs_is_scalar, n, dim, gshift, (float*)out);
  else if (elem_type == DBHIP_T_F64)
    hipLaunchKernelGGL((vec_rows_kernel<double, double>), grid, blk, 0, s, metric, (const double*)lhs, lhs_is_scalar, (const double*)rhs, rhs_is_scalar, n, dim, gshift, (double*)out);
  else
    hipLaunchKernelGGL((vec_rows_kernel<int8_t, float>), grid, blk, 0, s, metric, (const int8_t*)lhs, lhs_is_scalar, (const int8_t*)rhs, rhs_is_scalar, n, dim, gshift, (float*)out);
  kernel_timer_stop(s);
  DBHIP_LAUNCH_CHECK();
  return DBHIP_OK;
}

int32_t dbhip_vec_topk(int32_t metric, const float* base, int64_t n, int32_t dim, const float* queries,
                       int32_t nq, int32_t k, uint32_t* out_idx, float* out_dist, void* stream) {
  DBHIP_REQUIRE(metric >= DBHIP_VEC_COSINE && metric <= DBHIP_VEC_L1, "dbhip_vec_topk: bad metric");
  DBHIP_REQUIRE(dim > 0 && nq >= 0 && n >= 0 && k >= 1, "dbhip_vec_topk: bad shape");
  if (k > KMAX) {
    set_error("dbhip_vec_topk: k=%d > %d; compute distances and use dbhip_sort_perm with a limit", k, KMAX);
    return DBHIP_ERR_UNSUPPORTED;
  }
  DBHIP_REQUIRE(n < 0xFFFFFFFFLL, "dbhip_vec_topk: more than 2^32-1 base rows per shard");
  if (nq == 0) return DBHIP_OK;
  DBHIP_REQUIRE(queries && out_idx && out_dist && (base || n == 0), "dbhip_vec_topk: NULL argument");
  hipStream_t s = resolve_stream(stream);
  float* qnorm = nullptr;
  if (metric == DBHIP_VEC_COSINE) {
    qnorm = (float*)scratch((size_t)(nq + 1) * 4, 5, s);
    if (!qnorm) return DBHIP_ERR_HIP;
    hipLaunchKernelGGL(row_norm_kernel, dim3(grid_for((int64_t)nq * 64, 256)), dim3(256), 0, s, queries, (int64_t)nq, dim, qnorm);
  }
  // query batches bound the scratch of the sample pass (<= 1 GiB of scores)
  const int QB = 2048;
  kernel_timer_start(s);
  for (int qb = 0; qb < nq; qb += QB) {
    const int bn = nq - qb < QB ? nq - qb : QB;
    int32_t rc = topk_batch(metric, base, n, dim, queries + (int64_t)qb * dim, bn, k, qnorm ? qnorm + qb : nullptr,
                            out_idx + (int64_t)qb * k, out_dist + (int64_t)qb * k, s);
    if (rc) return rc;
  }
  kernel_timer_stop(s);
  return DBHIP_OK;
}

int32_t dbhip_vec_topk_merge(const float* dists, const uint32_t* ids, int64_t m, int32_t nq, int32_t k,
                             uint32_t* out_idx, float* out_dist, void* stream) {
  DBHIP_REQUIRE(nq >= 0 && m >= 0 && k >= 1, "dbhip_vec_topk_merge: bad shape");
  if (k > KMAX) {
    set_error("dbhip_vec_topk_merge: k=%d > %d", k, KMAX);
    return DBHIP_ERR_UNSUPPORTED;
  }
  if (nq == 0) return DBHIP_OK;
  DBHIP_REQUIRE(dists && ids && out_idx && out_dist, "dbhip_vec_topk_merge: NULL argument");
  return select_topk(dists, ids, nullptr, m, m, 0u, nq, k, false, out_dist, out_idx, resolve_stream(stream));
}

int32_t dbhip_score_u8(int32_t is_l1, const uint8_t* query, const uint8_t* base, int64_t n, int32_t dim,
                       float* out, void* stream) {
  if (n == 0) return DBHIP_OK;
  DBHIP_REQUIRE(query && base && out && dim > 0, "dbhip_score_u8: bad argument");
  if (dim % 16 == 0 && (((uintptr_t)base | (uintptr_t)query) & 15) == 0 && dim <= 16384)
    hipLaunchKernelGGL(score_u8_q16_kernel, dim3(grid_for(n * 16, 256)), dim3(256), (size_t)dim, resolve_stream(stream), query, base,
                       n, dim, is_l1, out);
  else
    hipLaunchKernelGGL(score_u8_kernel, dim3(grid_for(n * 64, 256)), dim3(256), 0, resolve_stream(stream), query, base,
                       n, dim, is_l1, out);
  DBHIP_LAUNCH_CHECK();
  return DBHIP_OK;
}

}  // extern "C"

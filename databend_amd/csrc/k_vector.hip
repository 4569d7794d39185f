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
#include "dev_common.h"
#include "runtime.h"

#include <math.h>

using namespace dbhip;

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

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
// MFMA tile kernel: 128 queries x 128 base rows per block, 4 waves as 2x2, each wave 64x64
// = 2x2 accumulators of v_mfma_f32_32x32x2_f32.
//   A operand (lane l): Q[q0 + (l&31)][k + (l>>5)]     B operand: Base[i0 + (l&31)][k + (l>>5)]
//   C/D reg r: col = l&31 (base row), row = (r&3) + 8*(r>>2) + 4*(l>>5) (query)
// ---------------------------------------------------------------------------
template <bool COSINE>
__global__ __launch_bounds__(256) void dot_mfma_kernel(const float* __restrict__ base, int64_t n, int dim,
                                                       const float* __restrict__ queries, int nq,
                                                       const float* __restrict__ bnorm,
                                                       const float* __restrict__ qnorm,
                                                       float* __restrict__ out, int64_t out_ld, int64_t i_origin) {
  __shared__ float As[128 * LDK];
  __shared__ float Bs[128 * LDK];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wq = (wave >> 1) * 64, wi = (wave & 1) * 64;
  const int64_t i0 = (int64_t)blockIdx.x * 128;
  const int q0 = blockIdx.y * 128;
  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  for (int k0 = 0; k0 < dim; k0 += BK) {
    // stage 128 x 32 floats of each operand: 4096 floats / 256 threads = 16 per thread, coalesced along k
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      int e = t * 256 + tid;
      int row = e >> 5, kk = e & 31;
      int k = k0 + kk;
      int q = q0 + row;
      int64_t i = i0 + row;
      As[row * LDK + kk] = (q < nq && k < dim) ? queries[(int64_t)q * dim + k] : 0.f;
      Bs[row * LDK + kk] = (i < n && k < dim) ? base[i * dim + k] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < BK; kk += 2) {
      const int kl = kk + (lane >> 5);
      float a0 = As[(wq + (lane & 31)) * LDK + kl];
      float a1 = As[(wq + 32 + (lane & 31)) * LDK + kl];
      float b0 = Bs[(wi + (lane & 31)) * LDK + kl];
      float b1 = Bs[(wi + 32 + (lane & 31)) * LDK + kl];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
    }
    __syncthreads();
  }
  // epilogue: lanes with consecutive l&31 write consecutive base rows of one query (coalesced)
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int64_t i = i0 + wi + b * 32 + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int q = q0 + wq + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (q < nq && i < n) {
          float v = acc[a][b][r];
          if (COSINE) v = 1.0f - v / (qnorm[q] * bnorm[i + i_origin]);
          out[(int64_t)q * out_ld + i] = v;
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
// top-k: one block per query over a chunk of distances; merges with the running best
// ---------------------------------------------------------------------------
constexpr int KMAX = 16;

__device__ __forceinline__ bool cand_less(float d1, uint32_t i1, float d2, uint32_t i2) {
  // ascending distance, NaN last, ties by lower row id
  bool n1 = d1 != d1, n2 = d2 != d2;
  if (n1 != n2) return n2;
  if (!n1 && d1 != d2) return d1 < d2;
  return i1 < i2;
}

__global__ __launch_bounds__(256) void topk_chunk_kernel(const float* __restrict__ dist, int64_t chunk_n,
                                                         int64_t ld, uint32_t row_origin, int k,
                                                         float* best_d, uint32_t* best_i, int have_prev) {
  const int q = blockIdx.x, tid = threadIdx.x;
  const float* d = dist + (int64_t)q * ld;
  float ld_[KMAX];
  uint32_t li[KMAX];
#pragma unroll
  for (int j = 0; j < KMAX; ++j) { ld_[j] = INFINITY; li[j] = 0xFFFFFFFFu; }
  auto push = [&](float v, uint32_t id) {
    if (!cand_less(v, id, ld_[KMAX - 1], li[KMAX - 1])) return;
#pragma unroll
    for (int j = KMAX - 1; j >= 0; --j) {
      bool here = (j == 0) || !cand_less(v, id, ld_[j - 1], li[j - 1]);
      if (cand_less(v, id, ld_[j], li[j])) {
        if (here) { ld_[j] = v; li[j] = id; }
        else { ld_[j] = ld_[j - 1]; li[j] = li[j - 1]; }
      }
    }
  };
  // NaN sorts after +inf: represent "empty" as (+inf, 0xFFFFFFFF) and let real NaNs displace empties
  for (int64_t i = tid; i < chunk_n; i += 256) {
    float v = d[i];
    push(v != v ? INFINITY : v, (v != v) ? (uint32_t)(row_origin + i) : (uint32_t)(row_origin + i));
  }
  if (have_prev && tid == 0)
    for (int j = 0; j < k; ++j) push(best_d[(int64_t)q * k + j], best_i[(int64_t)q * k + j]);
  // k rounds of block-wide argmin over the heads of the per-thread sorted lists
  __shared__ float sd[4];
  __shared__ uint32_t si[4];
  __shared__ int sw[4];
  __shared__ int winner;
  int head = 0;
  for (int round = 0; round < k; ++round) {
    float hd = INFINITY;
    uint32_t hi = 0xFFFFFFFFu;
#pragma unroll
    for (int j = 0; j < KMAX; ++j)
      if (j == head) { hd = ld_[j]; hi = li[j]; }
    if (head >= KMAX) { hd = INFINITY; hi = 0xFFFFFFFFu; }
    float bd = hd;
    uint32_t bi = hi;
    int bt = tid;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      float od = __shfl_xor(bd, off, 64);
      uint32_t oi = __shfl_xor(bi, off, 64);
      int ot = __shfl_xor(bt, off, 64);
      if (cand_less(od, oi, bd, bi) || (od == bd && oi == bi && ot < bt)) { bd = od; bi = oi; bt = ot; }
    }
    if ((tid & 63) == 0) { sd[tid >> 6] = bd; si[tid >> 6] = bi; sw[tid >> 6] = bt; }
    __syncthreads();
    if (tid == 0) {
      int w = 0;
      for (int x = 1; x < 4; ++x)
        if (cand_less(sd[x], si[x], sd[w], si[w])) w = x;
      winner = sw[w];
      best_d[(int64_t)q * k + round] = sd[w];
      best_i[(int64_t)q * k + round] = si[w];
    }
    __syncthreads();
    if (tid == winner) ++head;
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------
// u8-quantised scoring (cpp/avx2.c:45-139): one wave per base row, v_dot4-style packed dot
// ---------------------------------------------------------------------------
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

int32_t launch_distance(int metric, const float* base, int64_t n, int dim, const float* queries, int nq,
                        const float* bnorm, const float* qnorm, float* out, int64_t out_ld, int64_t i_origin,
                        hipStream_t s) {
  if (metric == DBHIP_VEC_DOT || metric == DBHIP_VEC_COSINE) {
    dim3 grid((unsigned)ceil_div(n, 128), (unsigned)ceil_div(nq, 128));
    if (metric == DBHIP_VEC_COSINE)
      hipLaunchKernelGGL(dot_mfma_kernel<true>, grid, dim3(256), 0, s, base, n, dim, queries, nq, bnorm, qnorm, out, out_ld, i_origin);
    else
      hipLaunchKernelGGL(dot_mfma_kernel<false>, grid, dim3(256), 0, s, base, n, dim, queries, nq, bnorm, qnorm, out, out_ld, i_origin);
  } else {
    dim3 grid((unsigned)ceil_div(n, 64), (unsigned)ceil_div(nq, 64));
    if (metric == DBHIP_VEC_L1)
      hipLaunchKernelGGL(diff_valu_kernel<true>, grid, dim3(256), 0, s, base, n, dim, queries, nq, out, out_ld);
    else
      hipLaunchKernelGGL(diff_valu_kernel<false>, grid, dim3(256), 0, s, base, n, dim, queries, nq, out, out_ld);
  }
  DBHIP_LAUNCH_CHECK();
  return DBHIP_OK;
}

}  // namespace

extern "C" {

int32_t dbhip_vec_distance(int32_t metric, const float* base, int64_t n, int32_t dim, const float* queries,
                           int32_t nq, float* out, void* stream) {
  DBHIP_REQUIRE(metric >= DBHIP_VEC_COSINE && metric <= DBHIP_VEC_L1, "dbhip_vec_distance: bad metric");
  DBHIP_REQUIRE(dim > 0 && nq >= 0 && n >= 0, "dbhip_vec_distance: bad shape");
  if (n == 0 || nq == 0) return DBHIP_OK;
  DBHIP_REQUIRE(base && queries && out, "dbhip_vec_distance: NULL argument");
  hipStream_t s = resolve_stream(stream);
  float* bnorm = nullptr;
  float* qnorm = nullptr;
  if (metric == DBHIP_VEC_COSINE) {
    bnorm = (float*)scratch((size_t)(n + nq) * 4, 5);
    if (!bnorm) return DBHIP_ERR_HIP;
    qnorm = bnorm + n;
    hipLaunchKernelGGL(row_norm_kernel, dim3(grid_for(n * 64, 256)), dim3(256), 0, s, base, n, dim, bnorm);
    hipLaunchKernelGGL(row_norm_kernel, dim3(grid_for((int64_t)nq * 64, 256)), dim3(256), 0, s, queries, (int64_t)nq, dim, qnorm);
  }
  return launch_distance(metric, base, n, dim, queries, nq, bnorm, qnorm, out, n, 0, s);
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
  // chunk so that the distance scratch stays <= ~1 GiB
  int64_t chunk = (int64_t)(1LL << 28) / (nq > 0 ? nq : 1);
  chunk = chunk < 4096 ? 4096 : chunk;
  chunk = (chunk / 128) * 128;
  if (chunk > n) chunk = ((n + 127) / 128) * 128;
  if (chunk < 128) chunk = 128;
  float* dist = (float*)scratch((size_t)chunk * nq * 4, 6);
  if (!dist) return DBHIP_ERR_HIP;
  float* bnorm = nullptr;
  float* qnorm = nullptr;
  if (metric == DBHIP_VEC_COSINE) {
    bnorm = (float*)scratch((size_t)(n + nq + 1) * 4, 5);
    if (!bnorm) return DBHIP_ERR_HIP;
    qnorm = bnorm + n;
    if (n) hipLaunchKernelGGL(row_norm_kernel, dim3(grid_for(n * 64, 256)), dim3(256), 0, s, base, n, dim, bnorm);
    hipLaunchKernelGGL(row_norm_kernel, dim3(grid_for((int64_t)nq * 64, 256)), dim3(256), 0, s, queries, (int64_t)nq, dim, qnorm);
  }
  bool first = true;
  kernel_timer_start(s);
  for (int64_t c0 = 0; c0 < n || first; c0 += chunk) {
    int64_t cn = n - c0 < chunk ? n - c0 : chunk;
    if (cn > 0) {
      int32_t rc = launch_distance(metric, base + c0 * dim, cn, dim, queries, nq, bnorm, qnorm, dist, chunk, c0, s);
      if (rc) return rc;
    }
    hipLaunchKernelGGL(topk_chunk_kernel, dim3(nq), dim3(256), 0, s, dist, cn > 0 ? cn : 0, chunk, (uint32_t)c0, k,
                       out_dist, out_idx, first ? 0 : 1);
    DBHIP_LAUNCH_CHECK();
    first = false;
    if (n == 0) break;
  }
  kernel_timer_stop(s);
  return DBHIP_OK;
}

int32_t dbhip_score_u8(int32_t is_l1, const uint8_t* query, const uint8_t* base, int64_t n, int32_t dim,
                       float* out, void* stream) {
  if (n == 0) return DBHIP_OK;
  DBHIP_REQUIRE(query && base && out && dim > 0, "dbhip_score_u8: bad argument");
  hipLaunchKernelGGL(score_u8_kernel, dim3(grid_for(n * 64, 256)), dim3(256), 0, resolve_stream(stream), query, base,
                     n, dim, is_l1, out);
  DBHIP_LAUNCH_CHECK();
  return DBHIP_OK;
}

}  // extern "C"

// k_kmeans.hip — the reference's vector-cluster KMeans and its f32 VectorDistanceKernel on the device (SURVEY §8f-4 / a18:
// src/query/storages/common/index/src/kmeans.rs, vector.rs:45-260; caller: fuse/.../transform_vector_cluster.rs, batches of at
// most 262,144 rows and 64 clusters).
//
// The algorithm is deterministic in the reference (fixed LCG seed, fixed summation orders), so the device reproduces it BIT FOR
// BIT — assignments and distances — by keeping every floating-point order the reference's production kernel (Avx: 8 fused
// multiply-add lanes over the first len - len % 8 elements, lanes summed left to right, tail summed left to right) and its
// sequential loops have:
//   distances     a (row, centroid) pair is one thread with the 8 lane accumulators in registers, elements in index order (assign:
//                 lanes of a wave = centroids, the row's element is wave-uniform, centroids stored transposed so the lanes' loads
//                 coalesce), or a group of 8 lanes = the 8 AVX lanes (kmeans++ and the final distances: 32-byte segments per row)
//   centroid sums the reference adds the rows of a cluster in row order, per dimension: one workgroup per cluster walks the
//                 assignments in row order, compacts its rows with ballots (order kept) and every thread owns dimensions
//   kmeans++      the running total of the minimum distances and the threshold scan are sequential f32 chains: one lane does
//                 them from LDS-staged chunks; the LCG draw is made on the host (both branches of the reference consume exactly
//                 one next_u64), so no host round trip is needed per centroid
// This file is compiled with -ffp-contract=off: a fused multiply-add happens exactly where fmaf() is written.
#include "dev_common.h"
#include "runtime.h"

#include <math.h>
#include <vector>

using namespace dbhip;

namespace {

enum { KM_L1 = 0, KM_L2 = 1, KM_DOT = 2 };
constexpr float KM_EPS = 1.1920929e-07f;

// correctly rounded f32 sqrt and division whatever the compiler's fast-math defaults: through f64 (53 bits >= 2 * 24 + 2, so the
// double rounding is innocuous)
__device__ __forceinline__ float km_sqrt(float x) { return (float)sqrt((double)x); }
__device__ __forceinline__ float km_div(float a, float b) { return (float)((double)a / (double)b); }

__device__ __forceinline__ float km_dot_distance(float dot) {   // normalize_dot_distance (kmeans.rs:366-373)
  const float d = 1.0f - dot;
  if (isfinite(d)) return d > 0.0f ? d : 0.0f;
  return 1.0f;
}
__device__ __forceinline__ float km_post_compare(int dt, float v) { return dt == KM_DOT ? km_dot_distance(v) : v; }
__device__ __forceinline__ float km_post_distance(int dt, float v) { return dt == KM_L2 ? km_sqrt(v) : (dt == KM_DOT ? km_dot_distance(v) : v); }
__device__ __forceinline__ float km_step(int dt, float a, float b, float acc) {   // one lane step of the Avx kernels
  if (dt == KM_DOT) return fmaf(a, b, acc);
  const float d = a - b;
  if (dt == KM_L2) return fmaf(d, d, acc);
  return acc + fabsf(d);
}
__device__ __forceinline__ float km_tail(int dt, float a, float b, float acc) {   // the scalar tail: never fused
  if (dt == KM_DOT) return acc + a * b;
  const float d = a - b;
  if (dt == KM_L2) return acc + d * d;
  return acc + fabsf(d);
}

// raw kernel value (dot / l2_squared / l1) of row `a` against `b`, by a group of 8 lanes (g = lane & 7 = the AVX lane); the
// result is valid in the group's lane 0
__device__ __forceinline__ float km_group8(int dt, const float* __restrict__ a, const float* __restrict__ b, int dim, int g) {
  const int m = dim - dim % 8;
  float v = 0.0f;
  for (int i = g; i < m; i += 8) v = km_step(dt, a[i], b[i], v);
  float s = 0.0f;
#pragma unroll
  for (int j = 0; j < 8; ++j) s = s + __shfl(v, (lane_id() & ~7) + j, 64);   // values.iter().sum(): left to right from 0.0
  float t = 0.0f;
  if (g == 0) for (int i = m; i < dim; ++i) t = km_tail(dt, a[i], b[i], t);
  return s + t;
}

// vector_norm (ndarray 0.15.6 unrolled_fold over the rounded squares) by ONE thread
__device__ float km_nd_norm(const float* a, int n) {
  float p[8] = {0, 0, 0, 0, 0, 0, 0, 0}, acc = 0.0f;
  int i = 0;
  for (; i + 8 <= n; i += 8)
#pragma unroll
    for (int j = 0; j < 8; ++j) { const float x = a[i + j] * a[i + j]; p[j] = p[j] + x; }
  acc = acc + (p[0] + p[4]); acc = acc + (p[1] + p[5]); acc = acc + (p[2] + p[6]); acc = acc + (p[3] + p[7]);
  for (; i < n; ++i) { const float x = a[i] * a[i]; acc = acc + x; }
  return km_sqrt(acc);
}

// normalize_vector of every row (vector_samples, transform_vector_cluster.rs:186-190): one thread per row
__global__ __launch_bounds__(256) void km_normalize_rows_kernel(const float* in, int64_t rows, int dim, float* out) {
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += (int64_t)gridDim.x * blockDim.x) {
    const float* a = in + r * dim;
    float* o = out + r * dim;
    const float norm = km_nd_norm(a, dim);
    if (norm <= KM_EPS) { for (int i = 0; i < dim; ++i) o[i] = a[i]; }
    else for (int i = 0; i < dim; ++i) o[i] = km_div(a[i], norm);
  }
}

// kmeans++ step: mind[r] = min(mind[r], compare(row r, centroid))
__global__ __launch_bounds__(256) void km_min_distance_kernel(int dt, const float* data, int64_t rows, int dim, const float* centroid, float* mind) {
  const int g = lane_id() & 7;
  const int64_t groups = ((int64_t)gridDim.x * blockDim.x) >> 3;
  const int64_t n_pad = (rows + 7) & ~7LL;   // whole waves stay convergent for the shuffles
  for (int64_t r = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 3; r < n_pad; r += groups) {
    const int64_t rr = r < rows ? r : rows - 1;
    const float d = km_post_compare(dt, km_group8(dt, data + rr * dim, centroid, dim, g));
    if (g == 0 && r < rows && d < mind[r]) mind[r] = d;
  }
}

// the sequential part of a kmeans++ step (kmeans.rs:262-286), one wave, lane 0 computing from LDS-staged chunks:
// total = sum of mind in row order; chosen = range_pick when total <= EPSILON or not finite, else the first row where
// rnd * total - (running sum) <= 0 (rows - 1 if none). Then centroid <- data[chosen].
__global__ __launch_bounds__(256) void km_pick_kernel(const float* mind, int64_t rows, float rnd, int64_t range_pick, const float* data, int dim,
                                                      float* centroid) {
  __shared__ float buf[4096];
  __shared__ int64_t chosen_s;
  __shared__ float total_s;
  float total = 0.0f;
  for (int64_t base = 0; base < rows; base += 4096) {
    const int64_t cnt = rows - base < 4096 ? rows - base : 4096;
    for (int i = threadIdx.x; i < cnt; i += 256) buf[i] = mind[base + i];
    __syncthreads();
    if (threadIdx.x == 0) for (int i = 0; i < cnt; ++i) total = total + buf[i];
    __syncthreads();
  }
  if (threadIdx.x == 0) total_s = total;
  __syncthreads();
  total = total_s;
  int64_t chosen;
  if (total <= KM_EPS || !isfinite(total)) {
    chosen = range_pick;
  } else {
    float threshold = rnd * total;
    bool found = false;
    chosen = rows - 1;
    for (int64_t base = 0; base < rows && !found; base += 4096) {
      const int64_t cnt = rows - base < 4096 ? rows - base : 4096;
      for (int i = threadIdx.x; i < cnt; i += 256) buf[i] = mind[base + i];
      __syncthreads();
      if (threadIdx.x == 0) {
        chosen_s = -1;
        for (int i = 0; i < cnt; ++i) {
          threshold = threshold - buf[i];
          if (threshold <= 0.0f) { chosen_s = base + i; break; }
        }
      }
      __syncthreads();
      if (chosen_s >= 0) { chosen = chosen_s; found = true; }
      __syncthreads();
    }
  }
  for (int i = threadIdx.x; i < dim; i += 256) centroid[i] = data[chosen * dim + i];
}

// centT[i][kpad] <- cent[c][i]
__global__ __launch_bounds__(256) void km_transpose_kernel(const float* cent, int k, int kpad, int dim, float* centT) {
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < (int64_t)dim * kpad; t += (int64_t)gridDim.x * blockDim.x) {
    const int i = (int)(t / kpad), c = (int)(t % kpad);
    centT[t] = c < k ? cent[(int64_t)c * dim + i] : 0.0f;
  }
}

// nearest_centroid for every row (kmeans.rs:293-307): a wave takes RW rows, its lanes the centroids (64 at a time); the first
// strict minimum in cluster order wins; (0, +inf) when no distance compares below +inf
constexpr int KM_RW = 4;
__global__ __launch_bounds__(256) void km_assign_kernel(int dt, const float* __restrict__ data, int64_t rows, int dim, const float* __restrict__ centT,
                                                        int k, int kpad, uint32_t* assign, float* dists, uint32_t* changed) {
  const int lane = lane_id();
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  const int m = dim - dim % 8;
  bool any_changed = false;
  for (int64_t r0 = wave * KM_RW; r0 < rows; r0 += nwaves * KM_RW) {
    float best_d[KM_RW];
    int best_c[KM_RW];
#pragma unroll
    for (int u = 0; u < KM_RW; ++u) { best_d[u] = INFINITY; best_c[u] = 0; }
    for (int c0 = 0; c0 < k; c0 += 64) {
      const int c = c0 + lane;   // (< kpad: the padded centroids are zeros and never selected)
      float acc[KM_RW][8];
#pragma unroll
      for (int u = 0; u < KM_RW; ++u)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[u][j] = 0.0f;
      for (int i = 0; i < m; i += 8) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float cv = centT[(int64_t)(i + j) * kpad + c];
#pragma unroll
          for (int u = 0; u < KM_RW; ++u) {
            const int64_t r = r0 + u < rows ? r0 + u : rows - 1;
            acc[u][j] = km_step(dt, data[r * dim + i + j], cv, acc[u][j]);
          }
        }
      }
#pragma unroll
      for (int u = 0; u < KM_RW; ++u) {
        const int64_t r = r0 + u < rows ? r0 + u : rows - 1;
        float s = 0.0f;
#pragma unroll
        for (int j = 0; j < 8; ++j) s = s + acc[u][j];
        float t = 0.0f;
        for (int i = m; i < dim; ++i) t = km_tail(dt, data[r * dim + i], centT[(int64_t)i * kpad + c], t);
        float d = km_post_compare(dt, s + t);
        if (!(c < k) || !(d < INFINITY)) d = INFINITY;   // NaN and +inf never win a strict `<` against the initial +inf
        // first strict minimum over the lanes: the smallest distance, the lowest cluster among equals
        float md = d;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) { const float o = __shfl_xor(md, off, 64); md = o < md ? o : md; }
        const uint64_t at = __ballot(d == md && md < INFINITY);
        if (at && md < best_d[u]) { best_d[u] = md; best_c[u] = c0 + (__ffsll((long long)at) - 1); }
      }
    }
    if (lane < KM_RW && r0 + lane < rows) {
      float bd = 0.0f; int bc = 0;
#pragma unroll
      for (int u = 0; u < KM_RW; ++u) if (lane == u) { bd = best_d[u]; bc = best_c[u]; }
      const int64_t r = r0 + lane;
      if (assign[r] != (uint32_t)bc) { any_changed = true; assign[r] = (uint32_t)bc; }
      dists[r] = bd;
    }
  }
  if (__ballot(any_changed) && lane == 0) atomicOr(changed, 1u);
}

// next centroid of cluster blockIdx.x (kmeans.rs:146-176): sum of its rows in ROW ORDER per dimension, scaled by 1 / count,
// normalised for Dot. Thread t owns dimensions t, t + 256, ... (dim <= 256 * KM_DPT). Empty clusters are left to km_empty_kernel.
constexpr int KM_DPT = 16;
__global__ __launch_bounds__(256) void km_update_kernel(int dt, const float* __restrict__ data, int64_t rows, int dim, const uint32_t* __restrict__ assign,
                                                        float* next, uint32_t* counts) {
  const int c = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  __shared__ uint32_t list[256];
  __shared__ uint32_t wcnt[4];
  __shared__ float vec[256 * KM_DPT];
  __shared__ float norm_s;
  float acc[KM_DPT];
#pragma unroll
  for (int q = 0; q < KM_DPT; ++q) acc[q] = 0.0f;
  uint32_t count = 0;
  for (int64_t base = 0; base < rows; base += 256) {
    const int64_t r = base + tid;
    const bool mine = r < rows && assign[r] == (uint32_t)c;
    const uint64_t b = __ballot(mine);
    if (lane == 0) wcnt[wave] = (uint32_t)__popcll(b);
    __syncthreads();
    uint32_t before = 0;
    for (int w = 0; w < wave; ++w) before += wcnt[w];
    const uint32_t nmatch = wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
    if (mine) list[before + (uint32_t)__popcll(b & ((1ULL << lane) - 1))] = (uint32_t)r;
    __syncthreads();
    for (uint32_t e = 0; e < nmatch; ++e) {
      const float* p = data + (int64_t)list[e] * dim;
#pragma unroll
      for (int q = 0; q < KM_DPT; ++q) {
        const int d = tid + 256 * q;
        if (d < dim) acc[q] = acc[q] + p[d];
      }
    }
    count += nmatch;
    __syncthreads();
  }
  if (tid == 0) counts[c] = count;
  if (count == 0) return;
  const float inv = km_div(1.0f, (float)count);
#pragma unroll
  for (int q = 0; q < KM_DPT; ++q) {
    const int d = tid + 256 * q;
    if (d < dim) { acc[q] = acc[q] * inv; vec[d] = acc[q]; }
  }
  if (dt == KM_DOT) {   // normalize_centroid -> normalize_vector
    __syncthreads();
    if (tid == 0) norm_s = km_nd_norm(vec, dim);
    __syncthreads();
    const float norm = norm_s;
    if (!(norm <= KM_EPS)) {
#pragma unroll
      for (int q = 0; q < KM_DPT; ++q) acc[q] = km_div(acc[q], norm);
    }
  }
#pragma unroll
  for (int q = 0; q < KM_DPT; ++q) {
    const int d = tid + 256 * q;
    if (d < dim) next[(int64_t)c * dim + d] = acc[q];
  }
}

// empty clusters take the row with the LAST maximal distance (Iterator::max_by keeps the later of equals; a NaN replaces and
// is replaced): rare, one lane scans sequentially only when some count is 0
__global__ __launch_bounds__(256) void km_empty_kernel(const float* data, int64_t rows, int dim, const float* dists, const uint32_t* counts, int k, float* next) {
  __shared__ int64_t far_s;
  __shared__ int any_s;
  if (threadIdx.x == 0) {
    int any = 0;
    for (int c = 0; c < k; ++c) any |= counts[c] == 0;
    any_s = any;
    if (any) {
      int64_t far = 0;
      for (int64_t i = 1; i < rows; ++i) if (!(dists[far] > dists[i])) far = i;
      far_s = far;
    }
  }
  __syncthreads();
  if (!any_s) return;
  for (int c = 0; c < k; ++c)
    if (counts[c] == 0)
      for (int i = threadIdx.x; i < dim; i += 256) next[(int64_t)c * dim + i] = data[far_s * dim + i];
}

// shift = sum over clusters (in order) of sqrt(l2_squared(old, new)); ctl[0] = stop (nothing changed, or shift <= 1e-4);
// resets the changed flag; cent <- next
__global__ __launch_bounds__(256) void km_shift_kernel(float* cent, const float* next, int k, int dim, uint32_t* changed, uint32_t* ctl, float* sq) {
  const int g = lane_id() & 7;
  const int kpad8 = (k + 7) & ~7;
  for (int c = threadIdx.x >> 3; c < kpad8 + 32; c += 32) {   // 32 groups of 8 lanes per workgroup; whole waves stay convergent
    const int cc = c < k ? c : k - 1;
    const float v = km_group8(KM_L2, cent + (int64_t)cc * dim, next + (int64_t)cc * dim, dim, g);
    if (g == 0 && c < k) sq[c] = km_sqrt(v);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float shift = 0.0f;
    for (int c = 0; c < k; ++c) shift = shift + sq[c];
    ctl[0] = (!changed[0] || shift <= 1e-4f) ? 1u : 0u;
    ctl[1] += 1;   // iterations
    changed[0] = 0;
  }
  __syncthreads();
  for (int64_t i = threadIdx.x; i < (int64_t)k * dim; i += 256) cent[i] = next[i];
}

// build_result (kmeans.rs:207-247): distance of every row to its own centroid
__global__ __launch_bounds__(256) void km_result_kernel(int dt, const float* data, int64_t rows, int dim, const float* cent, const uint32_t* assign, float* out) {
  const int g = lane_id() & 7;
  const int64_t groups = ((int64_t)gridDim.x * blockDim.x) >> 3;
  const int64_t n_pad = (rows + 7) & ~7LL;
  for (int64_t r = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 3; r < n_pad; r += groups) {
    const int64_t rr = r < rows ? r : rows - 1;
    const float v = km_group8(dt, data + rr * dim, cent + (int64_t)assign[rr] * dim, dim, g);
    if (g == 0 && r < rows) out[r] = km_post_distance(dt, v);
  }
}

__global__ __launch_bounds__(256) void km_fill_f32_kernel(float* p, int64_t n, float v) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = v;
}
__global__ __launch_bounds__(256) void km_vdk_kernel(int which, const float* a, const float* b, int64_t n, int dim, float* out) {
  const int g = lane_id() & 7;
  const int dt = which == 0 ? KM_DOT : (which == 1 ? KM_L2 : KM_L1);
  const int64_t groups = ((int64_t)gridDim.x * blockDim.x) >> 3;
  const int64_t n_pad = (n + 7) & ~7LL;
  for (int64_t r = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 3; r < n_pad; r += groups) {
    const int64_t rr = r < n ? r : n - 1;
    const float v = km_group8(dt, a + rr * dim, b + rr * dim, dim, g);
    if (g == 0 && r < n) out[r] = v;
  }
}

struct Lcg {   // kmeans.rs:35-77
  uint64_t state;
  explicit Lcg(uint64_t seed) : state(seed > 1 ? seed : 1) {}
  uint64_t next() { state = state * 6364136223846793005ULL + 1442695040888963407ULL; return state; }
};

}  // namespace

extern "C" {

int32_t dbhip_vec_kernel_f32(int32_t which, const float* a, const float* b, int64_t n, int32_t dim, float* out, void* stream) {
  DBHIP_REQUIRE(which >= 0 && which <= 2 && dim >= 1 && (n == 0 || (a && b && out)), "dbhip_vec_kernel_f32: bad argument");
  if (n == 0) return DBHIP_OK;
  hipLaunchKernelGGL(km_vdk_kernel, dim3(grid_for(n * 8, 256)), dim3(256), 0, resolve_stream(stream), which, a, b, n, dim, out);
  DBHIP_LAUNCH_CHECK();
  return DBHIP_OK;
}

int32_t dbhip_kmeans(int32_t distance_type, const float* data, int64_t rows, int32_t dim, int64_t rows_per_cluster, int32_t normalize_input,
                     uint32_t* out_assignments, float* out_distances, int64_t* out_k_host, int32_t* out_iterations_host, void* stream) {
  DBHIP_REQUIRE(distance_type >= KM_L1 && distance_type <= KM_DOT, "dbhip_kmeans: distance type 0 (L1), 1 (L2) or 2 (Dot)");
  DBHIP_REQUIRE(data && rows >= 1 && dim >= 1 && rows_per_cluster >= 1 && out_assignments && out_distances, "dbhip_kmeans: bad argument");
  if (dim > 256 * KM_DPT) { set_error("dbhip_kmeans: dimension %d beyond %d", dim, 256 * KM_DPT); return DBHIP_ERR_UNSUPPORTED; }
  hipStream_t s = resolve_stream(stream);
  int64_t k = (rows + rows_per_cluster - 1) / rows_per_cluster;   // rows.div_ceil(rows_per_cluster).clamp(1, rows)
  if (k < 1) k = 1;
  if (k > rows) k = rows;
  if (out_k_host) *out_k_host = k;
  if (out_iterations_host) *out_iterations_host = 0;
  if (k <= 1) {
    DBHIP_CHECK(hipMemsetAsync(out_assignments, 0, (size_t)rows * 4, s));
    DBHIP_CHECK(hipMemsetAsync(out_distances, 0, (size_t)rows * 4, s));
    return DBHIP_OK;
  }
  if (k > 65536) { set_error("dbhip_kmeans: %lld clusters (the reference clusters batches of at most 64)", (long long)k); return DBHIP_ERR_UNSUPPORTED; }
  const int kk = (int)k, kpad = (kk + 63) & ~63;
  // scratch: [normalised copy] cent, next, centT, mind / dists, counts, sq, ctl
  const size_t nd = (size_t)rows * dim, kd = (size_t)kk * dim;
  const size_t bytes = (normalize_input ? nd * 4 : 0) + kd * 4 * 2 + (size_t)dim * kpad * 4 + (size_t)rows * 4 + (size_t)kk * 8 + 256;
  uint8_t* ws = (uint8_t*)scratch(bytes + 256, 15, s);
  if (!ws) return DBHIP_ERR_HIP;
  float* p = (float*)ws;
  const float* x = data;
  if (normalize_input) {
    hipLaunchKernelGGL(km_normalize_rows_kernel, dim3(grid_for(rows, 256)), dim3(256), 0, s, data, rows, dim, p);
    x = p;
    p += nd;
  }
  float* cent = p; p += kd;
  float* next = p; p += kd;
  float* centT = p; p += (size_t)dim * kpad;
  float* mind = p; p += rows;          // kmeans++: running minimum; iterations: distance to the nearest centroid
  float* sq = p; p += kk;
  uint32_t* counts = (uint32_t*)p; p += kk;
  uint32_t* ctl = (uint32_t*)p;        // [0] stop, [1] iterations, [2] changed
  uint32_t* changed = ctl + 2;
  // ---- choose_initial_centroids (kmeans.rs:249-291) ----
  Lcg rng(0xD47ABA5EC1A57E12ULL);
  const int64_t first = (int64_t)(rng.next() % (uint64_t)rows);
  DBHIP_CHECK(hipMemcpyAsync(cent, x + (size_t)first * dim, (size_t)dim * 4, hipMemcpyDeviceToDevice, s));
  hipLaunchKernelGGL(km_fill_f32_kernel, dim3(grid_for(rows, 256)), dim3(256), 0, s, mind, rows, INFINITY);
  const int grid8 = grid_for(rows * 8, 256);
  for (int c = 1; c < kk; ++c) {
    hipLaunchKernelGGL(km_min_distance_kernel, dim3(grid8), dim3(256), 0, s, distance_type, x, rows, dim, cent + (size_t)(c - 1) * dim, mind);
    const uint64_t v = rng.next();   // ONE draw either way: gen_range(rows) when the total is degenerate, else next_f32()
    const float rnd = (float)(v >> 40) / (float)(1ULL << 24);
    hipLaunchKernelGGL(km_pick_kernel, dim3(1), dim3(256), 0, s, mind, rows, rnd, (int64_t)(v % (uint64_t)rows), x, dim, cent + (size_t)c * dim);
  }
  DBHIP_LAUNCH_CHECK();
  // ---- compute_kmeans (kmeans.rs:121-205) ----
  DBHIP_CHECK(hipMemsetAsync(out_assignments, 0xFF, (size_t)rows * 4, s));   // usize::MAX
  DBHIP_CHECK(hipMemsetAsync(ctl, 0, 16, s));
  int iterations = 0;
  for (int it = 0; it < 100; ++it) {
    hipLaunchKernelGGL(km_transpose_kernel, dim3(grid_for((int64_t)dim * kpad, 256)), dim3(256), 0, s, cent, kk, kpad, dim, centT);
    hipLaunchKernelGGL(km_assign_kernel, dim3(grid_for(ceil_div(rows, KM_RW) * 64, 256)), dim3(256), 0, s, distance_type, x, rows, dim, centT, kk, kpad,
                       out_assignments, mind, changed);
    hipLaunchKernelGGL(km_update_kernel, dim3(kk), dim3(256), 0, s, distance_type, x, rows, dim, out_assignments, next, counts);
    hipLaunchKernelGGL(km_empty_kernel, dim3(1), dim3(256), 0, s, x, rows, dim, mind, counts, kk, next);
    hipLaunchKernelGGL(km_shift_kernel, dim3(1), dim3(256), 0, s, cent, next, kk, dim, changed, ctl, sq);
    DBHIP_LAUNCH_CHECK();
    uint32_t h[2] = {0, 0};
    DBHIP_CHECK(hipMemcpyAsync(h, ctl, 8, hipMemcpyDeviceToHost, s));
    DBHIP_CHECK(hipStreamSynchronize(s));
    iterations = (int)h[1];
    if (h[0]) break;
  }
  hipLaunchKernelGGL(km_result_kernel, dim3(grid8), dim3(256), 0, s, distance_type, x, rows, dim, cent, out_assignments, out_distances);
  DBHIP_LAUNCH_CHECK();
  if (out_iterations_host) *out_iterations_host = iterations;
  DBHIP_CHECK(hipStreamSynchronize(s));   // the scratch (normalised copy, centroids) is reused by the next call
  return DBHIP_OK;
}

}  // extern "C"

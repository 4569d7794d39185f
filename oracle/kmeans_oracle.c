/*
 * kmeans_oracle.c — CPU restatement of the reference's vector-cluster KMeans and its f32 distance kernel.
 * TEST INFRASTRUCTURE ONLY (see oracle.h). Compiled with -ffp-contract=off: every fused multiply-add below is an explicit
 * fmaf() exactly where the reference's AVX2 path has _mm256_fmadd_ps, nothing else is fused.
 *
 * Reference (src/query/storages/common/index/src):
 *   vector.rs:45-160     VectorDistanceKernel::{dot, l2_squared, l1}; the kernel the production target picks is Avx
 *                        (detect_vector_distance_kernel_impl :128-151: x86_64 with avx2 + fma)
 *   vector.rs:190-260    impl_f32_dot_avx / impl_f32_l2_sqr_avx / impl_f32_l1_avx: 8 lanes over the first len - len % 8
 *                        elements (fmadd for dot and l2, add for l1), the 8 lanes summed left to right, plus the tail summed
 *                        left to right
 *   vector.rs:35-43      normalize_vector (vector_norm = sqrt of ndarray's sum of squares: products rounded first,
 *                        ndarray 0.15.6 unrolled_fold order — 8 partial sums, (p0+p4)+(p1+p5)+(p2+p6)+(p3+p7), tail)
 *   kmeans.rs:27-77      constants and the LCG (seed 0xD47ABA5EC1A57E12, multiplier 6364136223846793005, increment
 *                        1442695040888963407; next_f32 = (state >> 40) / 2^24; gen_range = state % upper)
 *   kmeans.rs:93-205     compute / compute_kmeans (assign: first strict minimum; centroid sums in row order; empty cluster ->
 *                        the row with the LAST maximal distance (Iterator::max_by keeps the later of equals); scale by
 *                        1 / count; Dot: normalise; shift = sum of sqrt(l2_squared(old, new)); stop when nothing changed or
 *                        shift <= 1e-4; at most 100 iterations)
 *   kmeans.rs:207-247    build_result (distance of every row to its centroid)
 *   kmeans.rs:249-291    choose_initial_centroids (kmeans++: sequential f32 total of the running minimum distances, threshold =
 *                        next_f32() * total, first index where the running difference reaches <= 0)
 *   kmeans.rs:322-373    KMeansDistanceKernel::{compare, distance, normalize_centroid}, normalize_dot_distance
 * "Parity unpinned": the reference has no test with known answers for KMeans; determinism (fixed seed, fixed summation orders)
 * is what makes a literal restatement meaningful.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "oracle.h"

enum { KM_L1 = 0, KM_L2 = 1, KM_DOT = 2 };

static float avx_dot(const float* a, const float* b, int n) {
  int m = n - n % 8;
  float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int i = 0; i < m; i += 8) for (int j = 0; j < 8; ++j) v[j] = fmaf(a[i + j], b[i + j], v[j]);
  float s = 0.0f;
  for (int j = 0; j < 8; ++j) s = s + v[j];
  float t = 0.0f;
  for (int i = m; i < n; ++i) t = t + a[i] * b[i];
  return s + t;
}
static float avx_l2_squared(const float* a, const float* b, int n) {
  int m = n - n % 8;
  float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int i = 0; i < m; i += 8) for (int j = 0; j < 8; ++j) { float d = a[i + j] - b[i + j]; v[j] = fmaf(d, d, v[j]); }
  float s = 0.0f;
  for (int j = 0; j < 8; ++j) s = s + v[j];
  float t = 0.0f;
  for (int i = m; i < n; ++i) { float d = a[i] - b[i]; t = t + d * d; }
  return s + t;
}
static float avx_l1(const float* a, const float* b, int n) {
  int m = n - n % 8;
  float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int i = 0; i < m; i += 8) for (int j = 0; j < 8; ++j) v[j] = v[j] + fabsf(a[i + j] - b[i + j]);
  float s = 0.0f;
  for (int j = 0; j < 8; ++j) s = s + v[j];
  float t = 0.0f;
  for (int i = m; i < n; ++i) t = t + fabsf(a[i] - b[i]);
  return s + t;
}
float orc_vdk(int which, const float* a, const float* b, int n) { /* 0 dot, 1 l2_squared, 2 l1 */
  return which == 0 ? avx_dot(a, b, n) : (which == 1 ? avx_l2_squared(a, b, n) : avx_l1(a, b, n));
}
static float nd_norm(const float* a, int n) { /* vector_norm: (&a * &a).sum().sqrt() */
  float p[8] = {0, 0, 0, 0, 0, 0, 0, 0}, acc = 0.0f;
  int i = 0;
  for (; i + 8 <= n; i += 8) for (int j = 0; j < 8; ++j) { float x = a[i + j] * a[i + j]; p[j] = p[j] + x; }
  acc = acc + (p[0] + p[4]); acc = acc + (p[1] + p[5]); acc = acc + (p[2] + p[6]); acc = acc + (p[3] + p[7]);
  for (; i < n; ++i) { float x = a[i] * a[i]; acc = acc + x; }
  return sqrtf(acc);
}
void orc_normalize_vector(float* v, int n) {
  float norm = nd_norm(v, n);
  if (norm <= 1.1920929e-07f) return; /* f32::EPSILON */
  for (int i = 0; i < n; ++i) v[i] = v[i] / norm;
}
static float dot_distance(float dot) {
  float d = 1.0f - dot;
  if (isfinite(d)) return d > 0.0f ? d : 0.0f; /* distance.max(0.0) */
  return 1.0f;
}
static float km_compare(int dt, const float* p, const float* c, int dim) {
  if (dt == KM_L1) return avx_l1(p, c, dim);
  if (dt == KM_L2) return avx_l2_squared(p, c, dim);
  return dot_distance(avx_dot(p, c, dim));
}
static float km_distance(int dt, const float* p, const float* c, int dim) {
  if (dt == KM_L1) return avx_l1(p, c, dim);
  if (dt == KM_L2) return sqrtf(avx_l2_squared(p, c, dim));
  return dot_distance(avx_dot(p, c, dim));
}

typedef struct { uint64_t state; } lcg;
static void lcg_init(lcg* r, uint64_t seed) { r->state = seed > 1 ? seed : 1; }
static uint64_t lcg_next(lcg* r) { r->state = r->state * 6364136223846793005ULL + 1442695040888963407ULL; return r->state; }
static float lcg_f32(lcg* r) { uint64_t v = lcg_next(r) >> 40; return (float)v / (float)(1ULL << 24); }
static uint64_t lcg_range(lcg* r, uint64_t upper) { return lcg_next(r) % upper; }

/* KMeans::compute. data: rows x dim (already normalised by the caller for Dot, like vector_samples does).
 * -> k; assignments (u32), distances (f32), *iterations. */
int64_t orc_kmeans(int distance_type, const float* data, int64_t rows, int dim, int64_t rows_per_cluster, uint32_t* assign,
                   float* dist_out, int* iterations_out) {
  const uint64_t SEED = 0xD47ABA5EC1A57E12ULL;
  if (rows <= 0 || dim <= 0 || rows_per_cluster <= 0) return -1;
  int64_t k = (rows + rows_per_cluster - 1) / rows_per_cluster;
  if (k < 1) k = 1;
  if (k > rows) k = rows;
  *iterations_out = 0;
  if (k <= 1) {
    for (int64_t i = 0; i < rows; ++i) { assign[i] = 0; dist_out[i] = 0.0f; }
    return 1;
  }
  float* cent = (float*)calloc((size_t)k * dim, sizeof(float));
  float* next = (float*)malloc((size_t)k * dim * sizeof(float));
  float* mind = (float*)malloc((size_t)rows * sizeof(float));
  float* dists = (float*)calloc((size_t)rows, sizeof(float));
  int64_t* counts = (int64_t*)malloc((size_t)k * sizeof(int64_t));
  /* choose_initial_centroids */
  lcg rng;
  lcg_init(&rng, SEED);
  int64_t first = (int64_t)lcg_range(&rng, (uint64_t)rows);
  memcpy(cent, data + (size_t)first * dim, (size_t)dim * sizeof(float));
  for (int64_t i = 0; i < rows; ++i) mind[i] = INFINITY;
  for (int64_t c = 1; c < k; ++c) {
    const float* last = cent + (size_t)(c - 1) * dim;
    float total = 0.0f;
    for (int64_t i = 0; i < rows; ++i) {
      float d = km_compare(distance_type, data + (size_t)i * dim, last, dim);
      if (d < mind[i]) mind[i] = d;
      total = total + mind[i];
    }
    int64_t chosen;
    if (total <= 1.1920929e-07f || !isfinite(total)) chosen = (int64_t)lcg_range(&rng, (uint64_t)rows);
    else {
      float threshold = lcg_f32(&rng) * total;
      chosen = rows - 1;
      for (int64_t i = 0; i < rows; ++i) {
        threshold = threshold - mind[i];
        if (threshold <= 0.0f) { chosen = i; break; }
      }
    }
    memcpy(cent + (size_t)c * dim, data + (size_t)chosen * dim, (size_t)dim * sizeof(float));
  }
  /* compute_kmeans */
  for (int64_t i = 0; i < rows; ++i) assign[i] = 0xFFFFFFFFu; /* usize::MAX */
  lcg rng2;
  lcg_init(&rng2, SEED ^ 0x9e3779b97f4a7c15ULL);
  int iterations = 0;
  for (int it = 0; it < 100; ++it) {
    ++iterations;
    int changed = 0;
    for (int64_t i = 0; i < rows; ++i) {
      int64_t best = 0;
      float bd = INFINITY;
      for (int64_t c = 0; c < k; ++c) {
        float d = km_compare(distance_type, data + (size_t)i * dim, cent + (size_t)c * dim, dim);
        if (d < bd) { best = c; bd = d; }
      }
      if (assign[i] != (uint32_t)best) { changed = 1; assign[i] = (uint32_t)best; }
      dists[i] = bd;
    }
    memset(next, 0, (size_t)k * dim * sizeof(float));
    memset(counts, 0, (size_t)k * sizeof(int64_t));
    for (int64_t i = 0; i < rows; ++i) {
      counts[assign[i]] += 1;
      float* cn = next + (size_t)assign[i] * dim;
      const float* p = data + (size_t)i * dim;
      for (int d = 0; d < dim; ++d) cn[d] = cn[d] + p[d];
    }
    for (int64_t c = 0; c < k; ++c) {
      float* cn = next + (size_t)c * dim;
      if (counts[c] == 0) {
        /* Iterator::max_by with partial_cmp(..).unwrap_or(Equal): the accumulated element is replaced unless it compares
         * Greater than the new one (so the later of equal maxima wins, and an incomparable NaN replaces / is replaced) */
        int64_t far = 0;
        for (int64_t i = 1; i < rows; ++i) if (!(dists[far] > dists[i])) far = i;
        memcpy(cn, data + (size_t)far * dim, (size_t)dim * sizeof(float));
        continue;
      }
      float inv = 1.0f / (float)counts[c];
      for (int d = 0; d < dim; ++d) cn[d] = cn[d] * inv;
      if (distance_type == KM_DOT) orc_normalize_vector(cn, dim);
    }
    float shift = 0.0f;
    for (int64_t c = 0; c < k; ++c) shift = shift + sqrtf(avx_l2_squared(cent + (size_t)c * dim, next + (size_t)c * dim, dim));
    memcpy(cent, next, (size_t)k * dim * sizeof(float));
    if (!changed || shift <= 1e-4f) break;
  }
  (void)rng2; /* the fallback `rng.gen_range(rows)` of max_by's None is unreachable for rows >= 1 */
  for (int64_t i = 0; i < rows; ++i) dist_out[i] = km_distance(distance_type, data + (size_t)i * dim, cent + (size_t)assign[i] * dim, dim);
  *iterations_out = iterations;
  free(cent); free(next); free(mind); free(dists); free(counts);
  return k;
}

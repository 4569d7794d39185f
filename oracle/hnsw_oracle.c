/* oracle/hnsw_oracle.c — CPU restatement of the reference's HNSW vector index (TEST INFRASTRUCTURE: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; the product never does).
 *
 * Restated from /root/reference/src/query/storages/common/index/src/hnsw_index/:
 *   quantization/encoded_vectors_u8.rs:54-215,301-413   EncodedVectorsU8::{encode, encode_query, score_point, score_internal}
 *   quantization/quantile.rs:24-38                       find_min_max_from_iter
 *   hnsw.rs:62-118,142-315,317-374                       HNSWIndex::{search, build, postprocess_score}, cosine_preprocess
 *   graph_layers.rs:72-175,218-247                       _search_on_level, search_on_level, search_entry, search
 *   graph_layers_builder.rs:246-262,300-341,343-520      get_random_layer, heuristic selection, link_new_point(_on_level),
 *                                                        link_with_heuristic
 *   entry_points.rs:56-120                               EntryPoints::{new_point, get_entry_point}
 *   search_context.rs:30-61, common/fixed_length_priority_queue.rs:52-66, common/types.rs:38-48
 *   point_scorer.rs:47-70,133-174                        RawScorer::{Original, Quantized}
 *
 * The priority queues are Rust's std::collections::BinaryHeap; ScoredPointOffset orders by score ONLY (types.rs:38-42), so
 * which of two equal scores leaves a heap first is decided by std's sift order. The std sources are not under /root/reference
 * (toolchain library, rust 1.9x `alloc/src/collections/binary_heap/mod.rs`); its published algorithm is restated below
 * (push = sift_up; pop = swap last into the root, sift_down_to_bottom, sift_up; PeekMut write = sift_down; into_sorted_vec
 * = repeated swap + sift_down_range), so ties resolve as they do in the reference.
 *
 * Pinning: tests/test_hnsw_cpu.py checks the quantiser + scoring + postprocess against the reference's own known answers
 * (tests/sqllogictests/suites/query/index/09_vector_index/09_0000_vector_index_base.test:60-335, table `t` / `t_native` =
 * the HNSW + u8 path; every block holds <= 8 vectors, so the graph is complete and the printed distances are exactly the
 * quantised scores). The graph BUILD of the reference is not reproducible (levels from thread_rng(), parallel inserts,
 * hnsw.rs:158-235): "parity unpinned" for the graph itself — the restatement is checked by structure and recall.
 *
 * Compiled with -ffp-contract=off: Rust never fuses a*b+c. */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_HNSW_DOT 0 /* cosine: vectors pre-normalised, score = dot (hnsw.rs:367-374)            */
#define ORC_HNSW_L1 1
#define ORC_HNSW_L2 2
#define ALIGNMENT 16 /* encoded_vectors_u8.rs:33 */

/* ------------------------------------------------------------------------------------------------------------------
 * cosine_preprocess (hnsw.rs:362-374): length = sequential f32 sum of x*x; untouched when ~0 or ~1
 * ---------------------------------------------------------------------------------------------------------------- */
void orc_cosine_preprocess(const float* v, int dim, float* out) {
  float length = 0.0f;
  for (int i = 0; i < dim; ++i) length += v[i] * v[i];
  const int keep = length < 1.1920929e-7f /* f32::EPSILON */ || fabsf(length - 1.0f) <= 1.0e-6f;
  if (keep) {
    for (int i = 0; i < dim; ++i) out[i] = v[i];
    return;
  }
  length = sqrtf(length);
  for (int i = 0; i < dim; ++i) out[i] = v[i] / length;
}

/* ------------------------------------------------------------------------------------------------------------------
 * EncodedVectorsU8
 * ---------------------------------------------------------------------------------------------------------------- */
typedef struct {
  int dim, actual_dim, distance, invert;
  int64_t count;
  float alpha, offset, multiplier;
} orc_u8_meta;

int orc_u8_actual_dim(int dim) { return dim + (ALIGNMENT - dim % ALIGNMENT) % ALIGNMENT; } /* :286-288 */

static uint8_t f32_to_u8(float i, float alpha, float offset) { /* :243-246; `as u8` saturates, NaN -> 0 */
  float x = (i - offset) / alpha;
  if (x != x) return 0;
  if (x < 0.0f) x = 0.0f;
  if (x > 127.0f) x = 127.0f;
  return (uint8_t)x;
}

/* find_alpha_offset_size_dim (:231-241) over ALL values of the (pre-processed) vectors; encode(.., quantile = None) */
void orc_u8_params(const float* data, int64_t n, int dim, int distance, orc_u8_meta* m) {
  float mn = 3.4028235e38f, mx = -3.4028235e38f; /* (f32::MAX, f32::MIN) */
  for (int64_t i = 0; i < n * dim; ++i) {
    if (data[i] < mn) mn = data[i];
    if (data[i] > mx) mx = data[i];
  }
  m->dim = dim;
  m->actual_dim = orc_u8_actual_dim(dim);
  m->distance = distance;
  m->invert = distance != ORC_HNSW_DOT; /* hnsw.rs:77-80 */
  m->count = n;
  if (n == 0) { m->alpha = 0; m->offset = 0; m->multiplier = 0; return; }
  m->alpha = (mx - mn) / 127.0f;
  m->offset = mn;
  float mult = distance == ORC_HNSW_DOT ? m->alpha * m->alpha : distance == ORC_HNSW_L1 ? m->alpha : -2.0f * m->alpha * m->alpha; /* :150-154 */
  m->multiplier = m->invert ? -mult : mult;
}

static float vector_offset_of(const uint8_t* codes, const orc_u8_meta* m) { /* :118-137 */
  float vo = 0.0f;
  if (m->distance == ORC_HNSW_DOT) {
    float s = 0.0f;
    for (int i = 0; i < m->actual_dim; ++i) s += (float)codes[i];
    vo = (float)m->actual_dim * m->offset * m->offset + s * m->alpha * m->offset;
  } else if (m->distance == ORC_HNSW_L2) {
    float s = 0.0f;
    for (int i = 0; i < m->actual_dim; ++i) s += (float)codes[i] * (float)codes[i];
    vo = (float)m->actual_dim * m->offset * m->offset + s * m->alpha * m->alpha;
  }
  return m->invert ? -vo : vo;
}

/* encode (:95-146): out = n records of (4-byte f32 vector offset, actual_dim codes) */
void orc_u8_encode(const float* data, const orc_u8_meta* m, uint8_t* out) {
  const int rec = m->actual_dim + 4;
  for (int64_t v = 0; v < m->count; ++v) {
    uint8_t* r = out + v * rec;
    for (int i = 0; i < m->dim; ++i) r[4 + i] = f32_to_u8(data[v * m->dim + i], m->alpha, m->offset);
    const float placeholder = m->distance == ORC_HNSW_DOT ? 0.0f : m->offset; /* :107-113 */
    for (int i = m->dim; i < m->actual_dim; ++i) r[4 + i] = f32_to_u8(placeholder, m->alpha, m->offset);
    /* NB (:118-137): the reference sums over `encoded_vector`, which at that point still starts with the four zero bytes of
     * the placeholder offset — zeros change neither sum */
    const float vo = vector_offset_of(r + 4, m);
    memcpy(r, &vo, 4);
  }
}

/* encode_query (:317-366) */
float orc_u8_encode_query(const float* q, const orc_u8_meta* m, uint8_t* codes) {
  for (int i = 0; i < m->dim; ++i) codes[i] = f32_to_u8(q[i], m->alpha, m->offset);
  const float placeholder = m->distance == ORC_HNSW_DOT ? 0.0f : m->offset;
  for (int i = m->dim; i < m->actual_dim; ++i) codes[i] = f32_to_u8(placeholder, m->alpha, m->offset);
  float off = 0.0f;
  if (m->distance == ORC_HNSW_DOT) {
    float s = 0.0f;
    for (int i = 0; i < m->actual_dim; ++i) s += (float)codes[i];
    off = s * m->alpha * m->offset;
  } else if (m->distance == ORC_HNSW_L2) {
    float s = 0.0f;
    for (int i = 0; i < m->actual_dim; ++i) s += (float)codes[i] * (float)codes[i];
    off = s * m->alpha * m->alpha;
  }
  return m->invert ? -off : off;
}

static int32_t raw_score(const uint8_t* q, const uint8_t* v, const orc_u8_meta* m) { /* impl_score_dot / impl_score_l1 (:392-413) */
  int32_t s = 0;
  if (m->distance == ORC_HNSW_L1) {
    for (int i = 0; i < m->actual_dim; ++i) s += abs((int)q[i] - (int)v[i]);
  } else {
    for (int i = 0; i < m->actual_dim; ++i) s += (int)q[i] * (int)v[i];
  }
  return s;
}

/* score_point_simple (:163-229): multiplier * score as f32 + query.offset + vector_offset */
float orc_u8_score_point(const uint8_t* qcodes, float qoffset, const uint8_t* encoded, const orc_u8_meta* m, uint32_t i) {
  const uint8_t* r = encoded + (size_t)i * (m->actual_dim + 4);
  float vo;
  memcpy(&vo, r, 4);
  return m->multiplier * (float)raw_score(qcodes, r + 4, m) + qoffset + vo;
}

float orc_hnsw_postprocess(int distance, float score) { /* hnsw.rs:317-343 */
  if (distance == ORC_HNSW_L1) return fabsf(score);
  if (distance == ORC_HNSW_L2) return sqrtf(fabsf(score));
  return fabsf(1.0f - score);
}

/* generate_scores (hnsw.rs:120-140): every row's post-processed quantised distance to the (pre-processed) query */
void orc_hnsw_generate_scores(const uint8_t* encoded, const orc_u8_meta* m, const float* query, float* out) {
  uint8_t* qc = (uint8_t*)malloc(m->actual_dim);
  const float qo = orc_u8_encode_query(query, m, qc);
  for (int64_t i = 0; i < m->count; ++i) out[i] = orc_hnsw_postprocess(m->distance, orc_u8_score_point(qc, qo, encoded, m, (uint32_t)i));
  free(qc);
}

/* ------------------------------------------------------------------------------------------------------------------
 * std::collections::BinaryHeap<ScoredPointOffset> (max-heap by score) — see the header
 * ---------------------------------------------------------------------------------------------------------------- */
typedef struct { uint32_t idx; float score; } sp_t;
typedef struct { sp_t* d; int len, cap; } heap_t;

/* OrderedFloat total order: NaN is the greatest and equal to itself */
static int sp_le(sp_t a, sp_t b) { /* a <= b */
  if (a.score != a.score) return b.score != b.score;
  if (b.score != b.score) return 1;
  return a.score <= b.score;
}
static int sp_lt(sp_t a, sp_t b) { return !sp_le(b, a); }
static int sp_ge(sp_t a, sp_t b) { return sp_le(b, a); }

static void heap_init(heap_t* h, int cap) { h->d = (sp_t*)malloc(sizeof(sp_t) * (size_t)(cap > 4 ? cap : 4)); h->len = 0; h->cap = cap > 4 ? cap : 4; }
static void heap_free(heap_t* h) { free(h->d); h->d = NULL; }
static void heap_reserve(heap_t* h) {
  if (h->len == h->cap) { h->cap *= 2; h->d = (sp_t*)realloc(h->d, sizeof(sp_t) * (size_t)h->cap); }
}
static int heap_sift_up(heap_t* h, int start, int pos) {
  sp_t e = h->d[pos];
  while (pos > start) {
    const int parent = (pos - 1) / 2;
    if (sp_le(e, h->d[parent])) break;
    h->d[pos] = h->d[parent];
    pos = parent;
  }
  h->d[pos] = e;
  return pos;
}
static void heap_sift_down_range(heap_t* h, int pos, int end) {
  sp_t e = h->d[pos];
  int child = 2 * pos + 1;
  const int lim = end >= 2 ? end - 2 : 0; /* end.saturating_sub(2) */
  while (child <= lim && end >= 2) {
    child += sp_le(h->d[child], h->d[child + 1]) ? 1 : 0;
    if (sp_ge(e, h->d[child])) { h->d[pos] = e; return; }
    h->d[pos] = h->d[child];
    pos = child;
    child = 2 * pos + 1;
  }
  if (child == end - 1 && sp_lt(e, h->d[child])) { h->d[pos] = h->d[child]; pos = child; }
  h->d[pos] = e;
}
static void heap_sift_down_to_bottom(heap_t* h, int pos) {
  const int end = h->len, start = pos;
  sp_t e = h->d[pos];
  int child = 2 * pos + 1;
  const int lim = end >= 2 ? end - 2 : 0;
  while (child <= lim && end >= 2) {
    child += sp_le(h->d[child], h->d[child + 1]) ? 1 : 0;
    h->d[pos] = h->d[child];
    pos = child;
    child = 2 * pos + 1;
  }
  if (child == end - 1) { h->d[pos] = h->d[child]; pos = child; }
  h->d[pos] = e;
  heap_sift_up(h, start, pos);
}
static void heap_push(heap_t* h, sp_t v) {
  heap_reserve(h);
  h->d[h->len++] = v;
  heap_sift_up(h, 0, h->len - 1);
}
static int heap_pop(heap_t* h, sp_t* out) {
  if (h->len == 0) return 0;
  sp_t item = h->d[--h->len];
  if (h->len > 0) {
    sp_t t = h->d[0];
    h->d[0] = item;
    item = t;
    heap_sift_down_to_bottom(h, 0);
  }
  *out = item;
  return 1;
}
/* into_sorted_vec: ascending */
static void heap_into_sorted(heap_t* h) {
  int end = h->len;
  while (end > 1) {
    --end;
    sp_t t = h->d[0]; h->d[0] = h->d[end]; h->d[end] = t;
    heap_sift_down_range(h, 0, end);
  }
}

/* FixedLengthPriorityQueue<T> = BinaryHeap<Reverse<T>> of bounded length: the same heap with the order reversed */
typedef struct { sp_t* d; int len, length; } flpq_t;
static int r_le(sp_t a, sp_t b) { return sp_le(b, a); } /* Reverse(a) <= Reverse(b)  <=>  b <= a */
static void flpq_sift_up(flpq_t* q, int pos) {
  sp_t e = q->d[pos];
  while (pos > 0) {
    const int parent = (pos - 1) / 2;
    if (r_le(e, q->d[parent])) break;
    q->d[pos] = q->d[parent];
    pos = parent;
  }
  q->d[pos] = e;
}
static void flpq_sift_down_range(flpq_t* q, int pos, int end) {
  sp_t e = q->d[pos];
  int child = 2 * pos + 1;
  const int lim = end >= 2 ? end - 2 : 0;
  while (child <= lim && end >= 2) {
    child += r_le(q->d[child], q->d[child + 1]) ? 1 : 0;
    if (r_le(q->d[child], e)) { q->d[pos] = e; return; } /* hole >= child */
    q->d[pos] = q->d[child];
    pos = child;
    child = 2 * pos + 1;
  }
  if (child == end - 1 && !r_le(q->d[child], e)) { q->d[pos] = q->d[child]; pos = child; } /* hole < child */
  q->d[pos] = e;
}
static void flpq_init(flpq_t* q, int length) { q->d = (sp_t*)malloc(sizeof(sp_t) * (size_t)(length + 1)); q->len = 0; q->length = length; }
static void flpq_free(flpq_t* q) { free(q->d); q->d = NULL; }
/* push (:52-66): returns 1 and *removed when the queue was full (the value that did not stay) */
static int flpq_push(flpq_t* q, sp_t v, sp_t* removed) {
  if (q->len < q->length) {
    q->d[q->len++] = v;
    flpq_sift_up(q, q->len - 1);
    return 0;
  }
  /* x = peek_mut(); if x.0 < value.0 { swap }  — Reverse order: x.0 < value.0  <=>  Reverse(x) ... compare the inner values */
  if (sp_lt(q->d[0], v)) {
    sp_t t = q->d[0];
    q->d[0] = v;
    v = t;
    flpq_sift_down_range(q, 0, q->len); /* PeekMut drop */
  }
  *removed = v;
  return 1;
}
static int flpq_top(const flpq_t* q, sp_t* out) { if (!q->len) return 0; *out = q->d[0]; return 1; }
/* into_sorted_vec of the Reverse heap, peeled: descending by score */
static void flpq_into_sorted(flpq_t* q) {
  int end = q->len;
  while (end > 1) {
    --end;
    sp_t t = q->d[0]; q->d[0] = q->d[end]; q->d[end] = t;
    flpq_sift_down_range(q, 0, end);
  }
}

/* ------------------------------------------------------------------------------------------------------------------
 * graph
 * ---------------------------------------------------------------------------------------------------------------- */
typedef struct {
  int64_t n;
  int m, m0, ef_construct;
  int* level;          /* [n] highest level of the point                                   */
  int64_t* first;      /* [n] index of the point's level-0 list in the arrays below        */
  uint32_t* links;     /* per list: capacity m0 (level 0) or m                             */
  int* nlinks;         /* per list                                                         */
  int64_t* list_off;   /* per list: offset into links                                      */
  int64_t nlists;
  uint8_t* ready;      /* [n]                                                              */
  /* EntryPoints (entry_points.rs:43-46) with every point passing the filter: entry_points has <= 1 element */
  int has_entry;
  uint32_t entry_point;
  int entry_level;
} orc_graph;

static int gm(const orc_graph* g, int level) { return level == 0 ? g->m0 : g->m; }
static uint32_t* g_links(const orc_graph* g, uint32_t p, int level) { return g->links + g->list_off[g->first[p] + level]; }
static int* g_n(const orc_graph* g, uint32_t p, int level) { return g->nlinks + g->first[p] + level; }

orc_graph* orc_hnsw_graph_new(int64_t n, int m, int ef_construct, const int* levels) {
  orc_graph* g = (orc_graph*)calloc(1, sizeof(orc_graph));
  g->n = n; g->m = m; g->m0 = 2 * m; g->ef_construct = ef_construct; /* hnsw.rs:149 */
  g->level = (int*)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
  g->first = (int64_t*)malloc(sizeof(int64_t) * (size_t)(n > 0 ? n : 1));
  int64_t lists = 0;
  for (int64_t i = 0; i < n; ++i) { g->level[i] = levels[i]; g->first[i] = lists; lists += levels[i] + 1; }
  g->nlists = lists;
  g->nlinks = (int*)calloc((size_t)(lists > 0 ? lists : 1), sizeof(int));
  g->list_off = (int64_t*)malloc(sizeof(int64_t) * (size_t)(lists > 0 ? lists : 1));
  int64_t off = 0;
  for (int64_t i = 0; i < n; ++i)
    for (int l = 0; l <= levels[i]; ++l) { g->list_off[g->first[i] + l] = off; off += (l == 0 ? g->m0 : g->m); }
  g->links = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)(off > 0 ? off : 1));
  g->ready = (uint8_t*)calloc((size_t)(n > 0 ? n : 1), 1);
  return g;
}
void orc_hnsw_graph_free(orc_graph* g) {
  if (!g) return;
  free(g->level); free(g->first); free(g->links); free(g->nlinks); free(g->list_off); free(g->ready); free(g);
}
int orc_hnsw_graph_links(const orc_graph* g, uint32_t p, int level, uint32_t* out) {
  if (level > g->level[p]) return -1;
  const int c = *g_n(g, p, level);
  memcpy(out, g_links(g, p, level), sizeof(uint32_t) * (size_t)c);
  return c;
}
void orc_hnsw_graph_set_links(orc_graph* g, uint32_t p, int level, const uint32_t* in, int c) {
  memcpy(g_links(g, p, level), in, sizeof(uint32_t) * (size_t)c);
  *g_n(g, p, level) = c;
  g->ready[p] = 1;
}
void orc_hnsw_graph_set_entry(orc_graph* g, uint32_t p, int level) { g->has_entry = 1; g->entry_point = p; g->entry_level = level; }
int orc_hnsw_graph_entry(const orc_graph* g, uint32_t* p, int* level) { *p = g->entry_point; *level = g->entry_level; return g->has_entry; }

/* scorer: Original (build) or Quantized (search) — point_scorer.rs:47-70 */
typedef struct {
  int quantized;
  /* Original */
  const float* column; int dim, distance; uint32_t self;
  /* Quantized */
  const uint8_t* encoded; const orc_u8_meta* meta; const uint8_t* qcodes; float qoffset;
} scorer_t;

static float original_score(const scorer_t* s, uint32_t a, uint32_t b) { /* calculate_score (:133-174) */
  const float* x = s->column + (size_t)a * s->dim;
  const float* y = s->column + (size_t)b * s->dim;
  float acc = 0.0f;
  if (s->distance == ORC_HNSW_DOT) { for (int i = 0; i < s->dim; ++i) acc += x[i] * y[i]; return acc; }
  if (s->distance == ORC_HNSW_L1) { for (int i = 0; i < s->dim; ++i) acc += fabsf(x[i] - y[i]); return -acc; }
  for (int i = 0; i < s->dim; ++i) { const float d = x[i] - y[i]; acc += d * d; } /* powi(2) */
  return -acc;
}
static float score_point(const scorer_t* s, uint32_t p) {
  if (s->quantized) return orc_u8_score_point(s->qcodes, s->qoffset, s->encoded, s->meta, p);
  return original_score(s, s->self, p);
}
static float score_internal(const scorer_t* s, uint32_t a, uint32_t b) { return original_score(s, a, b); } /* build only */

/* visited list: a byte per point (visited_pool.rs) */
typedef struct { uint8_t* v; uint32_t* touched; int64_t nt; } visited_t;
static void visited_init(visited_t* vl, int64_t n) { vl->v = (uint8_t*)calloc((size_t)(n > 0 ? n : 1), 1); vl->touched = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)(n > 0 ? n : 1)); vl->nt = 0; }
static void visited_reset(visited_t* vl) { for (int64_t i = 0; i < vl->nt; ++i) vl->v[vl->touched[i]] = 0; vl->nt = 0; }
static void visited_free(visited_t* vl) { free(vl->v); free(vl->touched); }
static int visited_check(const visited_t* vl, uint32_t p) { return vl->v[p]; }
static int visited_check_and_update(visited_t* vl, uint32_t p) { const int was = vl->v[p]; if (!was) { vl->v[p] = 1; vl->touched[vl->nt++] = p; } return was; }

/* SearchContext (search_context.rs) */
typedef struct { flpq_t nearest; heap_t candidates; } sctx_t;
static void sctx_init(sctx_t* c, sp_t entry, int ef) {
  flpq_init(&c->nearest, ef);
  sp_t dummy;
  flpq_push(&c->nearest, entry, &dummy);
  heap_init(&c->candidates, 64);
  heap_push(&c->candidates, entry);
}
static float sctx_lower_bound(const sctx_t* c) { sp_t t; return flpq_top(&c->nearest, &t) ? t.score : -3.4028235e38f; } /* min_value() */
static void sctx_process(sctx_t* c, sp_t sp) {
  sp_t removed;
  const int full = flpq_push(&c->nearest, sp, &removed);
  const int was_added = !full || removed.idx != sp.idx;
  if (was_added) heap_push(&c->candidates, sp);
}

/* links_map of the builder filters by ready_list (graph_layers_builder.rs:77-87); of the finished graph it does not */
static void search_on_level_inner(const orc_graph* g, sctx_t* c, int level, visited_t* vl, const scorer_t* s, int only_ready) {
  const int limit = gm(g, level);
  uint32_t ids[256];
  sp_t cand;
  while (heap_pop(&c->candidates, &cand)) {
    if (cand.score < sctx_lower_bound(c)) break;
    int np = 0;
    const uint32_t* l = g_links(g, cand.idx, level);
    const int cnt = *g_n(g, cand.idx, level);
    for (int i = 0; i < cnt; ++i)
      if ((!only_ready || g->ready[l[i]]) && !visited_check(vl, l[i])) ids[np++] = l[i];
    /* score_points(points_ids, limit) (point_scorer.rs:96-121): at most `limit` of them, in order */
    const int take = np < limit ? np : limit;
    for (int i = 0; i < take; ++i) {
      sp_t sp = {ids[i], score_point(s, ids[i])};
      sctx_process(c, sp);
      visited_check_and_update(vl, sp.idx);
    }
  }
}

static sp_t search_entry(const orc_graph* g, uint32_t entry, int top_level, int target_level, const scorer_t* s, int only_ready) {
  sp_t cur = {entry, score_point(s, entry)};
  for (int level = top_level; level > target_level; --level) { /* rev_range(top, target) */
    const int limit = gm(g, level);
    int changed = 1;
    while (changed) {
      changed = 0;
      const uint32_t* l = g_links(g, cur.idx, level);
      const int cnt = *g_n(g, cur.idx, level);
      uint32_t ids[256];
      int np = 0;
      for (int i = 0; i < cnt; ++i)
        if (!only_ready || g->ready[l[i]]) ids[np++] = l[i];
      const int take = np < limit ? np : limit;
      for (int i = 0; i < take; ++i) {
        const float sc = score_point(s, ids[i]);
        if (sc > cur.score) { changed = 1; cur.idx = ids[i]; cur.score = sc; }
      }
    }
  }
  return cur;
}

/* GraphLayers::search (graph_layers.rs:218-247) + HNSWIndex::search (hnsw.rs:100-118): ef = 4 * limit, result post-processed.
 * `query` is already pre-processed (preprocess_query, hnsw.rs:307-312). Returns the number of results. */
int orc_hnsw_search(const orc_graph* g, const uint8_t* encoded, const orc_u8_meta* meta, const float* query, int limit,
                    uint32_t* out_ids, float* out_dist) {
  if (!g->has_entry || limit <= 0) return 0;
  uint8_t* qc = (uint8_t*)malloc(meta->actual_dim);
  scorer_t s;
  memset(&s, 0, sizeof(s));
  s.quantized = 1; s.encoded = encoded; s.meta = meta; s.qcodes = qc;
  s.qoffset = orc_u8_encode_query(query, meta, qc);
  const int ef = limit * 4 > limit ? limit * 4 : limit; /* max(top, ef) */
  sp_t zero = search_entry(g, g->entry_point, g->entry_level, 0, &s, 0);
  visited_t vl;
  visited_init(&vl, g->n);
  visited_check_and_update(&vl, zero.idx);
  sctx_t c;
  sctx_init(&c, zero, ef);
  search_on_level_inner(g, &c, 0, &vl, &s, 0);
  flpq_into_sorted(&c.nearest);   /* descending: the heap array ends up ascending in Reverse order = descending score */
  int k = 0;
  for (int i = 0; i < c.nearest.len && k < limit; ++i, ++k) {
    out_ids[k] = c.nearest.d[i].idx;
    out_dist[k] = orc_hnsw_postprocess(meta->distance, c.nearest.d[i].score);
  }
  flpq_free(&c.nearest); heap_free(&c.candidates); visited_free(&vl); free(qc);
  return k;
}

/* select_candidate_with_heuristic_from_sorted (graph_layers_builder.rs:300-327): candidates in DESCENDING score order */
static int select_heuristic(const sp_t* cands, int nc, int m, const scorer_t* s, uint32_t* out) {
  int k = 0;
  for (int i = 0; i < nc; ++i) {
    if (k >= m) break;
    int good = 1;
    for (int j = 0; j < k; ++j)
      if (score_internal(s, cands[i].idx, out[j]) > cands[i].score) { good = 0; break; }
    if (good) out[k++] = cands[i].idx;
  }
  return k;
}

/* link_new_point (:343-389), sequential (the reference inserts the first 256 points this way and the rest in parallel) */
void orc_hnsw_link_new_point(orc_graph* g, const float* column, int dim, int distance, uint32_t p) {
  scorer_t s;
  memset(&s, 0, sizeof(s));
  s.column = column; s.dim = dim; s.distance = distance; s.self = p;
  const int level = g->level[p];
  if (g->has_entry) {
    sp_t level_entry;
    if (g->entry_level > level) level_entry = search_entry(g, g->entry_point, g->entry_level, level, &s, 1);
    else { level_entry.idx = g->entry_point; level_entry.score = score_internal(&s, p, g->entry_point); }
    const int linking_level = level < g->entry_level ? level : g->entry_level;
    visited_t vl;
    visited_init(&vl, g->n);
    for (int cl = linking_level; cl >= 0; --cl) {
      /* link_new_point_on_level (:418-462) */
      visited_reset(&vl);
      visited_check_and_update(&vl, level_entry.idx);
      sctx_t c;
      sctx_init(&c, level_entry, g->ef_construct);
      search_on_level_inner(g, &c, cl, &vl, &s, 1);
      /* nearest.iter_unsorted().max(): Iterator::max returns the LAST of equal maxima */
      if (c.nearest.len > 0) {
        sp_t best = c.nearest.d[0];
        for (int i = 1; i < c.nearest.len; ++i)
          if (sp_le(best, c.nearest.d[i])) best = c.nearest.d[i];
        level_entry = best;
      }
      /* link_with_heuristic (:464-520) */
      const int level_m = gm(g, cl);
      uint32_t* mine = g_links(g, p, cl);
      int* nmine = g_n(g, p, cl);
      for (int i = 0; i < *nmine; ++i)
        if (!visited_check(&vl, mine[i]) && g->ready[mine[i]]) { sp_t sp = {mine[i], score_point(&s, mine[i])}; sctx_process(&c, sp); }
      flpq_into_sorted(&c.nearest);
      uint32_t selected[256];
      const int ns = select_heuristic(c.nearest.d, c.nearest.len, level_m, &s, selected);
      memcpy(mine, selected, sizeof(uint32_t) * (size_t)ns);
      *nmine = ns;
      for (int k = 0; k < ns; ++k) {
        const uint32_t other = selected[k];
        uint32_t* ol = g_links(g, other, cl);
        int* on = g_n(g, other, cl);
        if (*on < level_m) { ol[(*on)++] = p; continue; }
        heap_t h;
        heap_init(&h, level_m + 1);
        sp_t e = {p, score_internal(&s, p, other)};
        heap_push(&h, e);
        for (int i = 0; i < *on && i < level_m; ++i) { sp_t x = {ol[i], score_internal(&s, ol[i], other)}; heap_push(&h, x); }
        heap_into_sorted(&h); /* ascending; the reference walks it .rev() */
        sp_t desc[260];
        for (int i = 0; i < h.len; ++i) desc[i] = h.d[h.len - 1 - i];
        uint32_t sel2[256];
        const int n2 = select_heuristic(desc, h.len, level_m, &s, sel2);
        memcpy(ol, sel2, sizeof(uint32_t) * (size_t)n2);
        *on = n2;
        heap_free(&h);
      }
      flpq_free(&c.nearest); heap_free(&c.candidates);
    }
    visited_free(&vl);
  }
  g->ready[p] = 1;
  /* entry_points.new_point (entry_points.rs:56-103), every point passing the filter */
  if (!g->has_entry) { g->has_entry = 1; g->entry_point = p; g->entry_level = level; }
  else if (g->entry_level < level) { g->entry_point = p; g->entry_level = level; }
}

/* HNSWIndex::build's insertion loop run sequentially (hnsw.rs:158-235); levels are the caller's (the reference draws them
 * from thread_rng(): get_random_layer = round(-ln(u) * 1 / ln(max(m, 2))), graph_layers_builder.rs:246-255) */
void orc_hnsw_build(orc_graph* g, const float* column, int dim, int distance) {
  for (int64_t i = 0; i < g->n; ++i) orc_hnsw_link_new_point(g, column, dim, distance, (uint32_t)i);
}

/* bulk import of a graph whose lists come in point-major, level-minor order (dbhip_hnsw_export_graph's layout) */
void orc_hnsw_graph_import(orc_graph* g, const uint32_t* links_flat, const int* nlinks, uint32_t entry_point, int entry_level) {
  int64_t off = 0;
  for (int64_t l = 0; l < g->nlists; ++l) {
    memcpy(g->links + g->list_off[l], links_flat + off, sizeof(uint32_t) * (size_t)nlinks[l]);
    g->nlinks[l] = nlinks[l];
    off += nlinks[l];
  }
  memset(g->ready, 1, (size_t)g->n);
  g->has_entry = 1; g->entry_point = entry_point; g->entry_level = entry_level;
}

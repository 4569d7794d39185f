/*
 * q1_typed.c — the CPU baseline of bench.py: TPC-H Q1 in the reference's pipeline shape, type-specialised.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle.h). Never linked into libdbhip.so.
 *
 * oracle.c's orc_q1_run walks the same pipeline through the GENERIC restatements (one `switch` on the column type per row
 * and per call node, the way an interpreter would) — right for a checker, but ~50x slower than what the reference's
 * monomorphised Rust does per row, so its time says little as a baseline. This file is the same pipeline with the
 * types fixed at compile time, column at a time over 65,536-row blocks, which is what rustc generates for
 *   TransformFilter            l_shipdate <= cutoff -> Bitmap -> selection      (filter_executor.rs:81-118)
 *   take                       gather of the 6 surviving columns               (kernels/take.rs:43)
 *   CompoundBlockOperator      1 - l_discount ; price * (..) ; 1 + l_tax ; (..) * (..)
 *                              one materialised column per call node           (decimal/arithmetic.rs:190-316)
 *   TransformPartialAggregate  group hash of the two String keys (group_hash.rs:522-553, :267-281), probe of a
 *                              linear-probing index, payload rows, per-row state updates:
 *                              sum Decimal64 -> i64 state, sum Decimal128 -> i128 state with the overflow check of
 *                              aggregate_sum.rs:203-216 (precision > 18), count                (aggregate_hashtable.rs:168-292)
 *   TransformFinalAggregate    merge of the per-thread partial tables          (transform_aggregate_final.rs:160-175)
 * Results are identical to orc_q1_run's (tests/test_oracle_cpu.py checks that on every size it runs), so either can
 * stand next to the GPU number; bench.py times this one.
 */
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "oracle.h"

typedef __int128 i128;

#define Q1T_SLOTS 64 /* power of two; the partial table of a thread (Q1 has 4 groups, 64 is the struct's limit) */

typedef struct {
  int used;
  uint64_t hash;
  uint8_t rf[16], ls[16]; /* the key views */
  int64_t sum_qty, sum_price, sum_disc;
  i128 sum_dp, sum_ch;
  uint64_t count;
} q1t_group;

typedef struct {
  q1t_group g[Q1T_SLOTS];
  int n, overflow, too_many;
} q1t_table;

static const i128 Q1T_DEC_MAX = (((i128)0x4B3B4CA85A86C47AULL) << 64) | (i128)0x098A223FFFFFFFFFULL; /* 10^38 - 1 */

static inline uint64_t q1t_hash_view(const uint8_t* v) {
  uint32_t len;
  memcpy(&len, v, 4);
  return orc_agg_hash_bytes(v + 4, len); /* Q1's keys are 1-byte strings: inline views */
}

static inline q1t_group* q1t_find(q1t_table* t, uint64_t h, const uint8_t* rf, const uint8_t* ls) {
  uint32_t s = (uint32_t)h & (Q1T_SLOTS - 1);
  for (int step = 0; step < Q1T_SLOTS; ++step, s = (s + 1) & (Q1T_SLOTS - 1)) {
    q1t_group* g = &t->g[s];
    if (!g->used) {
      if (t->n >= Q1T_SLOTS - 1) { t->too_many = 1; return NULL; }
      memset(g, 0, sizeof(*g));
      g->used = 1; g->hash = h;
      memcpy(g->rf, rf, 16); memcpy(g->ls, ls, 16);
      t->n++;
      return g;
    }
    if (g->hash == h && memcmp(g->rf, rf, 16) == 0 && memcmp(g->ls, ls, 16) == 0) return g;
  }
  t->too_many = 1;
  return NULL;
}

typedef struct {
  const int64_t *qty, *price, *disc, *tax;
  const uint8_t *rf, *ls;
  const int32_t* sd;
  int32_t cutoff;
  int64_t n, block_rows;
  int tid, nthreads;
  q1t_table table;
} q1t_worker;

static void* q1t_work(void* p) {
  q1t_worker* w = (q1t_worker*)p;
  const int64_t B = w->block_rows;
  uint64_t* bm = (uint64_t*)malloc(((size_t)B / 64 + 2) * 8);
  uint32_t* sel = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)B);
  int64_t *tq = malloc(8 * (size_t)B), *tp = malloc(8 * (size_t)B), *td = malloc(8 * (size_t)B), *tt = malloc(8 * (size_t)B);
  uint8_t *trf = malloc(16 * (size_t)B), *tls = malloc(16 * (size_t)B);
  int64_t *omd = malloc(8 * (size_t)B), *opt = malloc(8 * (size_t)B);
  i128 *dp = malloc(16 * (size_t)B), *ch = malloc(16 * (size_t)B);
  uint64_t* hs = malloc(8 * (size_t)B);
  memset(&w->table, 0, sizeof(w->table));
  const int64_t nblocks = (w->n + B - 1) / B;
  for (int64_t b = w->tid; b < nblocks; b += w->nthreads) {
    const int64_t s = b * B, m = w->n - s < B ? w->n - s : B;
    const int32_t* sd = w->sd + s;
    /* TransformFilter: comparison -> Bitmap (64 results per word, register_comparison.rs:52-96) */
    for (int64_t wd = 0; wd * 64 < m; ++wd) {
      uint64_t bits = 0;
      const int64_t lim = m - wd * 64 < 64 ? m - wd * 64 : 64;
      for (int64_t j = 0; j < lim; ++j) bits |= (uint64_t)(sd[wd * 64 + j] <= w->cutoff) << j;
      bm[wd] = bits;
    }
    /* Bitmap -> ascending selection */
    int64_t k = 0;
    for (int64_t wd = 0; wd * 64 < m; ++wd) {
      uint64_t bits = bm[wd];
      while (bits) { sel[k++] = (uint32_t)(wd * 64 + __builtin_ctzll(bits)); bits &= bits - 1; }
    }
    /* take: every projected column is gathered */
    const int64_t *q = w->qty + s, *pr = w->price + s, *di = w->disc + s, *ta = w->tax + s;
    const uint8_t *rf = w->rf + 16 * s, *ls = w->ls + 16 * s;
    for (int64_t j = 0; j < k; ++j) tq[j] = q[sel[j]];
    for (int64_t j = 0; j < k; ++j) tp[j] = pr[sel[j]];
    for (int64_t j = 0; j < k; ++j) td[j] = di[sel[j]];
    for (int64_t j = 0; j < k; ++j) tt[j] = ta[sel[j]];
    for (int64_t j = 0; j < k; ++j) memcpy(trf + 16 * j, rf + 16 * (size_t)sel[j], 16);
    for (int64_t j = 0; j < k; ++j) memcpy(tls + 16 * j, ls + 16 * (size_t)sel[j], 16);
    /* maps, one column per call node. 1 (UInt8 -> Decimal(3,0)) is rescaled to scale 2 = 100; the products keep the sum of the
       scales (4, then 6 <= 12: no rounding division, arithmetic.rs:98-110) */
    for (int64_t j = 0; j < k; ++j) omd[j] = 100 - td[j];
    for (int64_t j = 0; j < k; ++j) dp[j] = (i128)tp[j] * (i128)omd[j];
    for (int64_t j = 0; j < k; ++j) opt[j] = 100 + tt[j];
    for (int64_t j = 0; j < k; ++j) ch[j] = dp[j] * (i128)opt[j];
    /* partial aggregation: hashes of the batch first (group_hash_entries), then probe + state updates row by row */
    for (int64_t j = 0; j < k; ++j) hs[j] = q1t_hash_view(trf + 16 * j) * 0xd1cefa08eb382d69ULL ^ q1t_hash_view(tls + 16 * j);
    for (int64_t j = 0; j < k; ++j) {
      q1t_group* g = q1t_find(&w->table, hs[j], trf + 16 * j, tls + 16 * j);
      if (!g) break;
      g->sum_qty += tq[j];
      g->sum_price += tp[j];
      g->sum_dp = (i128)((unsigned __int128)g->sum_dp + (unsigned __int128)dp[j]);
      if (g->sum_dp > Q1T_DEC_MAX || g->sum_dp < -Q1T_DEC_MAX) w->table.overflow = 1; /* precision 31 > 18: checked */
      g->sum_ch = (i128)((unsigned __int128)g->sum_ch + (unsigned __int128)ch[j]);
      if (g->sum_ch > Q1T_DEC_MAX || g->sum_ch < -Q1T_DEC_MAX) w->table.overflow = 1;
      g->sum_disc += td[j];
      g->count += 1;
    }
  }
  free(bm); free(sel); free(tq); free(tp); free(td); free(tt); free(trf); free(tls); free(omd); free(opt); free(dp); free(ch); free(hs);
  return NULL;
}

/* Same contract as orc_q1_run (oracle.c): returns the number of groups, -1 if there are more than 63, -105 on a decimal
 * sum overflow. */
int orc_q1_run_typed(const int64_t* qty, const int64_t* price, const int64_t* disc, const int64_t* tax, const void* rf_views,
                     const void* ls_views, const int32_t* shipdate, int32_t cutoff, int64_t n, int threads, int64_t block_rows,
                     orc_q1_result* out) {
  if (threads < 1) threads = 1;
  q1t_worker* ws = (q1t_worker*)calloc((size_t)threads, sizeof(q1t_worker));
  pthread_t* th = (pthread_t*)calloc((size_t)threads, sizeof(pthread_t));
  for (int t = 0; t < threads; ++t) {
    ws[t].qty = qty; ws[t].price = price; ws[t].disc = disc; ws[t].tax = tax;
    ws[t].rf = (const uint8_t*)rf_views; ws[t].ls = (const uint8_t*)ls_views; ws[t].sd = shipdate;
    ws[t].cutoff = cutoff; ws[t].n = n; ws[t].block_rows = block_rows; ws[t].tid = t; ws[t].nthreads = threads;
    if (threads == 1) q1t_work(&ws[t]); else pthread_create(&th[t], NULL, q1t_work, &ws[t]);
  }
  if (threads > 1) for (int t = 0; t < threads; ++t) pthread_join(th[t], NULL);
  /* final aggregation: merge every partial table into the first */
  q1t_table* fin = &ws[0].table;
  int overflow = fin->overflow, too_many = fin->too_many;
  for (int t = 1; t < threads; ++t) {
    q1t_table* pt = &ws[t].table;
    overflow |= pt->overflow; too_many |= pt->too_many;
    for (int s = 0; s < Q1T_SLOTS; ++s) {
      const q1t_group* g = &pt->g[s];
      if (!g->used) continue;
      q1t_group* d = q1t_find(fin, g->hash, g->rf, g->ls);
      if (!d) { too_many = 1; break; }
      d->sum_qty += g->sum_qty; d->sum_price += g->sum_price; d->sum_disc += g->sum_disc; d->count += g->count;
      d->sum_dp = (i128)((unsigned __int128)d->sum_dp + (unsigned __int128)g->sum_dp);
      d->sum_ch = (i128)((unsigned __int128)d->sum_ch + (unsigned __int128)g->sum_ch);
      if (d->sum_dp > Q1T_DEC_MAX || d->sum_dp < -Q1T_DEC_MAX || d->sum_ch > Q1T_DEC_MAX || d->sum_ch < -Q1T_DEC_MAX) overflow = 1;
    }
  }
  too_many |= fin->too_many;
  int rc;
  if (too_many) rc = -1;
  else if (overflow) rc = -105;
  else {
    memset(out, 0, sizeof(*out));
    int i = 0;
    for (int s = 0; s < Q1T_SLOTS; ++s) {
      const q1t_group* g = &fin->g[s];
      if (!g->used) continue;
      memcpy(out->returnflag[i], g->rf, 16); memcpy(out->linestatus[i], g->ls, 16);
      out->sum_qty[i] = g->sum_qty; out->sum_price[i] = g->sum_price; out->sum_disc[i] = g->sum_disc;
      out->sum_disc_price[i] = g->sum_dp; out->sum_charge[i] = g->sum_ch; out->count[i] = g->count;
      ++i;
    }
    rc = i;
  }
  free(ws); free(th);
  return rc;
}

// gbk_partitioned.h — a fragment of k_groupby.hip (ONE translation unit: the kernels share the anonymous namespace's helpers and the table struct;
// split by kernel family in round 6, VERDICT r05 hygiene #18): the compact-row dispatch, the radix-partitioned pre-aggregation (histogram / scan / scatter / aggregate) and the choice of path per chunk.
// Included by k_groupby.hip only, in this order: gbk_rows.h, gbk_merge_lds.h, gbk_partitioned.h, gbk_api.h.

namespace {
__device__ __forceinline__ uint32_t part_of(uint64_t h, int pbits) { return (uint32_t)(h >> (64 - pbits)); }
#include "gb_compact.h"

bool gbc_enabled(const dbhip_groupby* g) {
  static const bool off = exp_env("DBHIP_GBC") && atoi(exp_env("DBHIP_GBC")) == 0;
  return !off && !g->gbc_off && g->hash_mask == ~0ULL && !g->has_long;
}

// the kernels are instantiated for 1-4 key words and 1 / 2 / 4 / 8 value words (a layout without value words runs as NV = 1)
#define GBC_NV_DISPATCH(KW_, nv_, CALL)                                                                          \
  do { if (nv_ == 1) { CALL(KW_, 1); } else if (nv_ == 2) { CALL(KW_, 2); } else if (nv_ == 4) { CALL(KW_, 4); } else { CALL(KW_, 8); } } while (0)
#define GBC_DISPATCH(D, CALL)                                                     \
  do {                                                                            \
    const int nvc_ = gbc_nv_class(D);                                             \
    switch ((D).kw) {                                                             \
      case 1: GBC_NV_DISPATCH(1, nvc_, CALL); break;                              \
      case 2: GBC_NV_DISPATCH(2, nvc_, CALL); break;                              \
      case 3: GBC_NV_DISPATCH(3, nvc_, CALL); break;                              \
      default: GBC_NV_DISPATCH(4, nvc_, CALL); break;                             \
    }                                                                             \
  } while (0)
#define GBC_FOR_ALL(M) M(1, 1) M(1, 2) M(1, 4) M(1, 8) M(2, 1) M(2, 2) M(2, 4) M(2, 8) M(3, 1) M(3, 2) M(3, 4) M(3, 8) M(4, 1) M(4, 2) M(4, 4) M(4, 8)

constexpr int PT_MAX_BITS = 14;
constexpr int PT_PMAX = 1 << PT_MAX_BITS;   // part_meta: tot[PT_PMAX] | base[PT_PMAX + 8] | pcount[PT_PMAX] | mat[nwg][P]
void decide_partitioning(dbhip_groupby* g, int64_t groups, int64_t rows_seen, int64_t n_block);
int64_t estimate_groups(int64_t d, int64_t s);
void part_geometry(const GbLayout& L, int* lcap, int* sw, size_t* lds_bytes);
void table_geometry(const dbhip_groupby* g, int* lcap, int* sw, size_t* lds_bytes);   // part_geometry, or the compact kernels' tables
int32_t partition_scatter(dbhip_groupby* g, const GbCols& C, int64_t row0, int64_t cn, int pbits, hipStream_t s);
constexpr int64_t PT_CHUNK = 64 << 20;

// After a chunk of the adaptive mode: D = groups the whole input is likely to hold (from the groups met in the rows seen so
// far). The next chunk takes as many rows as keep a partition's groups inside its LDS table — c rows drawn from D equally
// likely groups meet D (1 - exp(-c / D)) of them, wanted <= G = P x 0.6 x lcap — because every partial row costs a random
// access into the table (~6 G rows/s, r02o/r02p) and a group should cost one of those per chunk, not one per row. Fewer
// than 1.5 rows per group: nothing to pre-aggregate, the rows are inserted directly (gb_part_insert_kernel).
void adapt_chunk(dbhip_groupby* g, int64_t n_block) {
  int lcap, sw;
  size_t lds_bytes;
  table_geometry(g, &lcap, &sw, &lds_bytes);
  const double G = (double)((int64_t)1 << g->part_bits) * 0.6 * lcap;
  const int64_t est = estimate_groups(g->count_host, g->rows_seen);
  const double D = (double)est;
  const int64_t total = n_block > g->rows_seen ? n_block : g->rows_seen;
  double c = (double)PT_CHUNK;
  if (D > 64.0 * G) c = G;
  else if (D > G) c = -D * log(1.0 - G / D);
  if (c < (double)(1 << 20)) c = (double)(1 << 20);
  if (c > (double)PT_CHUNK) c = (double)PT_CHUNK;
  g->part_chunk = (int64_t)c;
  g->part_direct = D * 1.5 > (double)total ? 1 : 0;
  if (g->part_direct) g->part_chunk = PT_CHUNK;
  if (getenv("DBHIP_TRACE")) fprintf(stderr, "[dbhip] groupby adaptive: %lld groups in %lld rows -> ~%lld groups, next chunk %lld rows%s\n",
                                     (long long)g->count_host, (long long)g->rows_seen, (long long)est, (long long)g->part_chunk,
                                     g->part_direct ? " (direct insert)" : "");
}

// one partitioned chunk starting at *done; widens the partitioning (or gives it up) when too many rows spilled
int32_t partitioned_step(dbhip_groupby* g, const GbCols& C, int64_t n, hipStream_t s, int64_t* done) {
  const int64_t chunk = g->part_chunk > 0 ? g->part_chunk : PT_CHUNK;
  const int64_t cn = n - *done < chunk ? n - *done : chunk;
  int64_t spilled = 0;
  int32_t rc = add_chunk_partitioned(g, C, *done, cn, s, &spilled);
  if (rc) return rc;
  if (spilled < 0) { g->part_bits = -1; return DBHIP_OK; }   // long string keys: the caller's row path takes the rows from *done
  if (getenv("DBHIP_TRACE")) fprintf(stderr, "[dbhip] groupby partitioned chunk: rows=%lld pbits=%d spilled=%lld groups=%lld\n",
                                     (long long)cn, g->part_bits, (long long)spilled, (long long)g->count_host);
  *done += cn;
  g->rows_seen += cn;
  if (g->part_validate) {   // the check chunk of an extrapolated estimate: choose again with what it found
    g->part_validate = 0; g->part_validated = 1;
    decide_partitioning(g, g->count_host, g->rows_seen, n);
    if (g->part_bits < 0) g->fast_disabled = 1;
    return DBHIP_OK;
  }
  if (g->part_adapt) { adapt_chunk(g, n); return DBHIP_OK; }
  if (spilled * 20 > cn) {
    if (g->part_bits + 2 <= PT_MAX_BITS) g->part_bits += 2;
    else if (g->part_bits < PT_MAX_BITS) g->part_bits = PT_MAX_BITS;
    else if (chunk > (2 << 20)) g->part_chunk = chunk / 2;   // finest partitioning already: fewer groups per chunk
    else { g->part_bits = -1; g->fast_disabled = 1; }
  }
  return DBHIP_OK;
}

bool layout_has_wide_minmax(const GbLayout& L) {   // "row path only": locked min / max states, Decimal256 sums and keys
  for (int a = 0; a < L.naggs; ++a) if (gb_minmax_wide(L, a) || gb_sum256(L, a)) return true;
  for (int k = 0; k < L.nkeys; ++k) if (L.key_type[k] == DBHIP_T_DEC256) return true;
  return false;
}
bool fast_layout_ok(const GbLayout& L) {
  // (a Decimal128 min / max state is merged under a per-state lock: row path only)
  return L.nkey_words <= FK_MAXKW && L.nkeys <= FK_MAXKW && L.naggs <= FK_MAXA && L.W <= 24 && !layout_has_wide_minmax(L);
}

// add_block through the LDS pre-aggregation kernel, chunk by chunk. Returns -1 when the caller must
// use the generic row path for rows [*done, n).
int32_t add_block_fast(dbhip_groupby* g, const GbCols& C, int64_t n, hipStream_t s, int64_t* done) {
  const GbLayout& L = g->L;
  const int sw = L.W | 1;  // odd stride (in 8-byte words): conflict-free LDS rows
  int lcap = 64;
  while ((size_t)(lcap * 2) * (sw + 1) * 8 <= 48 * 1024) lcap *= 2;
  const size_t lds_bytes = (size_t)lcap * (sw + 1) * 8;
  // layout class: small = <= 2 key words, <= 2 one-word aggregates (8 rows per lane); else general (2 rows)
  bool hi = false;
  for (int a = 0; a < L.naggs; ++a) hi |= L.agg_type[a] == DBHIP_T_DEC128 && L.agg_kind[a] != DBHIP_AGG_COUNT;
  const bool small_layout = L.nkey_words <= 2 && L.nkeys <= 2 && L.naggs <= 2 && !hi;
  // compact-row kernels (gb_compact.h): this call's layout AND columns qualify
  GbcDesc GD;
  const bool gbc = !layout_has_wide_minmax(L) && gbc_enabled(g) && gbc_describe(L, C, &GD);
  GD.ctrl = g->ctrl;
  g->gbc_active = gbc ? 1 : 0;
  if (!gbc && !fast_layout_ok(L)) return -1;   // (the generic LDS kernel is instantiated up to FK_MAXKW key words / FK_MAXA aggregates)
  const int64_t CHUNK = 16 << 20;
  int blocks_per_cu = (int)((160 * 1024) / (lds_bytes + 1024));
  if (blocks_per_cu > 4) blocks_per_cu = 4;
  if (blocks_per_cu < 1) blocks_per_cu = 1;
  const int max_grid = 256 * blocks_per_cu;  // every workgroup resident at once: no tail round
  int32_t rc;
  while (*done < n) {
    DBHIP_POLL_CANCEL(s, "dbhip_groupby_add_block");
    if (g->part_bits > 0) {
      if (n - *done < g->part_min_rows) return -1;  // small remainder: row path
      if ((rc = partitioned_step(g, C, n, s, done))) return rc;
      continue;
    }
    if (g->fast_disabled) return -1;
    // A handful of groups (the probing chunk showed <= 8, nothing spilled): the rest of the block goes through the fused
    // few-groups kernel — key table in scalar registers, states in per-lane registers, no LDS atomics (k_fagg.hip; LDS
    // atomics of 64 lanes on 4 addresses serialise: 0.12 of the HBM rate on this path at 4 groups). A workgroup that
    // meets a 9th group makes it give up with nothing merged; the LDS path then takes the rows.
    // Only the RUN-TIME SPECIALISED form of that kernel is used here (r02g: interpreted it loses to the LDS path, 1.34 vs 1.04 ms
    // per 60 M rows; specialised, with every load of a chunk issued up front, r03: see DESIGN §2.3). Plain add_block has no
    // PREPARE, so the kernel is looked up in the in-process / on-disk caches; when it is nowhere yet a detached helper compiles it
    // into the on-disk cache and THIS block takes the LDS path — a query never waits for a compiler. DBHIP_FAGG_AUTO=0 disables.
    static const bool fagg_auto_off = exp_env("DBHIP_FAGG_AUTO") && atoi(exp_env("DBHIP_FAGG_AUTO")) == 0;
    // A table that has not seen a row yet tries the kernel OPTIMISTICALLY, without the probing chunk, when the kernel already
    // exists (no compile is started for a shape whose cardinality is unknown): a workgroup that meets a 9th group stops at
    // once and nothing is merged, so a high-cardinality block loses a few microseconds and goes on to probe as before.
    const bool fresh = !g->fast_trusted && g->rows_seen == 0 && g->count_host == 0;
    if (!fagg_auto_off && (g->fast_trusted || fresh) && !g->fagg_disabled && g->count_host <= 8 && n - *done >= (1 << 20)) {
      rc = dbhip_fagg_add_columns_internal(g, C, *done, n - *done, /*may_compile=*/g->fast_trusted != 0, s);
      if (rc == DBHIP_OK) {
        g->rows_seen += n - *done;
        *done = n;
        g->fast_trusted = 1;   // <= 8 groups per workgroup certainly fit a workgroup's LDS table
        return DBHIP_OK;
      }
      if (rc != DBHIP_ERR_CAPACITY && rc != DBHIP_ERR_UNSUPPORTED) return rc;
      // CAPACITY: too many groups for this kernel, for good. UNSUPPORTED: the shape is outside it for good — unless the refusal
      // only says "no kernel yet" (being compiled, or a fresh table that may not start a compile)
      if (rc == DBHIP_ERR_CAPACITY || (!dbhip_fagg_last_refusal_is_pending_internal() && !fresh)) g->fagg_disabled = 1;
    }
    // The first chunk of a big block is a small probe of the key distribution; when it spills
    // (almost) nothing the rest of the block is one launch (its spill buffer is sized for the worst
    // case but stays untouched), otherwise bounded chunks keep re-checking the spill ratio.
    int64_t limit = CHUNK;
    // probing chunk: 256 K rows through the 2-rows-per-lane kernel (64 workgroups x 8 tiles of 512 rows: what is to be
    // learnt is whether the groups fit a workgroup's table, and a workgroup's 8 tiles take a quarter of the time of 8 tiles
    // of 2048 rows — the probe runs on a quarter of the chip, r02m: 0.16 ms at 4 groups, 0.73 ms at 1000)
    const bool probing = !g->fast_trusted && n - *done > (4 << 20);
    if (probing) limit = 1 << 18;
    else if (g->fast_trusted) limit = n;
    const bool small = small_layout && !probing;
    static const int small_r = exp_env("DBHIP_LDS_R") ? atoi(exp_env("DBHIP_LDS_R")) : 4;   // 4 (116 VGPRs, 4 waves / SIMD) or 8 (178, 2): r02n 1.00 vs 1.68 ms at 4 groups
    // BIG table (r03): a small layout whose groups outgrew the 48 KB table (768 groups of 4 words) but fit one twice the size
    // runs ONE 1024-thread workgroup per CU on a 96 KB table (the same 4 waves per SIMD) instead of going through the
    // partitioning passes — 1000 groups: 1.97 ms partitioned, see DESIGN §2.3
    const bool big = small && g->lds_big && !gbc;
    // compact kernel: ONE 1024-thread workgroup per CU, 4 rows per lane, a table sized for the groups the probing chunk predicted
    // (the largest table, gbc_max_lcap = 4096 slots / 112 KB for key + sum + count, while nothing is known)
    const int gbc_lcap = gbc ? (g->gbc_lcap ? g->gbc_lcap : gbc_max_lcap(GD)) : 0;   // (nothing known yet: the largest table)
    const int R = gbc ? gbc_rows_per_lane(gbc_row_words(GD)) : (small ? ((small_r == 4 || big) ? 4 : 8) : 2);
    const int threads = (big || gbc) ? 1024 : 256;
    const int lcap_i = gbc ? gbc_lcap : (big ? lcap * 2 : lcap);
    const size_t lds_i = gbc ? gbc_agg_lds_bytes(GD, gbc_lcap, GBC_T) : (big ? lds_bytes * 2 : lds_bytes);
    const int max_grid_i = (big || gbc) ? 256 : max_grid;
    const int64_t tile_rows = (int64_t)threads * R;
    const int64_t cn = n - *done < limit ? n - *done : limit;
    const int64_t ntiles = ceil_div(cn, tile_rows);
    int grid = (int)(ntiles < max_grid_i ? ntiles : max_grid_i);
    // (the compact kernel's probing chunk: one 4096-row tile per workgroup — the table takes a tile's groups whatever they are, nothing
    // spills, and what is learnt is the number of groups in the chunk, not a spill ratio; r04f: 0.2 ms at 10^4 groups with 16 workgroups
    // x 4 tiles, most rows of which met full tables)
    if (!gbc && !g->fast_trusted && ntiles >= 64) {
      // probing chunk: >= 4 (8) tiles per workgroup, so that its spill ratio measures the key distribution and
      // not the tile size (one tile per workgroup pre-aggregates nothing once groups ~ rows per tile)
      const int64_t gmax = probing ? ntiles / 4 : ntiles / 8;   // (probing: 4 tiles of 512 rows against a table of 768 groups tell as much)
      if (grid > gmax) grid = (int)gmax;
    }
    const int64_t tpb = ceil_div(ntiles, grid);
    grid = (int)ceil_div(ntiles, tpb);
    if ((rc = ensure((void**)&g->partial, &g->partial_cap, (size_t)grid * lcap_i * L.W * 8))) return rc;
    // spill buffer: worst case (every row) for probing chunks; a trusted chunk spilled < 1 % last time, so 1/64 of its rows
    // (at least 4 M) is ample — and a 600 M-row block does not allocate a 72 GB buffer it never touches. Overflow is
    // detected (ctrl[3] bit 2) and the chunk redone.
    int64_t spill_cap = cn;
    if (g->fast_trusted && cn > (4 << 20)) spill_cap = cn / 64 > (4 << 20) ? cn / 64 : (4 << 20);
    if ((rc = ensure((void**)&g->rows_in, &g->rows_in_cap, (size_t)spill_cap * L.W * 8))) return rc;
    DBHIP_CHECK(hipMemsetAsync(&g->ctrl[5], 0, 16, s));
    FkArgs A;
    A.spill_cap = (uint64_t)spill_cap;
    A.row0 = *done; A.n = cn; A.tiles_per_block = tpb; A.lcap = lcap_i; A.sw = sw;
    A.llimit = (uint32_t)(lcap_i - lcap_i / 4);
    A.hash_mask = g->hash_mask; A.partial = g->partial; A.spill = g->rows_in; A.ctrl = g->ctrl;
    if (gbc) {
      // > 64 KB of dynamic LDS needs the attribute once per process and kernel
      static std::once_flag gbc_attr_once;
      static hipError_t gbc_attr_err = hipSuccess;
      std::call_once(gbc_attr_once, [] {
#define GBC_RAISE(KW_, NV_) if (gbc_attr_err == hipSuccess) gbc_attr_err = hipFuncSetAttribute((const void*)gbc_agg_kernel<KW_, NV_, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        GBC_FOR_ALL(GBC_RAISE)
#undef GBC_RAISE
      });
      DBHIP_CHECK(gbc_attr_err);
      GbcAggArgs G;
      memset(&G, 0, sizeof(G));
      G.row0 = *done; G.n = cn; G.lcap = lcap_i; G.llimit = A.llimit; G.partial = g->partial; G.pcount = nullptr;
      G.spill = g->rows_in; G.spill_cap = (uint64_t)spill_cap; G.ctrl = g->ctrl;
#define GBC_AGG(KW_, NV_) hipLaunchKernelGGL((gbc_agg_kernel<KW_, NV_, true>), dim3(grid), dim3(1024), lds_i, s, GD, C, G)
      GBC_DISPATCH(GD, GBC_AGG);
#undef GBC_AGG
    } else if (big) {
      static std::once_flag attr_once;   // > 64 KB of dynamic LDS needs the attribute once per process
      static hipError_t attr_err = hipSuccess;
      std::call_once(attr_once, [] { attr_err = hipFuncSetAttribute((const void*)gb_lds_preagg_kernel<2, 2, false, 4, 1024>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024); });
      DBHIP_CHECK(attr_err);
      hipLaunchKernelGGL((gb_lds_preagg_kernel<2, 2, false, 4, 1024>), dim3(grid), dim3(1024), lds_i, s, L, C, A);
    } else if (small && R == 4) hipLaunchKernelGGL((gb_lds_preagg_kernel<2, 2, false, 4>), dim3(grid), dim3(256), lds_bytes, s, L, C, A);
    else if (small) hipLaunchKernelGGL((gb_lds_preagg_kernel<2, 2, false, 8>), dim3(grid), dim3(256), lds_bytes, s, L, C, A);
    else hipLaunchKernelGGL((gb_lds_preagg_kernel<FK_MAXKW, FK_MAXA, true, 2>), dim3(grid), dim3(256), lds_bytes, s, L, C, A);
    DBHIP_LAUNCH_CHECK();
    uint64_t hc[8];
    DBHIP_CHECK(hipMemcpyAsync(hc, g->ctrl, sizeof(hc), hipMemcpyDeviceToHost, s));
    DBHIP_CHECK(hipStreamSynchronize(s));
    if (hc[3] & 2) {
      // a string key longer than 12 bytes: the LDS kernel's rows are two words per string; nothing of this chunk has been
      // merged — the row path (which keeps long strings in the table's arena) takes the block from here
      DBHIP_CHECK(hipMemsetAsync(&g->ctrl[3], 0, 8, s));
      g->has_long = 1; g->fast_disabled = 1; g->part_bits = -1;
      return -1;
    }
    if (hc[3] & 4) {
      // the trusted chunk spilled past its buffer (the key distribution changed inside the block): nothing of this
      // chunk has been merged yet — drop its output, stop trusting, and redo it in bounded probing chunks
      DBHIP_CHECK(hipMemsetAsync(&g->ctrl[3], 0, 8, s));
      g->fast_trusted = 0;
      continue;
    }
    if ((rc = merge_rows(g, g->partial, (int64_t)hc[5], s))) return rc;
    if ((rc = merge_rows(g, g->rows_in, (int64_t)hc[6], s))) return rc;
    *done += cn;
    g->rows_seen += cn;
    // most rows spilled: the LDS table is too small for this key distribution -> partition by hash
    // bits so that each partition fits, or (high cardinality) leave the rest to the row path
    // more groups than a workgroup's table takes (it would run full everywhere and hand most rows on)
    const bool too_many = g->count_host * 8 > (int64_t)A.llimit * 7;
    if (gbc && cn >= 65536) {
      // size the workgroups' tables for the groups the rows seen so far predict (load <= 0.6); more than the largest table holds:
      // partition (below)
      int64_t est = estimate_groups(g->count_host, g->rows_seen);
      if (est > ((int64_t)1 << 40)) est = (int64_t)1 << 40;   // ("all distinct so far" comes back as a huge number)
      // (a quarter full where the LDS allows it: a lane's first probe then settles ~9 rows in 10, and the rest walk one slot on)
      int want = 256;
      while (want < gbc_max_lcap(GD) && (int64_t)want < est * 4) want *= 2;
      const bool fits = (int64_t)want * 6 >= est * 10;
      if (fits && want != gbc_lcap) {
        g->gbc_lcap = want;
        if (getenv("DBHIP_TRACE")) fprintf(stderr, "[dbhip] groupby: %lld groups in %lld rows -> ~%lld groups: compact LDS table of %d slots\n",
                                           (long long)g->count_host, (long long)g->rows_seen, (long long)est, want);
      }
      if (fits) {
        g->fast_trusted = (int64_t)hc[6] * 100 <= cn || want > gbc_lcap;
        continue;
      }
    }
    if (((int64_t)hc[6] * 10 > cn || too_many) && cn >= 65536) {
      // twice the table is enough (estimated from the groups met so far): stay on the LDS path with the big table
      const int64_t big_limit = (int64_t)(lcap * 2 - lcap / 2) * 7 / 8;
      static const bool big_off = exp_env("DBHIP_LDS_BIG") && atoi(exp_env("DBHIP_LDS_BIG")) == 0;
      if (small_layout && !g->lds_big && !big_off && lds_bytes * 2 <= 128 * 1024 && estimate_groups(g->count_host, g->rows_seen) <= big_limit) {
        g->lds_big = 1;
        g->fast_trusted = 1;
        if (getenv("DBHIP_TRACE")) fprintf(stderr, "[dbhip] groupby: %lld groups in %lld rows -> the 96 KB LDS table\n", (long long)g->count_host, (long long)g->rows_seen);
        continue;
      }
      g->lds_big = 0;
      decide_partitioning(g, g->count_host, g->rows_seen, n);
      if (g->part_bits < 0) g->fast_disabled = 1;
    }
    if (getenv("DBHIP_TRACE")) fprintf(stderr, "[dbhip] groupby lds chunk: rows=%lld partial=%llu spilled=%llu groups=%lld -> pbits=%d\n",
                                       (long long)cn, (unsigned long long)hc[5], (unsigned long long)hc[6], (long long)g->count_host, g->part_bits);
    g->fast_trusted = !too_many && (int64_t)hc[6] * 100 <= cn;
  }
  return DBHIP_OK;
}

// ---------------------------------------------------------------------------
// Radix-partitioned pre-aggregation (medium cardinality: ~10^3 .. ~10^6 groups).
//
// Between "fits one workgroup's LDS table" and "every row is its own group" the row path contends on
// hot addresses (global atomics serialise per address) and the LDS path spills. The reference meets
// the same regime with radix-partitioned payloads (PartitionedPayload, partitioned_payload.rs:34-60,
// 160-240: partition = hash bits, each partition aggregated on its own); the device analogue:
//
//   hist     rows per partition (partition = top `pbits` bits of the group hash), LDS histogram per
//            workgroup, one global atomic per (workgroup, non-empty partition)
//   scan     exclusive scan of the <= 1024 counts (one workgroup)
//   scatter  tiles of 8192 rows: rank inside the tile by LDS atomics, ONE global cursor atomic per
//            (tile, partition), rows serialized straight into their partition's region
//   aggregate one workgroup per (partition, split): LDS hash table exactly as the pre-aggregation
//            kernel above (claim by hash, verify after the barrier, LDS atomics), <= lcap partial rows
//            per workgroup, merged into the HBM table by the row path; rows that do not fit are
//            listed and go through the row path too.
// Generic over the layout (rows are handled as W words in memory).
// ---------------------------------------------------------------------------
constexpr int PT_THREADS = 1024;
constexpr int PT_R = 4;

__device__ __forceinline__ uint64_t gb_keys_hash(const GbLayout& L, const GbCols& C, int64_t i, uint64_t* ctrl) {
  uint64_t h = 0;
  for (int k = 0; k < L.nkeys; ++k) {
    uint64_t w[2];
    bool valid;
    if (!gb_load_words(C.key[k], i, w, &valid)) atomicOr((unsigned long long*)&ctrl[3], 2ULL);
    const uint64_t hk = gb_hash_words(L.key_type[k], w, valid);
    h = (k == 0) ? hk : merge_hash(h, hk);
  }
  return h;
}


// group hashes of R rows, column by column (gb_load_words_n: the R loads of a column are in flight together and the
// layout is decoded once per column, not once per row)
template <int R>
__device__ __forceinline__ void gb_keys_hash_n(const GbLayout& L, const GbCols& C, const int64_t (&row)[R], uint64_t (&h)[R], uint64_t* ctrl) {
#pragma unroll
  for (int x = 0; x < R; ++x) h[x] = 0;
  for (int k = 0; k < L.nkeys; ++k) {
    uint64_t w0[R], w1[R];
    bool valid[R];
    if (!gb_load_words_n<R>(C.key[k], row, w0, w1, valid)) atomicOr((unsigned long long*)&ctrl[3], 2ULL);
    const int type = L.key_type[k];
#pragma unroll
    for (int x = 0; x < R; ++x) {
      const uint64_t w[2] = {w0[x], w1[x]};
      const uint64_t hk = gb_hash_words(type, w, valid[x]);
      h[x] = (k == 0) ? hk : merge_hash(h[x], hk);
    }
  }
}
// serialized images (gb_serialize_row) of R rows whose hashes are known, written to out[x] — column by column
template <int R>
__device__ __forceinline__ void gb_serialize_rows_n(const GbLayout& L, const GbCols& C, const int64_t (&row)[R], const uint64_t (&h)[R],
                                                    uint64_t* const (&out)[R]) {
  uint64_t vmask[R];
#pragma unroll
  for (int x = 0; x < R; ++x) vmask[x] = 0;
  for (int k = 0; k < L.nkeys; ++k) {
    uint64_t w0[R], w1[R];
    bool valid[R];
    gb_load_words_n<R>(C.key[k], row, w0, w1, valid);
    const int off = L.key_off[k];
    const bool two = L.key_words[k] == 2;
#pragma unroll
    for (int x = 0; x < R; ++x) {
      out[x][off] = w0[x];
      if (two) out[x][off + 1] = w1[x];
      if (valid[x]) vmask[x] |= 1ULL << k;
    }
  }
  if (L.validity_word >= 0) {
#pragma unroll
    for (int x = 0; x < R; ++x) out[x][L.validity_word] = vmask[x];
  }
#pragma unroll
  for (int x = 0; x < R; ++x) out[x][L.hash_word] = h[x];
  for (int a = 0; a < L.naggs; ++a) {
    uint64_t w0[R], w1[R];
    bool valid[R];
#pragma unroll
    for (int x = 0; x < R; ++x) { w0[x] = 0; w1[x] = 0; valid[x] = true; }
    if (C.arg[a].data != nullptr) gb_load_words_n<R>(C.arg[a], row, w0, w1, valid);
    const int off = L.agg_off[a], nw = L.agg_words[a];
#pragma unroll
    for (int x = 0; x < R; ++x) {
      uint64_t v[GB_MAX_STATE_WORDS];
      gb_row_contrib(L, a, w0[x], w1[x], valid[x], v);
      for (int k = 0; k < nw; ++k) out[x][off + k] = v[k];
    }
  }
}

// hist: workgroup b counts the rows of ITS row range [b * rows_per_wg, ...) per partition (LDS histogram) into
// mat[b][0..P) — the scatter kernel walks the same ranges, so after the scans below mat[b][p] is the first output row of
// workgroup b's run inside partition p and the scatter needs no global cursor (one device-scope atomic per (tile, partition)
// is one per ROW once the partitions outnumber a tile's rows — the cost the partitioning is there to avoid).
__global__ __launch_bounds__(PT_THREADS) void gb_part_hist_kernel(GbLayout L, GbCols C, int64_t row0, int64_t n, int pbits,
                                                                  int64_t rows_per_wg, uint32_t* mat, uint64_t* ctrl) {
  extern __shared__ uint32_t pt_lds[];
  const int P = 1 << pbits;
  const int T = blockDim.x;
  for (int s = threadIdx.x; s < P; s += T) pt_lds[s] = 0;
  __syncthreads();
  const int64_t lo = (int64_t)blockIdx.x * rows_per_wg;
  const int64_t hi = lo + rows_per_wg < n ? lo + rows_per_wg : n;
  for (int64_t t0 = lo; t0 < hi; t0 += (int64_t)T * PT_R) {
    int64_t row[PT_R];
    bool in[PT_R];
    uint64_t h[PT_R];
#pragma unroll
    for (int x = 0; x < PT_R; ++x) {
      const int64_t li = t0 + (int64_t)x * T + threadIdx.x;
      in[x] = li < hi;
      row[x] = row0 + (in[x] ? li : lo);
    }
    gb_keys_hash_n<PT_R>(L, C, row, h, ctrl);
#pragma unroll
    for (int x = 0; x < PT_R; ++x)
      if (in[x] && gb_row_passes(C, row[x])) atomicAdd(&pt_lds[part_of(h[x], pbits)], 1u);
  }
  __syncthreads();
  uint32_t* out = mat + (size_t)blockIdx.x * P;
  for (int s = threadIdx.x; s < P; s += T) out[s] = pt_lds[s];
}

// One workgroup per 64 partitions, 4 lanes per partition (each a quarter of the nwg workgroup rows of the matrix, loads
// coalesced over the 64 partitions). FINAL = false: tot[p] = sum over workgroups; FINAL = true: mat[b][p] <- base[p] +
// sum of mat[b'][p] for b' < b.
template <bool FINAL>
__global__ __launch_bounds__(256) void gb_part_colscan_kernel(uint32_t* mat, int P, int nwg, uint32_t* tot, const uint32_t* base) {
  __shared__ uint32_t seg[4][64];
  const int pl = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int p = blockIdx.x * 64 + pl;
  const int per = (nwg + 3) / 4;
  const int b0 = q * per, b1 = (b0 + per < nwg) ? b0 + per : nwg;
  uint32_t sum = 0;
  if (p < P)
    for (int b = b0; b < b1; ++b) sum += mat[(size_t)b * P + p];
  seg[q][pl] = sum;
  __syncthreads();
  if (!FINAL) {
    if (q == 0 && p < P) tot[p] = seg[0][pl] + seg[1][pl] + seg[2][pl] + seg[3][pl];
    return;
  }
  if (p >= P) return;
  uint32_t run = base[p];
  for (int k = 0; k < q; ++k) run += seg[k][pl];
  for (int b = b0; b < b1; ++b) {
    const uint32_t c = mat[(size_t)b * P + p];
    mat[(size_t)b * P + p] = run;
    run += c;
  }
}

// base[0..P] = exclusive scan of hist[0..P)   (P <= 16384, one workgroup of 1024, 16 entries per thread)
__global__ __launch_bounds__(1024) void gb_part_scan_kernel(const uint32_t* hist, int P, uint32_t* base) {
  __shared__ uint32_t wave_tot[16];
  const int t = threadIdx.x;
  constexpr int E = PT_PMAX / 1024;
  uint32_t v[E], tsum = 0;
#pragma unroll
  for (int k = 0; k < E; ++k) {
    v[k] = (t * E + k) < P ? hist[t * E + k] : 0;
    tsum += v[k];
  }
  uint32_t incl = tsum;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t o = __shfl_up(incl, d, 64);
    if (lane_id() >= d) incl += o;
  }
  if (lane_id() == 63) wave_tot[t >> 6] = incl;
  __syncthreads();
  uint32_t wbase = 0;
  for (int k = 0; k < (t >> 6); ++k) wbase += wave_tot[k];
  uint32_t run = wbase + incl - tsum;
#pragma unroll
  for (int k = 0; k < E; ++k) {
    const int i = t * E + k;
    if (i < P) base[i] = run;
    run += v[k];
    if (i == P - 1) base[P] = run;
  }
}

// serialized image of input row i (same encoding as gb_serialize_kernel) written to `r`
__device__ __forceinline__ void gb_serialize_row(const GbLayout& L, const GbCols& C, int64_t i, uint64_t* r,
                                                 uint64_t* ctrl) {
  uint64_t h = 0, vmask = 0;
  for (int k = 0; k < L.nkeys; ++k) {
    uint64_t w[2];
    bool valid;
    if (!gb_load_words(C.key[k], i, w, &valid)) atomicOr((unsigned long long*)&ctrl[3], 2ULL);
    const uint64_t hk = gb_hash_words(L.key_type[k], w, valid);
    h = (k == 0) ? hk : merge_hash(h, hk);
    r[L.key_off[k]] = w[0];
    if (L.key_words[k] == 2) r[L.key_off[k] + 1] = w[1];
    if (valid) vmask |= 1ULL << k;
  }
  if (L.validity_word >= 0) r[L.validity_word] = vmask;
  r[L.hash_word] = h;
  for (int a = 0; a < L.naggs; ++a) {
    uint64_t w[2] = {0, 0};
    bool valid = true;
    if (C.arg[a].data != nullptr) gb_load_words(C.arg[a], i, w, &valid);
    uint64_t v[GB_MAX_STATE_WORDS];
    gb_row_contrib(L, a, w[0], w[1], valid, v);
    for (int k = 0; k < L.agg_words[a]; ++k) r[L.agg_off[a] + k] = v[k];
  }
}

// the same image for a row whose hash is known (the keys are loaded again, from the L1, but not hashed again)
__device__ __forceinline__ void gb_serialize_row_hashed(const GbLayout& L, const GbCols& C, int64_t i, uint64_t h, uint64_t* r) {
  uint64_t vmask = 0;
  for (int k = 0; k < L.nkeys; ++k) {
    uint64_t w[2];
    bool valid;
    gb_load_words(C.key[k], i, w, &valid);
    r[L.key_off[k]] = w[0];
    if (L.key_words[k] == 2) r[L.key_off[k] + 1] = w[1];
    if (valid) vmask |= 1ULL << k;
  }
  if (L.validity_word >= 0) r[L.validity_word] = vmask;
  r[L.hash_word] = h;
  for (int a = 0; a < L.naggs; ++a) {
    uint64_t w[2] = {0, 0};
    bool valid = true;
    if (C.arg[a].data != nullptr) gb_load_words(C.arg[a], i, w, &valid);
    uint64_t v[GB_MAX_STATE_WORDS];
    gb_row_contrib(L, a, w[0], w[1], valid, v);
    for (int k = 0; k < L.agg_words[a]; ++k) r[L.agg_off[a] + k] = v[k];
  }
}

// scatter: workgroup b walks the row range it counted in the histogram kernel; lcur[p] (LDS) = next output row of its run in
// partition p, so a row's place is ONE LDS atomic and there is no global atomic in the loop.
// STAGED: the rows of a batch (one per thread) are serialized into LDS first and copied out by the whole workgroup, word by
// word in row order — a row's W words leave as one contiguous piece (and neighbours in a run as one longer piece) instead of
// W separate 8-byte stores per lane, each its own request to the L1 (r02o: 1.28 ms per 60 M rows at 16 partitions, 3.0 ms at
// 16384; the kernel was bound by the number of store requests, not by bytes or by the hash).
template <int PS_R>   // rows per thread of a staged batch; 0 = not staged
__global__ __launch_bounds__(PT_THREADS) void gb_part_scatter_kernel(GbLayout L, GbCols C, int64_t row0, int64_t n,
                                                                     int pbits, int64_t rows_per_wg, const uint32_t* mat,
                                                                     uint64_t* rows_out, uint64_t* ctrl) {
  extern __shared__ uint32_t pt_lds[];
  const int P = 1 << pbits;
  uint32_t* lcur = pt_lds;
  const int tid = threadIdx.x;
  const int T = blockDim.x;   // 256 (few partitions: several workgroups per CU overlap their load / stage / copy-out phases) or 1024
  const uint32_t* mine = mat + (size_t)blockIdx.x * P;
  for (int s = tid; s < P; s += T) lcur[s] = mine[s];
  __syncthreads();
  const int64_t lo = (int64_t)blockIdx.x * rows_per_wg;
  const int64_t hi = lo + rows_per_wg < n ? lo + rows_per_wg : n;
  if (PS_R == 0) {
    for (int64_t t0 = lo; t0 < hi; t0 += (int64_t)T * PT_R) {
#pragma unroll
      for (int x = 0; x < PT_R; ++x) {
        const int64_t li = t0 + (int64_t)x * T + tid;
        if (li < hi && gb_row_passes(C, row0 + li)) {
          const uint64_t h = gb_keys_hash(L, C, row0 + li, ctrl);
          const uint32_t pos = atomicAdd(&lcur[part_of(h, pbits)], 1u);
          gb_serialize_row_hashed(L, C, row0 + li, h, rows_out + (uint64_t)pos * L.W);
        }
      }
    }
    return;
  }
  const int SW = L.W | 1;                                      // odd stride in 8-byte words: conflict-free rows
  constexpr int SR = PS_R > 0 ? PS_R : 1;
  const int BR = T * SR;                                       // rows of a batch
  uint32_t* gpos = pt_lds + P;                                 // [BR] output row of the staged row, ~0 = none
  uint64_t* stage = (uint64_t*)(pt_lds + P + BR);              // [BR][SW]   (P and BR are even: 8-byte aligned)
  int wshift = 0;
  while ((1 << wshift) < L.W) ++wshift;                         // copy-out: 2^wshift lanes per row, lanes >= W idle
  const int k = tid & ((1 << wshift) - 1), rsub = tid >> wshift;
  const int rows_per_it = T >> wshift;
  for (int64_t t0 = lo; t0 < hi; t0 += BR) {
    int64_t row[SR];
    bool in[SR];
    uint64_t h[SR];
    uint64_t* out[SR];
#pragma unroll
    for (int x = 0; x < SR; ++x) {
      const int64_t li = t0 + (int64_t)x * T + tid;
      in[x] = li < hi;
      row[x] = row0 + (in[x] ? li : lo);
      out[x] = stage + (size_t)(x * T + tid) * SW;
    }
    gb_keys_hash_n<SR>(L, C, row, h, ctrl);
#pragma unroll
    for (int x = 0; x < SR; ++x) {
      uint32_t pos = 0xFFFFFFFFu;
      if (in[x] && gb_row_passes(C, row[x])) pos = atomicAdd(&lcur[part_of(h[x], pbits)], 1u);
      gpos[x * T + tid] = pos;
    }
    gb_serialize_rows_n<SR>(L, C, row, h, out);   // (rows that do not take part fill their own staging row and stay there)
    __syncthreads();
    if (k < L.W) {
      for (int r = rsub; r < BR; r += rows_per_it) {
        const uint32_t g = gpos[r];
        if (g != 0xFFFFFFFFu) rows_out[(uint64_t)g * L.W + k] = stage[(size_t)r * SW + k];
      }
    }
    __syncthreads();
  }
}

struct PaArgs {
  const uint64_t* rows;    // [n][W] grouped by partition
  const uint32_t* base;    // [P+1]
  int splits;              // workgroups per partition
  int lcap, sw;
  uint32_t llimit;
  uint64_t hash_mask;
  uint64_t* partial;       // [gridDim.x * lcap][W]
  uint32_t* spill_idx;     // row indices (into rows) that did not fit
  uint64_t* ctrl;          // [5] = #partial rows, [6] = #spilled rows
  uint32_t* pcount;        // non-NULL: workgroup b keeps its partial rows at partial[b * lcap ...] and their number here
                           // (the partition-exclusive merge below reads them per partition); NULL: one packed list
  // heavy partitions (round 5, the scheme of gb_compact.h's GbcAggArgs): partition p is worked on in nsp[p] >= splits sub-ranges, the
  // ones beyond `splits` by EXTRA workgroups (blockIdx.x >= nparts * splits; extra_map[e] = p | sub-range << 16, *extra_n of them);
  // with per-partition lists the partial rows of a split partition go to a packed list at partial[packed_base ...] (cursor ctrl[7])
  const uint32_t* nsp;
  const uint32_t* extra_n;
  const uint32_t* extra_map;
  int nparts;
  uint64_t packed_base;
};

constexpr int PA_R = 4;

// word `idx` (wave-uniform) of a row held in registers: a chain of selects, no dynamic register indexing
template <int N>
__device__ __forceinline__ uint64_t pa_pick(const uint64_t (&a)[N], int idx) {
  uint64_t r = 0;
#pragma unroll
  for (int k = 0; k < N; ++k) r = (k == idx) ? a[k] : r;
  return r;
}

// WMAX > 0: rows of at most WMAX words are loaded whole into registers at the top of a tile (PA_R x W independent loads in
// one block) and every later use is a register; WMAX = 0: any width, words re-read from memory where they are used (each
// such load is a round trip to the L1 behind a branch)
template <int WMAX>
__global__ __launch_bounds__(256) void gb_part_agg_kernel(GbLayout L, PaArgs A) {
  extern __shared__ uint64_t fk_lds[];
  __shared__ uint32_t lcount;
  uint64_t* lhash = fk_lds;
  uint64_t* lrows = fk_lds + A.lcap;
  const int tid = threadIdx.x;
  const uint32_t lmask = (uint32_t)A.lcap - 1;
  int p, sp;
  const int regular = A.nsp ? A.nparts * A.splits : (int)gridDim.x;
  if ((int)blockIdx.x < regular) { p = blockIdx.x / A.splits; sp = blockIdx.x % A.splits; }
  else {
    const uint32_t e = blockIdx.x - (uint32_t)regular;
    if (e >= *A.extra_n) return;
    const uint32_t m = A.extra_map[e];
    p = (int)(m & 0xFFFFu); sp = (int)(m >> 16);
  }
  const uint32_t nsp = A.nsp ? A.nsp[p] : (uint32_t)A.splits;
  const bool own_list = A.pcount && nsp == 1;      // the partition's own list (partition-exclusive merge); else a packed list
  const uint32_t pb = A.base[p], pe = A.base[p + 1];
  const uint32_t len = pe - pb;
  const uint32_t r_begin = pb + (uint32_t)(((uint64_t)len * (uint32_t)sp) / nsp);
  const uint32_t r_end = pb + (uint32_t)(((uint64_t)len * ((uint32_t)sp + 1)) / nsp);
  if (r_begin >= r_end) return;
  for (int s = tid; s < A.lcap; s += 256) lhash[s] = 0;
  if (tid == 0) lcount = 0;
  __syncthreads();

  for (uint32_t t0 = r_begin; t0 < r_end; t0 += 256 * PA_R) {
    uint32_t slot[PA_R];
    constexpr int WR = WMAX > 0 ? WMAX : 1;
    uint64_t rw[PA_R][WR];
    uint64_t hs[PA_R];
    if (WMAX > 0) {
#pragma unroll
      for (int x = 0; x < PA_R; ++x) {
        const uint32_t ri = t0 + x * 256 + tid;
        const uint64_t* r = A.rows + (uint64_t)(ri < r_end ? ri : r_begin) * L.W;
#pragma unroll
        for (int k = 0; k < WR; ++k) rw[x][k] = k < L.W ? r[k] : 0;
      }
#pragma unroll
      for (int x = 0; x < PA_R; ++x) hs[x] = pa_pick<WR>(rw[x], L.hash_word);
    } else {
      // the hashes of all PA_R rows of this thread first: PA_R independent loads in flight instead of one per probe
#pragma unroll
      for (int x = 0; x < PA_R; ++x) {
        const uint32_t ri = t0 + x * 256 + tid;
        hs[x] = ri < r_end ? A.rows[(uint64_t)ri * L.W + L.hash_word] : 0;
      }
    }
    // ---- phase A: match-or-claim by hash ----
#pragma unroll
    for (int x = 0; x < PA_R; ++x) {
      const uint32_t ri = t0 + x * 256 + tid;
      slot[x] = FK_SPILL - 1;  // padding
      if (ri < r_end) {
        const uint64_t* r = A.rows + (uint64_t)ri * L.W;
        const uint64_t h = hs[x];
        const uint64_t hw = probe_word(h, A.hash_mask);
        uint32_t pos = (uint32_t)hw & lmask;
        slot[x] = FK_SPILL;
        for (int step = 0; step < 64; ++step) {
          uint64_t cur = ((volatile uint64_t*)lhash)[pos];
          if (cur == 0) {
            if (((volatile uint32_t*)&lcount)[0] >= A.llimit) break;
            const unsigned long long old = atomicCAS((unsigned long long*)&lhash[pos], 0ULL, (unsigned long long)hw);
            if (old == 0) {
              atomicAdd(&lcount, 1u);
              uint64_t* d = lrows + (size_t)pos * A.sw;
              if (WMAX > 0) {
#pragma unroll
                for (int j = 0; j < WR; ++j)
                  if (j < L.nkey_words) d[j] = rw[x][j];
              } else {
                for (int j = 0; j < L.nkey_words; ++j) d[j] = r[j];
              }
              d[L.hash_word] = h;
              for (int a = 0; a < L.naggs; ++a) gb_state_identity(L, a, d + L.agg_off[a]);
              slot[x] = pos;
              break;
            }
            cur = old;
          }
          if (cur == hw) { slot[x] = pos; break; }
          pos = (pos + 1) & lmask;
        }
      }
    }
    __syncthreads();
    // ---- phase B: verify keys, merge with LDS atomics; the rest is listed for the row path ----
#pragma unroll
    for (int x = 0; x < PA_R; ++x) {
      const uint32_t ri = t0 + x * 256 + tid;
      bool spill = slot[x] == FK_SPILL;
      if (slot[x] < FK_SPILL - 1) {
        const uint64_t* r = A.rows + (uint64_t)ri * L.W;
        uint64_t* d = lrows + (size_t)slot[x] * A.sw;
        bool eq = true;
        if (WMAX > 0) {
#pragma unroll
          for (int j = 0; j < WR; ++j)
            if (j < L.nkey_words) eq &= (d[j] == rw[x][j]);
        } else {
          for (int j = 0; j < L.nkey_words; ++j) eq &= (d[j] == r[j]);
        }
        if (eq) {
          if (WMAX > 0) {
            for (int a = 0; a < L.naggs; ++a) {
              uint64_t v[GB_MAX_STATE_WORDS] = {0, 0, 0, 0};
              const int off = L.agg_off[a], nw = L.agg_words[a];
#pragma unroll
              for (int j = 0; j < GB_MAX_STATE_WORDS; ++j)
                if (j < nw) v[j] = pa_pick<WR>(rw[x], off + j);
              gb_atomic_merge(L, a, d + off, v);
            }
          } else {
            for (int a = 0; a < L.naggs; ++a) gb_atomic_merge(L, a, d + L.agg_off[a], r + L.agg_off[a]);
          }
        } else {
          spill = true;
        }
      }
      const uint64_t m = __ballot(spill);
      if (m) {
        const int leader = __ffsll((long long)m) - 1;
        unsigned long long sb = 0;
        if (lane_id() == leader) sb = atomicAdd((unsigned long long*)&A.ctrl[6], (unsigned long long)__popcll(m));
        sb = __shfl(sb, leader, 64);
        if (spill) A.spill_idx[sb + __popcll(m & ((1ULL << lane_id()) - 1))] = ri;
      }
    }
    // no barrier: the next tile only adds NEW slots (see gb_lds_preagg_kernel)
  }
  __syncthreads();
  const uint32_t occupied = lcount;
  __syncthreads();
  __shared__ unsigned long long pa_wg_base;
  if (tid == 0) {
    lcount = 0;
    if (A.pcount) {
      if (own_list) { A.pcount[blockIdx.x] = occupied; pa_wg_base = (unsigned long long)blockIdx.x * A.lcap; }
      else pa_wg_base = A.packed_base + (occupied ? atomicAdd((unsigned long long*)&A.ctrl[7], (unsigned long long)occupied) : 0ULL);
      if (occupied) atomicAdd((unsigned long long*)&A.ctrl[5], (unsigned long long)occupied);
    }
  }
  __syncthreads();
  for (int s = tid; s < A.lcap; s += 256) {  // lcap is a multiple of 256: wave-uniform
    const bool occ = lhash[s] != 0;
    const uint64_t m = __ballot(occ);
    unsigned long long base = 0;
    if (m && lane_id() == 0) {
      if (A.pcount) base = pa_wg_base + atomicAdd(&lcount, (uint32_t)__popcll(m));
      else base = atomicAdd((unsigned long long*)&A.ctrl[5], (unsigned long long)__popcll(m));
    }
    base = __shfl(base, 0, 64);
    if (occ) {
      const unsigned long long idx = base + __popcll(m & ((1ULL << lane_id()) - 1));
      const uint64_t* src = lrows + (size_t)s * A.sw;
      uint64_t* o = A.partial + idx * L.W;
      for (int k = 0; k < L.W; ++k) o[k] = src[k];
    }
  }
}

__global__ __launch_bounds__(256) void gb_gather_rows_kernel(const uint64_t* rows, const uint32_t* idx, int64_t n, int W,
                                                             uint64_t* out) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const uint64_t* r = rows + (uint64_t)idx[i] * W;
    uint64_t* o = out + i * W;
    for (int k = 0; k < W; ++k) o[k] = r[k];
  }
}

// LDS table geometry of the partition-aggregate kernel for this layout (0 slots = layout too wide)
// Partition-exclusive merge of the aggregation kernel's partial rows into the HBM table.
//
// The partition of a row is the TOP `pbits` bits of its hash and so is the top of its home slot (home_slot): partition p's
// groups start their probe inside slice p = slots [p * cap / P, (p + 1) * cap / P) of the table. One workgroup per
// partition: it alone inserts into and updates groups of its slice during this launch, so the states are merged with plain
// loads and stores — no device-scope atomic per state word, which is what bounds the row path (two fabric atomics per
// row: 17 ms per 60 M rows at 10^7 groups). Only the claim of an empty slot is an atomic (its neighbours may race for the
// same slot). A probe that would leave the slice (chains of the row path may cross a boundary) and a partial row whose
// slot holds other keys (a 64-bit hash collision) are listed in `retry` and go through the row path afterwards.
// Claim, barrier, then verify and merge: a claimed slot's keys are written before the barrier.
struct PmArgs {
  const uint64_t* partial;   // [P * lcap][W]
  const uint32_t* pcount;    // [P]
  int lcap, pbits;
  uint64_t* slot_hash;
  uint64_t* rows;
  int64_t cap;
  uint64_t hash_mask;
  uint32_t* retry;           // partial-row indices for the row path
  uint64_t* ctrl;            // [0] += new groups, [2] += listed rows
};

__global__ __launch_bounds__(256) void gb_part_merge_kernel(GbLayout L, PmArgs A) {
  extern __shared__ uint32_t pm_slot[];   // [lcap]
  __shared__ uint32_t pm_new;
  const int tid = threadIdx.x;
  const int p = blockIdx.x;
  const uint32_t n = A.pcount[p];
  if (n == 0) return;
  if (tid == 0) pm_new = 0;
  __syncthreads();
  const uint64_t* src = A.partial + (size_t)p * A.lcap * L.W;
  const uint64_t slice = (uint64_t)A.cap >> A.pbits;
  const uint64_t hi = ((uint64_t)p + 1) * slice;
  const uint32_t n_pad = (n + 63) & ~63u;
  for (uint32_t i = tid; i < n_pad; i += 256) {
    bool claimed = false;
    if (i < n) {
      const uint64_t* r = src + (size_t)i * L.W;
      const uint64_t hw = probe_word(r[L.hash_word], A.hash_mask);
      uint64_t pos = home_slot(hw, A.cap);
      uint32_t found = GB_INVALID_SLOT;
      for (; pos < hi; ++pos) {
        // workgroup scope: the slice has no other reader or writer during this launch, and a device-scope atomic is a trip
        // through the fabric (the L2s of the eight XCDs are not coherent with each other) — r02n: 3.4 ms per 4.7 M rows
        unsigned long long cur = A.slot_hash[pos];   // plain: a stale 0 only leads to the CAS, which returns the real content
        if (cur == 0) {
          unsigned long long old = 0ULL;
          __hip_atomic_compare_exchange_strong((unsigned long long*)&A.slot_hash[pos], &old, (unsigned long long)hw, __ATOMIC_RELAXED,
                                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          if (old == 0) {
            // a new group: the partial row IS its state — written whole, nothing to merge in the second phase (r04: the table of
            // 10^7 groups is far larger than any cache, every touch of a row is its own HBM sector; claim + identity + read back +
            // merge were ~7 of them per group, this is 3)
            uint64_t* d = A.rows + pos * L.W;
            for (int k = 0; k < L.W; ++k) d[k] = r[k];
            claimed = true;
            found = GB_INVALID_SLOT - 1;   // done
            break;
          }
          cur = old;
        }
        if (cur == hw) { found = (uint32_t)pos; break; }
      }
      pm_slot[i] = found;
    }
    const uint64_t m = __ballot(claimed);
    if (m && lane_id() == 0) atomicAdd(&pm_new, (uint32_t)__popcll(m));
  }
  __syncthreads();   // (a workgroup barrier orders this workgroup's global stores before its later loads: one CU, one L1)
  // the new groups of the WORKGROUP in one atomic: ctrl[0] is one address, and an atomic per wave and pass — 131 K of them at 16384
  // partitions — serialises at ~9 ns each (r04j: 1.2 of the 1.5 ms of this kernel per 5 M partial rows)
  if (tid == 0 && pm_new) atomicAdd((unsigned long long*)&A.ctrl[0], (unsigned long long)pm_new);
  for (uint32_t i = tid; i < n_pad; i += 256) {
    bool listed = false;
    if (i < n) {
      const uint64_t* r = src + (size_t)i * L.W;
      const uint32_t pos = pm_slot[i];
      listed = pos == GB_INVALID_SLOT;
      if (!listed && pos != GB_INVALID_SLOT - 1) {
        uint64_t* d = A.rows + (uint64_t)pos * L.W;
        bool eq = true;
        for (int k = 0; k < L.nkey_words; ++k) eq &= (d[k] == r[k]);
        if (eq) {
          for (int a = 0; a < L.naggs; ++a) gb_plain_merge(L, a, d + L.agg_off[a], r + L.agg_off[a]);
        } else {
          listed = true;
        }
      }
    }
    const uint64_t m = __ballot(listed);
    if (m) {
      unsigned long long base = 0;
      if (lane_id() == 0) base = atomicAdd((unsigned long long*)&A.ctrl[2], (unsigned long long)__popcll(m));
      base = __shfl(base, 0, 64);
      if (listed) A.retry[base + __popcll(m & ((1ULL << lane_id()) - 1))] = (uint32_t)((size_t)p * A.lcap + i);
    }
  }
}

// Partition-exclusive INSERT: the same ownership as gb_part_merge_kernel, applied to the partition's input rows themselves —
// for key distributions where a partition's groups do not fit an LDS table (about as many groups as rows: nothing to
// pre-aggregate). Workgroup p walks its rows [base[p], base[p+1]) in tiles: phase A finds or claims each row's slot in
// slice p (workgroup-scope CAS), barrier, phase B verifies the keys and merges the state contribution with workgroup-scope
// atomics (several rows of a tile may belong to one group). Rows that leave the slice or meet other keys under their hash are
// listed for the row path.
// (r02t tried the other ownership split — 1024 buckets of the next hash bits, one thread per bucket walking its rows with
// plain loads and stores, CAS only to claim: 8.0 ms per 60 M rows at 10^7 groups against 6.3 ms for this kernel; the serial
// dependent chain per thread costs more than the atomics it saves. r02s counters for this kernel: 2.3 atomics and 2.6 L2
// misses per row, 5.3 GB written per 60 M rows — global atomics are executed memory-side whatever their scope.)
struct PiArgs {
  const uint64_t* rows;    // [n][W] grouped by partition
  const uint32_t* base;    // [P+1]
  int pbits;
  uint64_t* slot_hash;
  uint64_t* table;
  int64_t cap;
  uint64_t hash_mask;
  uint32_t* spill_idx;     // rows (indices into `rows`) for the row path
  uint64_t* ctrl;          // [0] += new groups, [6] += listed rows
};
constexpr int PI_R = 4;

__global__ __launch_bounds__(256) void gb_part_insert_kernel(GbLayout L, PiArgs A) {
  __shared__ uint32_t wg_new;
  const int tid = threadIdx.x;
  const int p = blockIdx.x;
  const uint32_t r_begin = A.base[p], r_end = A.base[p + 1];
  if (r_begin >= r_end) return;
  if (tid == 0) wg_new = 0;
  const uint64_t slice = (uint64_t)A.cap >> A.pbits;
  const uint64_t hi = ((uint64_t)p + 1) * slice;
  uint32_t my_new = 0;
  for (uint32_t t0 = r_begin; t0 < r_end; t0 += 256 * PI_R) {
    uint32_t slot[PI_R];
    uint64_t hs[PI_R];
#pragma unroll
    for (int x = 0; x < PI_R; ++x) {
      const uint32_t ri = t0 + x * 256 + tid;
      hs[x] = ri < r_end ? A.rows[(uint64_t)ri * L.W + L.hash_word] : 0;
    }
#pragma unroll
    for (int x = 0; x < PI_R; ++x) {
      const uint32_t ri = t0 + x * 256 + tid;
      slot[x] = GB_INVALID_SLOT - 1;   // padding
      if (ri < r_end) {
        const uint64_t* r = A.rows + (uint64_t)ri * L.W;
        const uint64_t hw = probe_word(hs[x], A.hash_mask);
        slot[x] = GB_INVALID_SLOT;
        for (uint64_t pos = home_slot(hw, A.cap); pos < hi; ++pos) {
          unsigned long long cur = __hip_atomic_load((unsigned long long*)&A.slot_hash[pos], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          if (cur == 0) {
            unsigned long long old = 0ULL;
            __hip_atomic_compare_exchange_strong((unsigned long long*)&A.slot_hash[pos], &old, (unsigned long long)hw, __ATOMIC_RELAXED,
                                                 __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (old == 0) {
              uint64_t* d = A.table + pos * L.W;
              for (int k = 0; k < L.nkey_words; ++k) d[k] = r[k];
              d[L.hash_word] = hs[x];
              for (int a = 0; a < L.naggs; ++a) gb_state_identity(L, a, d + L.agg_off[a]);
              ++my_new;
              slot[x] = (uint32_t)pos;
              break;
            }
            cur = old;
          }
          if (cur == hw) { slot[x] = (uint32_t)pos; break; }
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int x = 0; x < PI_R; ++x) {
      const uint32_t ri = t0 + x * 256 + tid;
      bool listed = slot[x] == GB_INVALID_SLOT;
      if (slot[x] < GB_INVALID_SLOT - 1) {
        const uint64_t* r = A.rows + (uint64_t)ri * L.W;
        uint64_t* d = A.table + (uint64_t)slot[x] * L.W;
        bool eq = true;
        for (int k = 0; k < L.nkey_words; ++k) eq &= (d[k] == r[k]);
        if (eq) {
          for (int a = 0; a < L.naggs; ++a) gb_wg_merge(L, a, d + L.agg_off[a], r + L.agg_off[a]);
        } else {
          listed = true;
        }
      }
      const uint64_t m = __ballot(listed);
      if (m) {
        unsigned long long sb = 0;
        if (lane_id() == 0) sb = atomicAdd((unsigned long long*)&A.ctrl[6], (unsigned long long)__popcll(m));
        sb = __shfl(sb, 0, 64);
        if (listed) A.spill_idx[sb + __popcll(m & ((1ULL << lane_id()) - 1))] = ri;
      }
    }
    // no barrier: the next tile's claims touch other slots' keys only; keys of slots matched above never change
  }
  if (my_new) atomicAdd(&wg_new, my_new);
  __syncthreads();
  if (tid == 0 && wg_new) atomicAdd((unsigned long long*)&A.ctrl[0], (unsigned long long)wg_new);
}

// hist -> scans -> scatter with compact rows: rows [row0, row0 + cn) into g->rows_in grouped by the top `pbits` hash bits;
// base[0..P] (device, g->part_meta + PT_PMAX) = first row of every partition
int32_t gbc_partition_scatter(dbhip_groupby* g, const GbCols& C, const GbcDesc& D, int64_t row0, int64_t cn, int pbits, hipStream_t s) {
  const int P = 1 << pbits;
  const int RW = gbc_row_words(D);
  int32_t rc;
  if ((rc = ensure((void**)&g->rows_in, &g->rows_in_cap, (size_t)cn * RW * 8 + 64))) return rc;
  // two 512-thread workgroups per CU while the cursors leave room for two staging areas (the phases of a batch — load, rank, stage,
  // barrier, copy out, barrier — of one workgroup overlap with the other's; r04d counters: 73 % of the wave cycles parked with one
  // 1024-thread workgroup per CU), one of 1024 threads beyond
  // (r04e: two 512-thread workgroups per CU instead of one of 1024 — 1024 row ranges instead of 512 — were SLOWER: 0.41 vs 0.38 ms at
  // 16 partitions, 0.69 vs 0.55 ms at 256: a workgroup's run inside a partition gets half as long)
  static const int gbc_t = exp_env("DBHIP_GBC_T") ? atoi(exp_env("DBHIP_GBC_T")) : GBC_T;
  const int T = (RW > 8 && P > 1024) ? 512 : gbc_t;   // (rows of 9 ... 12 words beside 16 K cursors: 512 staged rows fit the LDS)
  int64_t nwg = ceil_div(cn, (int64_t)T * 16);
  if (nwg > 512) nwg = 512;
  const int64_t rows_per_wg = ceil_div(cn, nwg);
  nwg = ceil_div(cn, rows_per_wg);
  if ((rc = ensure((void**)&g->part_meta, &g->part_meta_cap, ((size_t)(3 * PT_PMAX + 8) + (size_t)nwg * P) * 4))) return rc;
  uint32_t* tot = g->part_meta;
  uint32_t* base = g->part_meta + PT_PMAX;
  uint32_t* mat = g->part_meta + 3 * PT_PMAX + 8;
  // (dynamic LDS beyond 64 KB has to be asked for once per kernel)
  static std::once_flag raised_once;
  static hipError_t raised_err = hipSuccess;
  std::call_once(raised_once, [] {
    auto raise = [](const void* f, int bytes) { if (raised_err == hipSuccess) raised_err = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, bytes); };
#define GBC_RAISE(KW_, NV_)                                                        \
    raise((const void*)gbc_scatter_direct_kernel<KW_, NV_>, 150 * 1024);           \
    raise((const void*)gbc_scatter_kernel<KW_, NV_>, 150 * 1024);                  \
    raise((const void*)gbc_hist_kernel<KW_>, 64 * 1024);
    GBC_FOR_ALL(GBC_RAISE)
#undef GBC_RAISE
  });
  DBHIP_CHECK(raised_err);
  const int SR = RW <= 2 ? 4 : (RW <= 4 ? 2 : 1);
  // up to 1024 partitions: no histogram pass — fixed regions (the uniform share + 5 % + 16 K rows) and one global atomic per
  // (batch, partition); a region that overflows is found after the chunk's first read-back and the chunk redone the exact way
  static const bool no_direct = exp_env("DBHIP_GBC_DIRECT") && atoi(exp_env("DBHIP_GBC_DIRECT")) == 0;
  g->gbc_part_cap = 0;
  if (P <= 1024 && !g->gbc_nodirect && !no_direct && cn < ((int64_t)1 << 31)) {
    // a partition's share of the rows follows its share of the GROUPS: with G groups spread over P partitions a partition holds
    // G / P +- sqrt(G / P) of them (10^4 groups, 16 partitions: +-4 % — r04h: a flat 5 % of slack overflowed there); five sigma + 5 %
    int64_t est = estimate_groups(g->count_host > 0 ? g->count_host : 1, g->rows_seen > 0 ? g->rows_seen : 1);
    if (est < g->count_host) est = g->count_host;
    double per_part = (double)est / P;
    if (per_part < 1.0) per_part = 1.0;
    double slack = 0.05 + 5.0 / sqrt(per_part);
    if (slack > 1.0) slack = 1.0;
    const int64_t cap = cn / P + (int64_t)((double)(cn / P) * slack) + 16384;
    if ((rc = ensure((void**)&g->rows_in, &g->rows_in_cap, (size_t)cap * P * RW * 8 + 64))) return rc;
    DBHIP_CHECK(hipMemsetAsync(tot, 0, (size_t)P * 4, s));
    const size_t lds_d = (size_t)2 * P * 4 + (size_t)T * SR * 4 + (size_t)T * SR * RW * 8;
#define GBC_SCATTER_D(KW_, NV_) hipLaunchKernelGGL((gbc_scatter_direct_kernel<KW_, NV_>), dim3((int)nwg), dim3(T), lds_d, s, D, C, row0, cn, pbits, rows_per_wg, (uint32_t)cap, tot, g->rows_in, g->ctrl)
    GBC_DISPATCH(D, GBC_SCATTER_D);
#undef GBC_SCATTER_D
    DBHIP_LAUNCH_CHECK();
    g->gbc_part_cap = (uint32_t)cap;
    return DBHIP_OK;
  }
  switch (D.kw) {
    case 1: hipLaunchKernelGGL(gbc_hist_kernel<1>, dim3((int)nwg), dim3(T), (size_t)P * 4, s, D, C, row0, cn, pbits, rows_per_wg, mat); break;
    case 2: hipLaunchKernelGGL(gbc_hist_kernel<2>, dim3((int)nwg), dim3(T), (size_t)P * 4, s, D, C, row0, cn, pbits, rows_per_wg, mat); break;
    case 3: hipLaunchKernelGGL(gbc_hist_kernel<3>, dim3((int)nwg), dim3(T), (size_t)P * 4, s, D, C, row0, cn, pbits, rows_per_wg, mat); break;
    default: hipLaunchKernelGGL(gbc_hist_kernel<4>, dim3((int)nwg), dim3(T), (size_t)P * 4, s, D, C, row0, cn, pbits, rows_per_wg, mat); break;
  }
  hipLaunchKernelGGL((gb_part_colscan_kernel<false>), dim3((P + 63) / 64), dim3(256), 0, s, mat, P, (int)nwg, tot, base);
  hipLaunchKernelGGL(gb_part_scan_kernel, dim3(1), dim3(1024), 0, s, tot, P, base);
  hipLaunchKernelGGL((gb_part_colscan_kernel<true>), dim3((P + 63) / 64), dim3(256), 0, s, mat, P, (int)nwg, tot, base);
  const size_t lds = (size_t)P * 4 + (size_t)T * SR * 4 + (size_t)T * SR * RW * 8;
  if (lds > 150 * 1024) { set_error("groupby: compact scatter needs %zu bytes of LDS", lds); return DBHIP_ERR_INVALID; }
#define GBC_SCATTER(KW_, NV_) hipLaunchKernelGGL((gbc_scatter_kernel<KW_, NV_>), dim3((int)nwg), dim3(T), lds, s, D, C, row0, cn, pbits, rows_per_wg, mat, g->rows_in)
  GBC_DISPATCH(D, GBC_SCATTER);
#undef GBC_SCATTER
  DBHIP_LAUNCH_CHECK();
  return DBHIP_OK;
}

void part_geometry(const GbLayout& L, int* lcap, int* sw, size_t* lds_bytes) {
  *sw = L.W | 1;
  int c = 0;
  if ((size_t)256 * (*sw + 1) * 8 <= 64 * 1024) {
    c = 256;
    while ((size_t)(c * 2) * (*sw + 1) * 8 <= 64 * 1024) c *= 2;
  }
  *lcap = c;
  *lds_bytes = (size_t)c * (*sw + 1) * 8;
}

// The compact kernels' partition tables: up to 2048 slots (56 KB for key + sum + count: two 512-thread workgroups per CU), i.e.
// a quarter of the partitions the generic kernels' 1024-slot tables of 48-byte rows ask for — the scatter gets cheaper with every
// halving of the partition count (longer runs per workgroup and partition).
int gbc_part_threads(int lcap) { return lcap >= 4096 ? 1024 : (lcap >= 2048 ? 512 : 256); }
int gbc_part_lcap(const GbLayout& L, int lcap_max) {
  static const int env_c = exp_env("DBHIP_GBC_PARTLCAP") ? atoi(exp_env("DBHIP_GBC_PARTLCAP")) : 0;   // (experiments)
  const int max_c = env_c ? env_c : (lcap_max ? lcap_max : 2048);
  const size_t slot = (size_t)(L.nkey_words + (L.W - L.agg_off[0])) * 8 + 4;
  const size_t qrow = 48;   // (a deferred-row queue per wave: rows of up to 6 words, or the positions of wider rows)
  // two workgroups per CU (75 KB each) up to 2048 slots, one (150 KB) for 4096
  int c = 256;
  while (c < max_c) {
    const int n = c * 2;
    const size_t bytes = (size_t)n * slot + (size_t)(gbc_part_threads(n) / 64) * GBC_QCAP * qrow;
    if (bytes > (n >= 4096 ? (size_t)150 : (size_t)75) * 1024) break;
    c = n;
  }
  return c;
}
void table_geometry(const dbhip_groupby* g, int* lcap, int* sw, size_t* lds_bytes) {
  part_geometry(g->L, lcap, sw, lds_bytes);
  if (g->gbc_active) {
    *lcap = gbc_part_lcap(g->L, g->gbc_part_lcap_max);
    *lds_bytes = (size_t)*lcap * ((size_t)(g->L.nkey_words + (g->L.W - g->L.agg_off[0])) * 8 + 4);
  }
}

// One chunk [row0, row0 + cn) through hist -> scan -> scatter -> aggregate -> merge.
// *spilled = rows that did not fit their partition's LDS table (went through the row path).
// hist -> scan -> scatter: rows [row0, row0 + cn) serialized into g->rows_in grouped by the top `pbits` hash bits;
// base[0..P] (device, g->part_meta + PT_PMAX) = first row of every partition
int32_t partition_scatter(dbhip_groupby* g, const GbCols& C, int64_t row0, int64_t cn, int pbits, hipStream_t s) {
  const GbLayout& L = g->L;
  const int P = 1 << pbits;
  int32_t rc;
  if ((rc = ensure((void**)&g->rows_in, &g->rows_in_cap, (size_t)cn * L.W * 8))) return rc;
  // workgroups of the histogram / scatter pair: contiguous row ranges, 1024 threads (r02x tried 256-thread workgroups for few
  // partitions — several per CU to overlap their phases: scatter 0.90 ms against 0.78 ms per 60 M rows, and a 4 x taller
  // histogram matrix to scan)
  const int T = PT_THREADS;
  const int64_t nwg_max = T == 256 ? 2048 : 512;
  int64_t nwg = ceil_div(cn, (int64_t)T * 16);
  if (nwg > nwg_max) nwg = nwg_max;
  const int64_t rows_per_wg = ceil_div(cn, nwg);
  nwg = ceil_div(cn, rows_per_wg);
  if ((rc = ensure((void**)&g->part_meta, &g->part_meta_cap, ((size_t)(3 * PT_PMAX + 8) + (size_t)nwg * P) * 4))) return rc;
  uint32_t* tot = g->part_meta;
  uint32_t* base = g->part_meta + PT_PMAX;
  uint32_t* mat = g->part_meta + 3 * PT_PMAX + 8;
  hipLaunchKernelGGL(gb_part_hist_kernel, dim3((int)nwg), dim3(T), (size_t)P * 4, s, L, C, row0, cn, pbits,
                     rows_per_wg, mat, g->ctrl);
  hipLaunchKernelGGL((gb_part_colscan_kernel<false>), dim3((P + 63) / 64), dim3(256), 0, s, mat, P, (int)nwg, tot, base);
  hipLaunchKernelGGL(gb_part_scan_kernel, dim3(1), dim3(1024), 0, s, tot, P, base);
  hipLaunchKernelGGL((gb_part_colscan_kernel<true>), dim3((P + 63) / 64), dim3(256), 0, s, mat, P, (int)nwg, tot, base);
  // staged copy-out while a batch of rows (2 or 1 per thread) fits the LDS beside the cursors; else lanes store their rows themselves
  static const bool no_stage = exp_env("DBHIP_GB_NOSTAGE") != nullptr;
  const size_t row_bytes = 4 + (size_t)(L.W | 1) * 8;
  const size_t lds2 = (size_t)P * 4 + (size_t)T * 2 * row_bytes, lds1 = (size_t)P * 4 + (size_t)T * row_bytes;
  const size_t lds_max = 144 * 1024;
  static std::once_flag raised_once;   // (dynamic LDS beyond 64 KB has to be asked for once per kernel)
  static hipError_t raised_err = hipSuccess;
  std::call_once(raised_once, [] {
    raised_err = hipFuncSetAttribute((const void*)gb_part_scatter_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024);
    if (raised_err == hipSuccess) raised_err = hipFuncSetAttribute((const void*)gb_part_scatter_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024);
  });
  DBHIP_CHECK(raised_err);
  if (!no_stage && lds2 <= lds_max)
    hipLaunchKernelGGL((gb_part_scatter_kernel<2>), dim3((int)nwg), dim3(T), lds2, s, L, C, row0, cn, pbits, rows_per_wg, mat,
                       g->rows_in, g->ctrl);
  else if (!no_stage && lds1 <= lds_max)
    hipLaunchKernelGGL((gb_part_scatter_kernel<1>), dim3((int)nwg), dim3(T), lds1, s, L, C, row0, cn, pbits, rows_per_wg, mat,
                       g->rows_in, g->ctrl);
  else
    hipLaunchKernelGGL((gb_part_scatter_kernel<0>), dim3((int)nwg), dim3(T), (size_t)P * 4, s, L, C, row0, cn, pbits,
                       rows_per_wg, mat, g->rows_in, g->ctrl);
  DBHIP_LAUNCH_CHECK();
  return DBHIP_OK;
}

int32_t add_chunk_partitioned(dbhip_groupby* g, const GbCols& C, int64_t row0, int64_t cn, hipStream_t s,
                              int64_t* spilled) {
  const GbLayout& L = g->L;
  const int pbits = g->part_bits;
  const int P = 1 << pbits;
  int lcap, sw;
  size_t lds_bytes;
  int32_t rc;
  // compact rows (gb_compact.h) when the layout and the columns qualify; the direct-insert mode consumes serialized rows
  GbcDesc D;
  const bool gbc = g->gbc_active && !g->gbc_skip && !g->part_direct && gbc_enabled(g) && gbc_describe(L, C, &D);
  D.ctrl = g->ctrl;
  if (!gbc) g->gbc_active = 0;   // (the geometry of everything that follows is the generic kernels')
  table_geometry(g, &lcap, &sw, &lds_bytes);
  if (gbc) rc = gbc_partition_scatter(g, C, D, row0, cn, pbits, s);
  else rc = partition_scatter(g, C, row0, cn, pbits, s);
  if (rc) return rc;
  if (!gbc && (rc = ensure((void**)&g->spill_idx, &g->spill_idx_cap, (size_t)cn * 4))) return rc;
  uint32_t* base = g->part_meta + PT_PMAX;
  if (g->part_direct && g->hash_mask == ~0ULL) {
    // room: every row of the chunk may be a new group, but a table for 64 M new groups that then holds 10 M is a waste the
    // flush pays for — size for the groups the rows seen so far predict (at least twice the chunk's share of them), and let
    // a slice that runs full hand its rows to the row path, which grows the table for good
    int64_t expect = cn;
    if (g->rows_seen >= (4 << 20)) {
      const int64_t est = estimate_groups(g->count_host, g->rows_seen);
      const int64_t more = est > g->count_host ? est - g->count_host : 0;
      if (more * 2 + (1 << 20) < expect) expect = more * 2 + (1 << 20);
    }
    while ((g->count_host + expect) * 135 > g->cap * 100 || g->cap < (int64_t)P * 64)
      if ((rc = grow(g, s))) return rc;
    DBHIP_CHECK(hipMemsetAsync(&g->ctrl[5], 0, 16, s));
    PiArgs I;
    I.rows = g->rows_in; I.base = base; I.pbits = pbits; I.slot_hash = g->slot_hash; I.table = g->rows; I.cap = g->cap;
    I.hash_mask = g->hash_mask; I.spill_idx = g->spill_idx; I.ctrl = g->ctrl;
    hipLaunchKernelGGL(gb_part_insert_kernel, dim3(P), dim3(256), 0, s, L, I);
    DBHIP_LAUNCH_CHECK();
    uint64_t* hc = pinned_words(0);
    if (!hc) return DBHIP_ERR_HIP;
    DBHIP_CHECK(hipMemcpyAsync(hc, g->ctrl, 8 * sizeof(uint64_t), hipMemcpyDeviceToHost, s));
    DBHIP_CHECK(hipStreamSynchronize(s));
    if (hc[3] & 2) {
      DBHIP_CHECK(hipMemsetAsync(&g->ctrl[3], 0, 8, s));
      g->has_long = 1; g->fast_disabled = 1;
      *spilled = -1;
      return DBHIP_OK;
    }
    g->count_host = (int64_t)hc[0];
    const int64_t nlist = (int64_t)hc[6];
    if (nlist > 0) {
      if ((rc = ensure((void**)&g->spill_rows, &g->spill_rows_cap, (size_t)nlist * L.W * 8))) return rc;
      hipLaunchKernelGGL(gb_gather_rows_kernel, dim3(grid_for(nlist, 256)), dim3(256), 0, s, g->rows_in, g->spill_idx,
                         nlist, L.W, g->spill_rows);
      DBHIP_LAUNCH_CHECK();
      if ((rc = merge_rows(g, g->spill_rows, nlist, s))) return rc;
    }
    if (getenv("DBHIP_TRACE")) fprintf(stderr, "[dbhip] groupby partitioned insert: rows=%lld listed=%lld groups=%lld cap=%lld\n",
                                       (long long)cn, (long long)nlist, (long long)g->count_host, (long long)g->cap);
    *spilled = 0;   // (listed rows are no sign of a partitioning that is too coarse)
    return DBHIP_OK;
  }
  // the LDS table no larger than the partition's groups ask for (4 x the expected number, >= 256 slots): 12 KB instead of
  // 48 KB lets 12 workgroups instead of 3 share a CU, and the tile loop is a chain of load -> probe -> barrier -> merge
  {
    const int64_t est = estimate_groups(g->count_host > 0 ? g->count_host : 1, g->rows_seen > 0 ? g->rows_seen : 1);
    const int64_t groups = est > g->count_host ? est : g->count_host;
    const int64_t per_part = groups / P + 8;
    while (lcap > 256 && (int64_t)(lcap / 2) >= 4 * per_part) lcap /= 2;
    lds_bytes = gbc ? gbc_agg_lds_bytes(D, lcap, lcap >= 2048 ? 512 : 256) : (size_t)lcap * (sw + 1) * 8;
  }
  // workgroups per partition: fill the chip (>= ~1024 workgroups) without making splits tiny
  int splits = 1;
  const int want_wgs = gbc ? 512 : 1024;   // (every workgroup hands on a partial row per group it met: half the workgroups, half the rows to merge)
  while (P * splits < want_wgs && cn / ((int64_t)P * splits * 2) >= 4096) splits *= 2;
  const int agrid = P * splits;
  if ((rc = ensure((void**)&g->partial, &g->partial_cap, (size_t)agrid * lcap * L.W * 8))) return rc;
  // one workgroup per partition and a table at least as fine as the partitioning: the partial rows stay per partition and
  // are merged by the partition's own workgroup (gb_part_merge_kernel); otherwise one packed list for the row path
  static const bool no_excl = exp_env("DBHIP_GB_NOEXCL") != nullptr;
  const bool exclusive = splits == 1 && !no_excl && g->hash_mask == ~0ULL;
  uint32_t* pcount = g->part_meta + 2 * PT_PMAX + 8;
  if (exclusive) DBHIP_CHECK(hipMemsetAsync(pcount, 0, (size_t)P * 4, s));
  DBHIP_CHECK(hipMemsetAsync(&g->ctrl[5], 0, 24, s));   // [5] partial rows, [6] spilled rows, [7] rows of the packed list of heavy partitions
  PaArgs A;
  A.rows = g->rows_in; A.base = base; A.splits = splits; A.lcap = lcap; A.sw = sw;
  A.llimit = (uint32_t)(lcap - lcap / 4);
  A.hash_mask = g->hash_mask; A.partial = g->partial; A.spill_idx = g->spill_idx; A.ctrl = g->ctrl;
  A.pcount = exclusive ? pcount : nullptr;
  const int64_t gbc_spill_cap = cn / 8 + 65536;
  if (gbc) {
    if ((rc = ensure((void**)&g->gbc_spill, &g->gbc_spill_cap, (size_t)gbc_spill_cap * L.W * 8))) return rc;
    GbcAggArgs G;
    memset(&G, 0, sizeof(G));
    G.rows = g->rows_in; G.base = base; G.splits = splits; G.lcap = lcap; G.llimit = A.llimit; G.partial = g->partial;
    if (g->gbc_part_cap) { G.pcursor = g->part_meta; G.part_cap = g->gbc_part_cap; }   // (the direct scatter's cursors: part_meta[0..P))
    G.pcount = A.pcount; G.spill = g->gbc_spill; G.spill_cap = (uint64_t)gbc_spill_cap; G.ctrl = g->ctrl;
    // HEAVY partitions (one key with a large share of the rows — NULLs, a default value — lands in ONE partition, and with one
    // workgroup per sub-range that workgroup is the whole kernel's tail: r05, 25 % NULL keys: 8.6 ms at 2 x 10^4 groups, 128 ms at
    // 10^6 where the partition has one workgroup): a partition longer than twice the average sub-range gets more sub-ranges, worked
    // on by EXTRA workgroups behind the regular P x splits (at most cn / max_rows of them; those not needed leave at once). Their
    // partial rows go to a packed list behind the per-partition lists and through the row path.
    static const bool no_heavy = exp_env("DBHIP_GBC_HEAVY") && atoi(exp_env("DBHIP_GBC_HEAVY")) == 0;
    int extra_max = 0;
    if (!no_heavy) {
      int64_t max_rows = 2 * (cn / agrid);
      if (max_rows < 32768) max_rows = 32768;
      extra_max = (int)(cn / max_rows) + 1;
      if ((rc = ensure((void**)&g->gbc_split, &g->gbc_split_cap, ((size_t)P + 2 + (size_t)extra_max) * 4))) return rc;
      if ((rc = ensure((void**)&g->partial, &g->partial_cap, ((size_t)agrid + 2 * (size_t)extra_max) * lcap * L.W * 8))) return rc;
      G.partial = g->partial;
      G.nsp = g->gbc_split; G.extra_n = g->gbc_split + P + 1; G.extra_map = g->gbc_split + P + 2;
      G.nparts = P; G.packed_base = (uint64_t)agrid * lcap;
      hipLaunchKernelGGL(gbc_split_map_kernel, dim3(1), dim3(1024), 0, s, G.pcursor, G.part_cap, base, P, splits, (uint32_t)max_rows, (uint32_t)extra_max, g->gbc_split);
    }
    static const int agg_t = exp_env("DBHIP_GBC_AGGT") ? atoi(exp_env("DBHIP_GBC_AGGT")) : 0;   // (experiments)
    const int threads = agg_t ? agg_t : gbc_part_threads(lcap);
    lds_bytes = gbc_agg_lds_bytes(D, lcap, threads);
    static std::once_flag agg_raised_once;
    static hipError_t agg_raised_err = hipSuccess;
    std::call_once(agg_raised_once, [] {
#define GBC_RAISE(KW_, NV_) if (agg_raised_err == hipSuccess) agg_raised_err = hipFuncSetAttribute((const void*)gbc_agg_kernel<KW_, NV_, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
      GBC_FOR_ALL(GBC_RAISE)
#undef GBC_RAISE
    });
    DBHIP_CHECK(agg_raised_err);
#define GBC_AGG(KW_, NV_) hipLaunchKernelGGL((gbc_agg_kernel<KW_, NV_, false>), dim3(agrid + extra_max), dim3(threads), lds_bytes, s, D, C, G)
    GBC_DISPATCH(D, GBC_AGG);
#undef GBC_AGG
  } else {
    // heavy partitions: the same split as for the compact kernels (the generic partitions are exact: base[], from the histogram pass)
    static const bool no_heavy = exp_env("DBHIP_GBC_HEAVY") && atoi(exp_env("DBHIP_GBC_HEAVY")) == 0;
    int extra_max = 0;
    A.nsp = nullptr; A.extra_n = nullptr; A.extra_map = nullptr; A.nparts = P; A.packed_base = (uint64_t)agrid * lcap;
    if (!no_heavy) {
      int64_t max_rows = 2 * (cn / agrid);
      if (max_rows < 32768) max_rows = 32768;
      extra_max = (int)(cn / max_rows) + 1;
      if ((rc = ensure((void**)&g->gbc_split, &g->gbc_split_cap, ((size_t)P + 2 + (size_t)extra_max) * 4))) return rc;
      if ((rc = ensure((void**)&g->partial, &g->partial_cap, ((size_t)agrid + 2 * (size_t)extra_max) * lcap * L.W * 8))) return rc;
      A.partial = g->partial;
      A.nsp = g->gbc_split; A.extra_n = g->gbc_split + P + 1; A.extra_map = g->gbc_split + P + 2;
      hipLaunchKernelGGL(gbc_split_map_kernel, dim3(1), dim3(1024), 0, s, (const uint32_t*)nullptr, 0u, base, P, splits, (uint32_t)max_rows, (uint32_t)extra_max, g->gbc_split);
    }
    if (L.W <= 8) hipLaunchKernelGGL((gb_part_agg_kernel<8>), dim3(agrid + extra_max), dim3(256), lds_bytes, s, L, A);
    else hipLaunchKernelGGL((gb_part_agg_kernel<0>), dim3(agrid + extra_max), dim3(256), lds_bytes, s, L, A);
  }
  DBHIP_LAUNCH_CHECK();
  uint64_t* hc = pinned_words(0);
  if (!hc) return DBHIP_ERR_HIP;
  DBHIP_CHECK(hipMemcpyAsync(hc, g->ctrl, 8 * sizeof(uint64_t), hipMemcpyDeviceToHost, s));
  DBHIP_CHECK(hipStreamSynchronize(s));
  if (hc[3] & 2) {  // a long string key: this chunk goes to the row path (nothing was merged yet), see add_block_fast
    DBHIP_CHECK(hipMemsetAsync(&g->ctrl[3], 0, 8, s));
    g->has_long = 1; g->fast_disabled = 1;
    *spilled = -1;
    return DBHIP_OK;
  }
  if (gbc && (hc[3] & 8)) {
    // a partition outgrew the fixed region of the histogram-less scatter (heavy keys): nothing of this chunk has been merged —
    // redo it with the exact histogram
    DBHIP_CHECK(hipMemsetAsync(&g->ctrl[3], 0, 8, s));
    g->gbc_nodirect = 1;
    if (getenv("DBHIP_TRACE")) fprintf(stderr, "[dbhip] groupby: a partition outgrew its region, chunk redone with the histogram pass\n");
    return add_chunk_partitioned(g, C, row0, cn, s, spilled);
  }
  if (gbc && (hc[3] & 4)) {
    // more rows than the compact kernels' spill buffer holds met full tables (the estimate behind the partitioning was far off):
    // nothing of this chunk has been merged — redo it with the generic kernels, whose spill list covers every row
    DBHIP_CHECK(hipMemsetAsync(&g->ctrl[3], 0, 8, s));
    g->gbc_skip = 1;   // (this chunk only: the partitioning is widened by the caller on what the redone chunk reports)
    rc = add_chunk_partitioned(g, C, row0, cn, s, spilled);
    g->gbc_skip = 0;
    g->gbc_active = 1;
    return rc;
  }
  const int64_t nspill = (int64_t)hc[6];
  const int64_t npartial = (int64_t)hc[5];
  const int64_t npacked = exclusive ? (int64_t)hc[7] : 0;   // (partial rows of heavy partitions' sub-ranges, behind the per-partition lists)
  int64_t nlisted = 0;
  if (exclusive && npartial > 0) {
    // every partial row may be a new group: make room first (the slices move with the capacity, the kernel takes it as it is)
    while ((g->count_host + npartial) * 135 > g->cap * 100 || g->cap < (int64_t)P * 64)
      if ((rc = grow(g, s))) return rc;
    if ((rc = ensure((void**)&g->retry, &g->retry_cap, (size_t)npartial * 4))) return rc;
    DBHIP_CHECK(hipMemsetAsync(&g->ctrl[2], 0, 8, s));
    PmArgs M;
    M.partial = g->partial; M.pcount = pcount; M.lcap = lcap; M.pbits = pbits;
    M.slot_hash = g->slot_hash; M.rows = g->rows; M.cap = g->cap; M.hash_mask = g->hash_mask;
    M.retry = g->retry; M.ctrl = g->ctrl;
    hipLaunchKernelGGL(gb_part_merge_kernel, dim3(P), dim3(256), (size_t)lcap * 4, s, L, M);
    DBHIP_LAUNCH_CHECK();
    DBHIP_CHECK(hipMemcpyAsync(hc, g->ctrl, 8 * sizeof(uint64_t), hipMemcpyDeviceToHost, s));
    DBHIP_CHECK(hipStreamSynchronize(s));
    g->count_host = (int64_t)hc[0];
    nlisted = (int64_t)hc[2];
  }
  const int64_t ngather = gbc ? 0 : nspill;   // (the compact kernels wrote their spilled rows in table layout already: g->gbc_spill)
  if (ngather + nlisted > 0) {
    // compact the listed rows BEFORE merge_rows may touch its own scratch (g->retry is part of it)
    if ((rc = ensure((void**)&g->spill_rows, &g->spill_rows_cap, (size_t)(ngather + nlisted) * L.W * 8))) return rc;
    if (ngather > 0)
      hipLaunchKernelGGL(gb_gather_rows_kernel, dim3(grid_for(ngather, 256)), dim3(256), 0, s, g->rows_in, g->spill_idx,
                         ngather, L.W, g->spill_rows);
    if (nlisted > 0)
      hipLaunchKernelGGL(gb_gather_rows_kernel, dim3(grid_for(nlisted, 256)), dim3(256), 0, s, g->partial, g->retry,
                         nlisted, L.W, g->spill_rows + (size_t)ngather * L.W);
    DBHIP_LAUNCH_CHECK();
  }
  if (!exclusive && (rc = merge_rows(g, g->partial, npartial, s))) return rc;
  if (ngather + nlisted > 0 && (rc = merge_rows(g, g->spill_rows, ngather + nlisted, s))) return rc;
  if (npacked > 0 && (rc = merge_rows(g, g->partial + (size_t)agrid * lcap * L.W, npacked, s))) return rc;
  if (gbc && nspill > 0 && (rc = merge_rows(g, g->gbc_spill, nspill, s))) return rc;
  if (getenv("DBHIP_TRACE")) fprintf(stderr, "[dbhip] groupby partitioned merge: exclusive=%d partial=%lld listed=%lld spilled=%lld cap=%lld\n",
                                     (int)exclusive, (long long)npartial, (long long)nlisted, (long long)nspill, (long long)g->cap);
  *spilled = nspill;
  return DBHIP_OK;
}

// Called once the LDS pre-aggregation (or the first row-path chunk) has shown that the key
// distribution does not fit one workgroup's table: `groups` distinct groups were seen in the first
// `rows_seen` rows. Chooses the partition count, or gives up (row path) for high cardinality.
// Number of distinct groups the WHOLE input is likely to hold, from `d` distinct groups met in the first `s` rows: for D equally
// likely groups E[d] = D (1 - exp(-s / D)); solved for D by bisection. (Skewed keys make this an under-estimate, which errs
// towards partitioning; a partitioning that turns out too narrow widens itself, partitioned_step.) A prefix that is all
// distinct tells nothing: returned as "huge".
int64_t estimate_groups(int64_t d, int64_t s) {
  if (s <= 0 || d <= 0) return d;
  const double r = (double)d / (double)s;
  if (r > 0.97) return INT64_MAX / 16;
  double lo = 1e-6, hi = 64.0;  // x = D / s
  for (int it = 0; it < 60; ++it) {
    const double x = 0.5 * (lo + hi);
    if (x * (1.0 - exp(-1.0 / x)) < r) lo = x; else hi = x;
  }
  const double D = 0.5 * (lo + hi) * (double)s;
  return D < (double)d ? d : (int64_t)D;
}

// Called once the LDS pre-aggregation (or the first row-path chunk) has shown that the key
// distribution does not fit one workgroup's table: `groups` distinct groups were seen in the first
// `rows_seen` rows; `n_block` = rows of the add_block call that is being worked on. Chooses the partition count, or gives up
// (row path) when fewer than ~8 rows per group are to be expected.
void decide_partitioning(dbhip_groupby* g, int64_t groups, int64_t rows_seen, int64_t n_block) {
  int lcap, sw;
  size_t lds_bytes;
  const int64_t est = estimate_groups(groups, rows_seen);
  // compact kernels: 4096-slot tables (one 1024-thread workgroup per CU) once the groups would otherwise ask for more than 1024
  // partitions — the scatter loses more with every doubling of the partition count than the aggregation gains from the second
  // workgroup per CU (r04 sweep, 10^6 groups: 1.69 ms against 2.11 ms; 10^4 / 10^5 groups: 2048 slots win, 1.15 / 1.33 against 1.29 / 1.45)
  // (only where that keeps the partitions at <= 1024, the histogram-less scatter: at 10^7 groups 16384 workgroups each setting up and
  // flushing a 112 KB table cost more than they save — 6.6 against 6.2 ms)
  g->gbc_part_lcap_max = (g->gbc_active && est > 500000 && est <= 1500000) ? 4096 : 0;
  table_geometry(g, &lcap, &sw, &lds_bytes);
  g->part_bits = -1;
  if (lcap == 0 || g->part_forbidden) return;
  const int64_t total = n_block > rows_seen ? n_block : rows_seen;
  const int64_t per_part = lcap * 3 / 8;  // target groups per partition: half of the LDS table's limit
  int bits = 4;
  while (bits < PT_MAX_BITS && ((int64_t)per_part << bits) < est) ++bits;
  g->part_chunk = 0;
  g->part_direct = 0;
  g->part_adapt = 0;
  if (((int64_t)per_part << bits) < est) {
    // more groups than the finest partitioning's LDS tables hold at once, or a probe that was (nearly) all distinct and
    // says nothing: finest partitioning, a 4 M-row chunk to learn from, then chunks sized by the estimate (adapt_chunk)
    static const bool no_adapt = exp_env("DBHIP_GB_NODIRECT") != nullptr;
    if (no_adapt) { g->part_bits = -1; return; }
    bits = PT_MAX_BITS;
    g->part_adapt = 1;
    g->part_chunk = 4 << 20;
  }
  g->part_bits = bits;
  // An estimate EXTRAPOLATED from a probe that met a new group in more than every fourth row is only as good as its assumption of
  // equally likely groups: one heavy key (25 % NULLs) made 10^6 groups look like 3 x 10^5 (r05), the partitioning came out four times
  // too coarse and 9 M of 60 M rows left the full tables for the row path (124 ms). Such an estimate is checked on a 4 M-row chunk
  // first; the partitioning of the rest follows what that chunk found (partitioned_step).
  static const bool no_validate = exp_env("DBHIP_GB_VALIDATE") && atoi(exp_env("DBHIP_GB_VALIDATE")) == 0;
  if (!g->part_adapt && !g->part_validated && !no_validate && groups * 4 > rows_seen && total - rows_seen > (16 << 20)) {
    g->part_validate = 1;
    // (1 M rows by default, DBHIP_GB_VALIDATE_ROWS: at 10^6 groups under a 25 % heavy key they put the estimate within 1.3 x, which the
    // tables' slack absorbs — a partition is sized for 3/8 of its table and spills at 3/4; 4 M rows cost the uniform 10^6 case 0.3 ms)
    static const int64_t vrows = [] { const char* e = exp_env("DBHIP_GB_VALIDATE_ROWS"); const long long v = e ? atoll(e) : 0; return (int64_t)(v >= (1 << 18) ? v : (1 << 20)); }();
    g->part_chunk = vrows;
  }
  if (getenv("DBHIP_TRACE")) fprintf(stderr, "[dbhip] groupby: %lld groups in the first %lld rows -> ~%lld groups in %lld rows, %d partition bits\n",
                                     (long long)groups, (long long)rows_seen, (long long)est, (long long)total, bits);
}

}  // namespace

// gbk_merge_lds.h — a fragment of k_groupby.hip (ONE translation unit: the kernels share the anonymous namespace's helpers and the table struct;
// split by kernel family in round 6, VERDICT r05 hygiene #18): table memory and the string arena, merge_rows (the row path's host side), and the LDS pre-aggregation path of add_block.
// Included by k_groupby.hip only, in this order: gbk_rows.h, gbk_merge_lds.h, gbk_partitioned.h, gbk_api.h.

namespace {

// Per-table scratch and the table arrays come from the library's block cache (dbhip_alloc / dbhip_free: freed blocks of
// >= 1 MiB are kept in size-class lists), so a plan that creates a table per block does not pay hipMalloc / hipFree of
// GB-sized buffers per call (a fresh 9 GB hipMalloc costs tens of milliseconds).
int32_t ensure(void** p, size_t* cap, size_t bytes) {
  if (*cap >= bytes) return DBHIP_OK;
  if (*p) {
    int32_t rc = dbhip_free(*p);  // synchronises the device before the block may be re-used
    if (rc) return rc;
    *p = nullptr; *cap = 0;
  }
  size_t want = bytes + (bytes >> 3) + 256;
  int32_t rc = dbhip_alloc(want, p);
  if (rc) { *p = nullptr; return rc; }
  *cap = want;
  return DBHIP_OK;
}

// all or nothing: on failure the table keeps its old arrays and capacity (grow() and create() rely on that)
int32_t alloc_table(dbhip_groupby* g, int64_t cap, hipStream_t s) {
  uint64_t* nh = nullptr;
  uint64_t* nr = nullptr;
  int32_t rc = dbhip_alloc((size_t)cap * 8, (void**)&nh);
  if (rc == DBHIP_OK) rc = dbhip_alloc((size_t)cap * g->L.W * 8, (void**)&nr);
  hipError_t e = rc == DBHIP_OK ? hipMemsetAsync(nh, 0, (size_t)cap * 8, s) : hipSuccess;
  if (rc != DBHIP_OK || e != hipSuccess) {
    if (nh) (void)dbhip_free(nh);
    if (nr) (void)dbhip_free(nr);
    return rc != DBHIP_OK ? rc : hip_fail(e, "groupby: allocating the table");
  }
  g->slot_hash = nh;
  g->rows = nr;
  g->cap = cap;
  return DBHIP_OK;
}

int32_t grow(dbhip_groupby* g, hipStream_t s) {
  uint64_t* old_hash = g->slot_hash;
  uint64_t* old_rows = g->rows;
  int64_t old_cap = g->cap;
  int32_t rc = alloc_table(g, old_cap * 4, s);
  if (rc) return rc;
  hipLaunchKernelGGL(gb_rehash_kernel, dim3(grid_for(old_cap, 256)), dim3(256), 0, s, g->L, old_hash,
                     old_rows, old_cap, g->slot_hash, g->rows, g->cap, g->hash_mask);
  DBHIP_LAUNCH_CHECK();
  DBHIP_CHECK(hipStreamSynchronize(s));
  int32_t r1 = dbhip_free(old_hash), r2 = dbhip_free(old_rows);
  return r1 ? r1 : r2;
}

// ---- min / max over String: the winners' bytes move into the table's arena before a call returns (gb_device.h) ----
bool layout_has_str_minmax(const GbLayout& L) {
  for (int a = 0; a < L.naggs; ++a) if (gb_minmax_str(L, a)) return true;
  return false;
}
int32_t refuse_str_minmax_state(const GbLayout& L, const char* fn) {
  (void)L; (void)fn;   // (round 5: String min / max, Decimal256 sums, min / max and keys all have their state-block form)
  return DBHIP_OK;
}
// mode 0: sum the (8-byte rounded) sizes of the long values whose bytes lie outside [lo, hi) into *acc;
// mode 1: copy them into the arena (bump cursor ctrl[8]) and point the state at the copy;
// mode 2: the arena moved from [lo, hi) by `delta`: states that point into the old range follow it.
// mode 0 / 1: count / copy the long min / max String winners that still lie OUTSIDE the arena [lo, hi); mode 2: the arena moved by
// `delta`; mode 3: count the LIVE bytes of the arena — long keys and the winners inside it — into acc; mode 4: move the live bytes
// from the old arena at `lo` into the new one at `arena` (cursor ctrl[8], zeroed by the host) and rewrite key offsets / winner addresses
__global__ __launch_bounds__(256) void gb_pin_strings_kernel(GbLayout L, const uint64_t* __restrict__ slot_hash, uint64_t* __restrict__ rows,
                                                             int64_t cap, uint64_t lo, uint64_t hi, int mode, uint8_t* arena, int64_t delta,
                                                             uint64_t* ctrl, unsigned long long* acc) {
  uint64_t mine = 0;
  for (int64_t sl = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; sl < cap; sl += (int64_t)gridDim.x * blockDim.x) {
    if (slot_hash[sl] == 0) continue;
    if (mode >= 3) {
      uint64_t* r = rows + sl * L.W;
      for (int k = 1; k < L.nkey_words; ++k) {
        if (!((L.str_w1_mask >> k) & 1)) continue;
        const uint32_t len = (uint32_t)r[k - 1];
        if (len <= 12) continue;
        const uint64_t room = ((uint64_t)len + 7) & ~7ULL;
        if (mode == 3) { mine += room; continue; }
        const uint64_t off = atomicAdd((unsigned long long*)&ctrl[8], (unsigned long long)room);
        const uint8_t* src = (const uint8_t*)lo + r[k];
        for (uint32_t i = 0; i < len; ++i) arena[off + i] = src[i];
        r[k] = off;
      }
    }
    for (int a = 0; a < L.naggs; ++a) {
      if (!gb_minmax_str(L, a)) continue;
      uint64_t* st = rows + sl * L.W + L.agg_off[a];
      const uint32_t len = (uint32_t)st[0];
      if (!st[1] || len <= 12) continue;
      const bool inside = st[2] >= lo && st[2] < hi;
      if (mode == 2) { if (inside) st[2] = (uint64_t)((int64_t)st[2] + delta); continue; }
      const uint64_t room = ((uint64_t)len + 7) & ~7ULL;
      if (mode == 3) { if (inside) mine += room; continue; }
      if (mode == 4 ? !inside : inside) continue;
      if (mode == 0) { mine += room; continue; }
      const uint64_t off = atomicAdd((unsigned long long*)&ctrl[8], (unsigned long long)room);
      const uint8_t* src = (const uint8_t*)st[2];
      for (uint32_t i = 0; i < len; ++i) arena[off + i] = src[i];
      st[2] = (uint64_t)(arena + off);
    }
  }
  if (mode == 0 || mode == 3) {
    mine = wave_sum_u64(mine);
    if (mine && lane_id() == 0) atomicAdd(acc, (unsigned long long)mine);
  }
}

// room for `extra` more bytes of long string keys (ctrl[8] = bytes in use); offsets into the arena stay valid when it moves
int32_t reserve_arena(dbhip_groupby* g, uint64_t extra, hipStream_t s) {
  if (extra == 0) return DBHIP_OK;
  uint64_t used = 0;
  DBHIP_CHECK(hipMemcpyAsync(&used, &g->ctrl[8], 8, hipMemcpyDeviceToHost, s));
  DBHIP_CHECK(hipStreamSynchronize(s));
  const size_t need = (size_t)(used + extra);
  if (need <= g->arena_cap) return DBHIP_OK;
  size_t want = g->arena_cap ? g->arena_cap * 2 : ((size_t)1 << 20);
  while (want < need) want *= 2;
  uint8_t* na = nullptr;
  int32_t rc = dbhip_alloc(want, (void**)&na);
  if (rc) return rc;
  if (used) DBHIP_CHECK(hipMemcpyAsync(na, g->arena, (size_t)used, hipMemcpyDeviceToDevice, s));
  if (g->arena && layout_has_str_minmax(g->L))   // min / max String states hold ADDRESSES into the arena: they follow it
    hipLaunchKernelGGL(gb_pin_strings_kernel, dim3(grid_for(g->cap, 256)), dim3(256), 0, s, g->L, g->slot_hash, g->rows, g->cap,
                       (uint64_t)g->arena, (uint64_t)g->arena + g->arena_cap, 2, na, (int64_t)((intptr_t)na - (intptr_t)g->arena), g->ctrl,
                       (unsigned long long*)nullptr);
  DBHIP_CHECK(hipStreamSynchronize(s));
  if (g->arena) (void)dbhip_free(g->arena);
  g->arena = na;
  g->arena_cap = want;
  return DBHIP_OK;
}
// min / max over String: after rows were merged, every long winner whose bytes still lie in a caller's buffer (or another table's
// arena) is copied into this table's arena. One counting pass, the reservation, one copying pass — per merge_rows call.
int32_t pin_string_states(dbhip_groupby* g, hipStream_t s) {
  if (!layout_has_str_minmax(g->L)) return DBHIP_OK;
  unsigned long long* acc = (unsigned long long*)&g->ctrl[10];
  DBHIP_CHECK(hipMemsetAsync(acc, 0, 8, s));
  const int grid = grid_for(g->cap, 256);
  hipLaunchKernelGGL(gb_pin_strings_kernel, dim3(grid), dim3(256), 0, s, g->L, g->slot_hash, g->rows, g->cap, (uint64_t)g->arena,
                     (uint64_t)g->arena + g->arena_cap, 0, g->arena, (int64_t)0, g->ctrl, acc);
  DBHIP_LAUNCH_CHECK();
  uint64_t bytes = 0;
  DBHIP_CHECK(hipMemcpyAsync(&bytes, acc, 8, hipMemcpyDeviceToHost, s));
  DBHIP_CHECK(hipStreamSynchronize(s));
  if (bytes == 0) return DBHIP_OK;
  int32_t rc = reserve_arena(g, bytes, s);
  if (rc) return rc;
  hipLaunchKernelGGL(gb_pin_strings_kernel, dim3(grid), dim3(256), 0, s, g->L, g->slot_hash, g->rows, g->cap, (uint64_t)g->arena,
                     (uint64_t)g->arena + g->arena_cap, 1, g->arena, (int64_t)0, g->ctrl, acc);
  DBHIP_LAUNCH_CHECK();
  // Displaced winners stay behind in the arena (max() over an ascending column pins a new string per group and block): once as
  // many bytes were pinned as the arena held live at the last look (at least 1 MiB), the live bytes are counted, and an arena more
  // than twice that size is rebuilt from the current keys and winners.
  g->arena_pinned += bytes;
  if (g->arena_pinned < (g->arena_live > ((uint64_t)1 << 20) ? g->arena_live : ((uint64_t)1 << 20))) return DBHIP_OK;
  g->arena_pinned = 0;
  DBHIP_CHECK(hipMemsetAsync(acc, 0, 8, s));
  hipLaunchKernelGGL(gb_pin_strings_kernel, dim3(grid), dim3(256), 0, s, g->L, g->slot_hash, g->rows, g->cap, (uint64_t)g->arena,
                     (uint64_t)g->arena + g->arena_cap, 3, g->arena, (int64_t)0, g->ctrl, acc);
  DBHIP_LAUNCH_CHECK();
  uint64_t two[2] = {0, 0};   // live bytes, bytes in use
  DBHIP_CHECK(hipMemcpyAsync(&two[0], acc, 8, hipMemcpyDeviceToHost, s));
  DBHIP_CHECK(hipMemcpyAsync(&two[1], &g->ctrl[8], 8, hipMemcpyDeviceToHost, s));
  DBHIP_CHECK(hipStreamSynchronize(s));
  g->arena_live = two[0];
  if (two[1] <= 2 * two[0] + ((uint64_t)1 << 20)) return DBHIP_OK;
  size_t want = (size_t)1 << 20;
  while (want < 2 * two[0]) want *= 2;
  uint8_t* na = nullptr;
  if ((rc = dbhip_alloc(want, (void**)&na))) return rc;
  DBHIP_CHECK(hipMemsetAsync(&g->ctrl[8], 0, 8, s));
  hipLaunchKernelGGL(gb_pin_strings_kernel, dim3(grid), dim3(256), 0, s, g->L, g->slot_hash, g->rows, g->cap, (uint64_t)g->arena,
                     (uint64_t)g->arena + g->arena_cap, 4, na, (int64_t)0, g->ctrl, acc);
  DBHIP_LAUNCH_CHECK();
  DBHIP_CHECK(hipStreamSynchronize(s));
  if (getenv("DBHIP_TRACE")) fprintf(stderr, "[dbhip] groupby: string arena compacted, %llu bytes in use -> %llu live (capacity %zu -> %zu)\n",
                                     (unsigned long long)two[1], (unsigned long long)two[0], g->arena_cap, want);
  (void)dbhip_free(g->arena);
  g->arena = na;
  g->arena_cap = want;
  return DBHIP_OK;
}
// after a kernel that summed the long-string bytes of its rows into ctrl[9]: read it, remember that the table holds long
// strings, make room
int32_t reserve_arena_for_chunk(dbhip_groupby* g, hipStream_t s) {
  uint64_t lb = 0;
  DBHIP_CHECK(hipMemcpyAsync(&lb, &g->ctrl[9], 8, hipMemcpyDeviceToHost, s));
  DBHIP_CHECK(hipStreamSynchronize(s));
  if (lb == 0) return DBHIP_OK;
  g->has_long = 1;
  return reserve_arena(g, lb, s);
}
bool layout_has_strings(const GbLayout& L) { return L.str_w1_mask != 0; }
bool layout_has_wide_minmax(const GbLayout& L);

// probe + accumulate + retry over rows_in[n] (device rows in table layout)
int32_t merge_rows_unpinned(dbhip_groupby* g, const uint64_t* rows_in, int64_t n, hipStream_t s, const uint64_t* n_dev, const uint64_t* abort_dev,
                            int deferred = 0);
int32_t merge_rows(dbhip_groupby* g, const uint64_t* rows_in, int64_t n, hipStream_t s,
                   const uint64_t* n_dev = nullptr, const uint64_t* abort_dev = nullptr) {
  int32_t rc = merge_rows_unpinned(g, rows_in, n, s, n_dev, abort_dev);
  if (rc == DBHIP_OK && n > 0) rc = pin_string_states(g, s);
  return rc;
}
// `deferred` (the pipelined fused aggregation): the caller has made sure that the table cannot outgrow its load factor whatever the
// rows hold; the three kernels are queued and NOTHING is read back — the table's count_host is stale until the caller's checkpoint.
// deferred == 2 (the queued partial-state exchange, gbk_api.h): the same, but the rows are other ranks' groups — any number of them —
// so the accumulate kernel is chosen by the bound n as on the synchronous path, not pinned to the handful-of-groups kernel
int32_t merge_rows_unpinned(dbhip_groupby* g, const uint64_t* rows_in, int64_t n, hipStream_t s,
                            const uint64_t* n_dev, const uint64_t* abort_dev, int deferred) {
  if (n == 0) return DBHIP_OK;
  if (n > 0xFFFFFFF0LL) {
    set_error("groupby: more than 2^32 rows in one call");
    return DBHIP_ERR_INVALID;
  }
  int32_t rc;
  if ((rc = ensure((void**)&g->gid, &g->gid_cap, (size_t)n * 4))) return rc;
  if ((rc = ensure((void**)&g->retry, &g->retry_cap, (size_t)n * 4))) return rc;
  const int grid = grid_for(n, 256);
  uint64_t* host_ctrl = pinned_words(0);
  if (!host_ctrl) return DBHIP_ERR_HIP;
  const uint64_t* cur_rows = rows_in;
  int64_t cur_n = n;
  const DevCount dc{n_dev, abort_dev};
  // No growth possible even if every row were a new group: probe, accumulate and retry are queued back to back and
  // the host reads the control block ONCE (the small merges behind the fused kernels are all host round trips).
  // (n is the caller's BOUND: one workgroup walks 256 rows per step at ~25 us a step — 2,048 partial rows behind dbhip_q1_fused took 240 us
  //  this way against 70 us through the three kernels, r06 probe — so only merges of at most one step take it)
  static const int64_t small_n = exp_env("DBHIP_GB_SMALL_MERGE") ? atoll(exp_env("DBHIP_GB_SMALL_MERGE")) : 256;
  if ((deferred || (g->count_host + n) * 135 <= g->cap * 100) && n <= small_n) {
    // one launch of one workgroup instead of a memset and three kernels (gb_merge_small_kernel: such a merge is bound by the host's cost per
    // queued operation)
    hipLaunchKernelGGL(gb_merge_small_kernel, dim3(1), dim3(256), 0, s, g->L, cur_rows, cur_n, g->slot_hash, g->rows, g->cap, g->hash_mask, g->gid,
                       g->retry, g->ctrl, dc, g->arena);
    DBHIP_LAUNCH_CHECK();
    if (deferred) return DBHIP_OK;
    DBHIP_CHECK(hipMemcpyAsync(host_ctrl, g->ctrl, 5 * sizeof(uint64_t), hipMemcpyDeviceToHost, s));
    DBHIP_CHECK(hipStreamSynchronize(s));
    g->count_host = (int64_t)host_ctrl[0];
    if (host_ctrl[3] & 2) {
      set_error("groupby: a string key longer than 12 bytes was met; keep the CPU operator for this block");
      return DBHIP_ERR_UNSUPPORTED;
    }
    if (host_ctrl[1]) {
      set_error("groupby: collision chain filled the table during retry; create the table with a larger capacity");
      return DBHIP_ERR_CAPACITY;
    }
    return DBHIP_OK;
  }
  if (deferred || (g->count_host + n) * 135 <= g->cap * 100) {
    DBHIP_CHECK(hipMemsetAsync(&g->ctrl[1], 0, 16, s));
    hipLaunchKernelGGL(gb_probe_kernel, dim3(grid), dim3(256), 0, s, g->L, cur_rows, cur_n, g->slot_hash, g->rows, g->cap,
                       g->hash_mask, g->gid, g->ctrl, dc, g->arena);
    // (the wave-combining kernel only where the table is known to hold a handful of groups: on an empty table the first
    // merge may bring 50 K groups, r02y: 0.11 ms there against 0.02 ms for the plain kernel)
    // (a Decimal128 min / max state is merged under a lock: always combine the rows of a wave first, one acquisition per wave and state)
    if (deferred == 1 || (!deferred && g->count_host > 0 && g->count_host <= 32 && n <= 65536) || n <= 2048 || layout_has_wide_minmax(g->L))
      // (deferred = the pipelined fused aggregation's window merge: a handful of groups, every wave's leader lane ends in atomics on the
      // SAME few rows — 256 waves cost 30 us of serialised atomics for 16 K partial rows; 16 workgroups walk them grid-stride instead)
      hipLaunchKernelGGL(gb_accum_lowcard_kernel, dim3(deferred == 1 && grid > 16 ? 16 : grid), dim3(256), 0, s, g->L, cur_rows, cur_n, g->rows, g->gid, g->retry,
                         g->ctrl, dc, g->arena);
    else
      hipLaunchKernelGGL(gb_accum_kernel, dim3(grid), dim3(256), 0, s, g->L, cur_rows, cur_n, g->rows, g->gid, g->retry, g->ctrl, dc, g->arena);
    hipLaunchKernelGGL(gb_retry_kernel, dim3(1), dim3(64), 0, s, g->L, cur_rows, g->slot_hash, g->rows, g->cap, g->hash_mask,
                       g->gid, g->retry, g->ctrl, g->arena);
    DBHIP_LAUNCH_CHECK();
    if (deferred) return DBHIP_OK;
    DBHIP_CHECK(hipMemcpyAsync(host_ctrl, g->ctrl, 5 * sizeof(uint64_t), hipMemcpyDeviceToHost, s));
    DBHIP_CHECK(hipStreamSynchronize(s));
    g->count_host = (int64_t)host_ctrl[0];
    if (host_ctrl[3] & 2) {
      set_error("groupby: a string key longer than 12 bytes was met; keep the CPU operator for this block");
      return DBHIP_ERR_UNSUPPORTED;
    }
    if (host_ctrl[1]) {
      set_error("groupby: collision chain filled the table during retry; create the table with a larger capacity");
      return DBHIP_ERR_CAPACITY;
    }
    return DBHIP_OK;
  }
  if (n_dev || abort_dev) {
    set_error("groupby: device-side row count needs a table that cannot grow during the merge");
    return DBHIP_ERR_INVALID;
  }
  for (int attempt = 0; attempt < 40; ++attempt) {
    // ctrl[1] (overflow) and ctrl[2] (retry count) are per-attempt
    DBHIP_CHECK(hipMemsetAsync(&g->ctrl[1], 0, 16, s));
    hipLaunchKernelGGL(gb_probe_kernel, dim3(grid), dim3(256), 0, s, g->L, cur_rows, cur_n, g->slot_hash,
                       g->rows, g->cap, g->hash_mask, g->gid, g->ctrl, dc, g->arena);
    DBHIP_LAUNCH_CHECK();
    DBHIP_CHECK(hipMemcpyAsync(host_ctrl, g->ctrl, 5 * sizeof(uint64_t), hipMemcpyDeviceToHost, s));
    DBHIP_CHECK(hipStreamSynchronize(s));
    bool too_full = host_ctrl[1] != 0 || (int64_t)host_ctrl[0] * 135 > g->cap * 100;
    if (too_full) {
      if ((rc = grow(g, s))) return rc;
      continue;  // redo the (idempotent) probe against the bigger table
    }
    g->count_host = (int64_t)host_ctrl[0];
    if (g->count_host <= 32 || layout_has_wide_minmax(g->L)) {
      hipLaunchKernelGGL(gb_accum_lowcard_kernel, dim3(grid), dim3(256), 0, s, g->L, cur_rows, cur_n,
                         g->rows, g->gid, g->retry, g->ctrl, dc, g->arena);
    } else {
      hipLaunchKernelGGL(gb_accum_kernel, dim3(grid), dim3(256), 0, s, g->L, cur_rows, cur_n, g->rows,
                         g->gid, g->retry, g->ctrl, dc, g->arena);
    }
    hipLaunchKernelGGL(gb_retry_kernel, dim3(1), dim3(64), 0, s, g->L, cur_rows, g->slot_hash, g->rows,
                       g->cap, g->hash_mask, g->gid, g->retry, g->ctrl, g->arena);
    DBHIP_LAUNCH_CHECK();
    DBHIP_CHECK(hipMemcpyAsync(host_ctrl, g->ctrl, 5 * sizeof(uint64_t), hipMemcpyDeviceToHost, s));
    DBHIP_CHECK(hipStreamSynchronize(s));
    g->count_host = (int64_t)host_ctrl[0];
    if (host_ctrl[3] & 2) {
      set_error("groupby: a string key longer than 12 bytes was met; keep the CPU operator for this block");
      return DBHIP_ERR_UNSUPPORTED;
    }
    if (host_ctrl[1] & 2) {
      // The retry path ran out of room after part of the block was accumulated.
      // Forced-collision corner (test hook) only: report instead of double counting.
      set_error("groupby: collision chain filled the table during retry; create the table with a larger capacity");
      return DBHIP_ERR_CAPACITY;
    }
    return DBHIP_OK;
  }
  set_error("groupby: table did not converge after repeated growth");
  return DBHIP_ERR_CAPACITY;
}


// ---------------------------------------------------------------------------
// LDS pre-aggregation path of add_block ("partial aggregation inside the workgroup").
//
// The reference bounds its partial AggregateHashTable to the CPU cache and lets duplicates through
// (aggregate/mod.rs:98-124, aggregate_hashtable.rs:277-290); the device analogue is a hash table in
// the workgroup's LDS: every row is hashed, matched/claimed in the LDS table (64-bit CAS on the hash
// word) and its state contribution merged with LDS atomics, so low- and medium-cardinality group-bys
// touch HBM only to read the argument columns once (coalesced, 2 rows per lane in flight).
// At the end each workgroup flushes its <= LCAP partial rows; they are merged into the HBM table by
// the row path above, exactly like partial payloads in TransformFinalAggregate. Rows that do not
// fit the LDS table (it is full, or a true 64-bit hash collision) are serialized to a spill buffer
// and go through the row path as well. Layout limits of this path: <= 4 key words, <= 6 aggregates.
// Key equality inside one tile is decided after a workgroup barrier (claim by hash, verify after the
// barrier), so no lane ever spins on another lane.
// ---------------------------------------------------------------------------
constexpr int FK_MAXKW = 4;
constexpr int FK_MAXA = 6;
constexpr uint32_t FK_SPILL = 0xFFFFFFFFu;

// Register image of one input row. KW / NA / HI are compile-time bounds of the layout class
// (key words, aggregates, "some argument needs a second word" = Decimal128), so that the small and
// common shapes (1-2 integer keys, sum + count) keep 8 rows per lane in flight.
template <int KW, int NA, bool HI>
struct FkRow {
  uint64_t kw[KW];
  uint64_t h;
  uint64_t aw[NA];
  uint64_t ah[HI ? NA : 1];
  uint32_t avalid;
};

template <int KW>
__device__ __forceinline__ void fk_put(uint64_t (&a)[KW], int off, uint64_t v) {
#pragma unroll
  for (int j = 0; j < KW; ++j) a[j] = (j == off) ? v : a[j];
}

template <int KW, int NA, bool HI>
__device__ __forceinline__ void fk_load(const GbLayout& L, const GbCols& C, int64_t i, FkRow<KW, NA, HI>& r, uint64_t* ctrl) {
  uint64_t h = 0, vmask = 0;
#pragma unroll
  for (int j = 0; j < KW; ++j) r.kw[j] = 0;
#pragma unroll
  for (int k = 0; k < KW; ++k) {
    if (k < L.nkeys) {
      uint64_t w[2];
      bool valid;
      if (!gb_load_words(C.key[k], i, w, &valid)) atomicOr((unsigned long long*)&ctrl[3], 2ULL);
      const uint64_t hk = gb_hash_words(L.key_type[k], w, valid);
      h = (k == 0) ? hk : merge_hash(h, hk);
      fk_put<KW>(r.kw, L.key_off[k], w[0]);
      if (L.key_words[k] == 2) fk_put<KW>(r.kw, L.key_off[k] + 1, w[1]);
      if (valid) vmask |= 1ULL << k;
    }
  }
  if (L.validity_word >= 0) fk_put<KW>(r.kw, L.validity_word, vmask);
  r.h = h;
  r.avalid = 0;
  if (HI) {
#pragma unroll
    for (int a = 0; a < (HI ? NA : 1); ++a) r.ah[a] = 0;
  }
#pragma unroll
  for (int a = 0; a < NA; ++a) {
    r.aw[a] = 0;
    if (a < L.naggs) {
      uint64_t w[2] = {0, 0};
      bool valid = true;
      if (C.arg[a].data != nullptr) gb_load_words(C.arg[a], i, w, &valid);
      r.aw[a] = w[0];
      if (HI) r.ah[a] = w[1];
      if (valid) r.avalid |= 1u << a;
    }
  }
}

// fk_load for the R rows of a lane's tile, column by column (gb_load_words_n): with the per-row version every load sits in
// its own basic block behind the type switch's scalar branch and the R x (keys + arguments) loads of a tile are serialised
// memory round trips — what bounded this path at ~1.0 ms per 60 M rows whatever the group count (r02u)
template <int KW, int NA, bool HI, int R>
__device__ __forceinline__ void fk_load_n(const GbLayout& L, const GbCols& C, const int64_t (&row)[R], FkRow<KW, NA, HI> (&r)[R], uint64_t* ctrl) {
  uint64_t vmask[R];
#pragma unroll
  for (int x = 0; x < R; ++x) {
    vmask[x] = 0;
    r[x].h = 0;
    r[x].avalid = 0;
#pragma unroll
    for (int j = 0; j < KW; ++j) r[x].kw[j] = 0;
  }
#pragma unroll
  for (int k = 0; k < KW; ++k) {
    if (k < L.nkeys) {
      uint64_t w0[R], w1[R];
      bool valid[R];
      if (!gb_load_words_n<R>(C.key[k], row, w0, w1, valid)) atomicOr((unsigned long long*)&ctrl[3], 2ULL);
      const int type = L.key_type[k], off = L.key_off[k];
      const bool two = L.key_words[k] == 2;
#pragma unroll
      for (int x = 0; x < R; ++x) {
        const uint64_t w[2] = {w0[x], w1[x]};
        const uint64_t hk = gb_hash_words(type, w, valid[x]);
        r[x].h = (k == 0) ? hk : merge_hash(r[x].h, hk);
        fk_put<KW>(r[x].kw, off, w0[x]);
        if (two) fk_put<KW>(r[x].kw, off + 1, w1[x]);
        if (valid[x]) vmask[x] |= 1ULL << k;
      }
    }
  }
  if (L.validity_word >= 0) {
#pragma unroll
    for (int x = 0; x < R; ++x) fk_put<KW>(r[x].kw, L.validity_word, vmask[x]);
  }
#pragma unroll
  for (int a = 0; a < NA; ++a) {
#pragma unroll
    for (int x = 0; x < R; ++x) {
      r[x].aw[a] = 0;
      if (HI) r[x].ah[a] = 0;
    }
    if (a < L.naggs) {
      uint64_t w0[R], w1[R];
      bool valid[R];
#pragma unroll
      for (int x = 0; x < R; ++x) { w0[x] = 0; w1[x] = 0; valid[x] = true; }
      if (C.arg[a].data != nullptr) gb_load_words_n<R>(C.arg[a], row, w0, w1, valid);
#pragma unroll
      for (int x = 0; x < R; ++x) {
        r[x].aw[a] = w0[x];
        if (HI) r[x].ah[a] = w1[x];
        if (valid[x]) r[x].avalid |= 1u << a;
      }
    }
  }
}

struct FkArgs {
  int64_t row0, n;         // rows [row0, row0 + n) of the columns
  int64_t tiles_per_block;
  int lcap, sw;            // LDS table capacity (pow2) and row stride in words
  uint32_t llimit;         // max occupied LDS slots
  uint64_t hash_mask;
  uint64_t* partial;       // [gridDim.x * lcap][W]
  uint64_t* spill;         // [spill_cap][W]
  uint64_t spill_cap;
  uint64_t* ctrl;          // [5] = #partial rows, [6] = #spill rows, [3] error bits
};

template <int KW, int NA, bool HI, int R, int THREADS = 256>
__global__ __launch_bounds__(THREADS) void gb_lds_preagg_kernel(GbLayout L, GbCols C, FkArgs A) {
  extern __shared__ uint64_t fk_lds[];
  __shared__ uint32_t lcount;
  uint64_t* lhash = fk_lds;
  uint64_t* lrows = fk_lds + A.lcap;
  const int tid = threadIdx.x;
  const uint32_t lmask = (uint32_t)A.lcap - 1;
  for (int s = tid; s < A.lcap; s += THREADS) lhash[s] = 0;
  if (tid == 0) lcount = 0;
  __syncthreads();

  const int64_t tile_rows = THREADS * R;
  const int64_t t_begin = (int64_t)blockIdx.x * A.tiles_per_block;
  const int64_t ntiles = (A.n + tile_rows - 1) / tile_rows;
  int64_t t_end = t_begin + A.tiles_per_block;
  if (t_end > ntiles) t_end = ntiles;

  for (int64_t t = t_begin; t < t_end; ++t) {
    FkRow<KW, NA, HI> r[R];
    uint32_t slot[R];
    // ---- loads of the whole tile first (R rows per lane in flight) ----
    {
      int64_t row[R];
#pragma unroll
      for (int x = 0; x < R; ++x) {
        const int64_t li = t * tile_rows + x * THREADS + tid;
        row[x] = A.row0 + (li < A.n ? li : 0);
      }
      fk_load_n<KW, NA, HI, R>(L, C, row, r, A.ctrl);
    }
    // ---- phase A: match-or-claim by hash ----
#pragma unroll
    for (int x = 0; x < R; ++x) {
      const int64_t li = t * tile_rows + x * THREADS + tid;
      slot[x] = FK_SPILL;
      if (li < A.n && gb_row_passes(C, A.row0 + li)) {
        const uint64_t hw = probe_word(r[x].h, A.hash_mask);
        uint32_t pos = (uint32_t)hw & lmask;
        for (int step = 0; step < 64; ++step) {
          uint64_t cur = ((volatile uint64_t*)lhash)[pos];
          if (cur == 0) {
            if (((volatile uint32_t*)&lcount)[0] >= A.llimit) break;
            const unsigned long long old = atomicCAS((unsigned long long*)&lhash[pos], 0ULL, (unsigned long long)hw);
            if (old == 0) {
              atomicAdd(&lcount, 1u);
              uint64_t* d = lrows + (size_t)pos * A.sw;
#pragma unroll
              for (int j = 0; j < KW; ++j)
                if (j < L.nkey_words) d[j] = r[x].kw[j];
              d[L.hash_word] = r[x].h;
#pragma unroll
              for (int a = 0; a < NA; ++a)
                if (a < L.naggs) gb_state_identity(L, a, d + L.agg_off[a]);
              slot[x] = pos;
              break;
            }
            cur = old;
          }
          if (cur == hw) { slot[x] = pos; break; }
          pos = (pos + 1) & lmask;
        }
      } else {
        slot[x] = FK_SPILL - 1;  // padding row: neither aggregated nor spilled
      }
    }
    __syncthreads();  // keys and identity states of every slot claimed in this tile are visible
    // ---- phase B: verify keys, merge states with LDS atomics; the rest spills ----
#pragma unroll
    for (int x = 0; x < R; ++x) {
      bool spill = slot[x] == FK_SPILL;
      if (slot[x] < FK_SPILL - 1) {
        uint64_t* d = lrows + (size_t)slot[x] * A.sw;
        bool eq = true;
#pragma unroll
        for (int j = 0; j < KW; ++j)
          if (j < L.nkey_words) eq &= (d[j] == r[x].kw[j]);
        if (eq) {
#pragma unroll
          for (int a = 0; a < NA; ++a)
            if (a < L.naggs) {
              uint64_t v[GB_MAX_STATE_WORDS];
              gb_row_contrib(L, a, r[x].aw[a], HI ? r[x].ah[a] : 0, (r[x].avalid >> a) & 1, v);
              gb_atomic_merge(L, a, d + L.agg_off[a], v);
            }
        } else {
          spill = true;  // same probe hash, different keys
        }
      }
      const uint64_t m = __ballot(spill);
      if (m) {
        const int leader = __ffsll((long long)m) - 1;
        unsigned long long base = 0;
        if (lane_id() == leader) base = atomicAdd((unsigned long long*)&A.ctrl[6], (unsigned long long)__popcll(m));
        base = __shfl(base, leader, 64);
        if (spill) {
          const unsigned long long si = base + __popcll(m & ((1ULL << lane_id()) - 1));
          // a chunk whose key distribution was trusted gets a small spill buffer: rows past it are dropped and flagged
          // (ctrl[3] bit 2) — the host then discards the whole chunk's output and redoes it on another path
          if (si >= A.spill_cap) { atomicOr((unsigned long long*)&A.ctrl[3], 4ULL); continue; }
          uint64_t* o = A.spill + si * L.W;
#pragma unroll
          for (int j = 0; j < KW; ++j)
            if (j < L.nkey_words) o[j] = r[x].kw[j];
          o[L.hash_word] = r[x].h;
#pragma unroll
          for (int a = 0; a < NA; ++a)
            if (a < L.naggs) {
              uint64_t v[GB_MAX_STATE_WORDS];
              gb_row_contrib(L, a, r[x].aw[a], HI ? r[x].ah[a] : 0, (r[x].avalid >> a) & 1, v);
              for (int k = 0; k < L.agg_words[a]; ++k) o[L.agg_off[a] + k] = v[k];
            }
        }
      }
    }
    // no barrier needed here: the next tile only adds NEW slots; slots matched above never change keys
  }
  __syncthreads();
  // ---- flush the workgroup's partial rows (one cursor atomic per wave, not per row) ----
  for (int s = tid; s < A.lcap; s += THREADS) {  // lcap is a multiple of THREADS: the loop is wave-uniform
    const bool occ = lhash[s] != 0;
    const uint64_t m = __ballot(occ);
    unsigned long long base = 0;
    if (m && lane_id() == 0) base = atomicAdd((unsigned long long*)&A.ctrl[5], (unsigned long long)__popcll(m));
    base = __shfl(base, 0, 64);
    if (occ) {
      const unsigned long long idx = base + __popcll(m & ((1ULL << lane_id()) - 1));
      const uint64_t* src = lrows + (size_t)s * A.sw;
      uint64_t* o = A.partial + idx * L.W;
      for (int k = 0; k < L.W; ++k) o[k] = src[k];
    }
  }
}

int32_t add_chunk_partitioned(dbhip_groupby* g, const GbCols& C, int64_t row0, int64_t cn, hipStream_t s,
                              int64_t* spilled);
}  // namespace

// k_groupby.hip — group hash (a7) and the HBM hash-aggregation table (a8-a13).
//
// Reference: AggregateHashTable (src/query/expression/src/aggregate/aggregate_hashtable.rs:168-408)
// = group_hash_entries -> HashIndex::probe_and_create (hash_index/index.rs:148-216) ->
// Payload rows (payload.rs) -> AggregateFunction::accumulate_keys / batch_merge.
// The reference's SwissTable-of-row-pointers is a CPU-cache design; results are
// defined as a SET of (group keys, states) (tests compare sorted), so the device
// table is free to use its own geometry:
//
//   slot_hash[cap]  u64, 0 = empty; claimed with one 64-bit atomicCAS on the group
//                   hash itself (atomics are coherent across XCDs; plain payload
//                   writes are not, so no kernel reads a key another workgroup of the
//                   SAME launch wrote — MI355X_MICROARCH.md §inter-workgroup visibility)
//   rows[cap][W]    the group's row (keys, hash, states), see gb_layout.h
//
//   add_block / merge_serialized:
//     serialize  columns -> rows_in[n][W] (keys, hash, per-row state contribution)
//     probe      slot = first slot on the linear probe path whose hash word equals
//                the row's hash (insert if an empty slot is met first)       [launch 1]
//     accumulate verify the keys against rows[slot] (visible: launch boundary), then
//                merge the contribution with atomics; low-cardinality tables first
//                combine equal slots inside the wave                         [launch 2]
//     retry      rows whose keys differ from the slot's keys (a true 64-bit hash
//                collision) continue the probe serially                      [launch 3]
//   Growth: when the load factor 1/1.35 (aggregate/mod.rs:55) is exceeded the table is
//   rebuilt x4 (aggregate_hashtable.rs:314-333) and the block's probe is redone
//   (the probe phase is idempotent; states are untouched until it succeeds).
#include <mutex>
#include "gb_device.h"
#include "runtime.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <new>
#include <vector>

using namespace dbhip;

#define GB_INVALID_SLOT 0xFFFFFFFFu

struct dbhip_groupby {
  int64_t hint_groups;   // dbhip_groupby_create's initial_capacity: the caller's estimate of the number of groups
  GbLayout L;
  int64_t cap;            // power of two
  uint64_t* slot_hash;    // [cap]
  uint64_t* rows;         // [cap * W]
  uint64_t* ctrl;         // device control block: [0]=count [1]=overflow [2]=retry_n [3]=error [4]=flush_n
  uint64_t hash_mask;     // test hook: AND-mask applied to the probe hash (forces collisions)
  int64_t count_host;     // groups known to the host after the last sync
  // per-call scratch (owned by the table so concurrent tables do not share it)
  uint64_t* rows_in; size_t rows_in_cap;
  uint32_t* gid; size_t gid_cap;
  uint32_t* retry; size_t retry_cap;
  // LDS pre-aggregation path (add_block on short layouts)
  uint64_t* partial; size_t partial_cap;   // per-workgroup partial rows
  int fast_disabled;                       // set once most rows of a chunk spilled (high NDV)
  int fast_trusted;                        // last chunk spilled < 1 %: no more probing chunks
  int lds_big;                             // small layouts whose groups outgrew the 48 KB table but fit a 96 KB one (1024-thread workgroups)
  // radix-partitioned pre-aggregation (medium cardinality)
  int part_bits;                           // 0 = undecided, > 0 = log2(partitions), < 0 = not worth it (row path)
  int part_forbidden;                      // test hook: never choose the partitioned path
  int64_t part_min_rows;                   // smallest chunk worth partitioning
  int64_t part_chunk;                      // rows per partitioned chunk (0 = PT_CHUNK)
  int part_direct;                         // partitioned rows are inserted straight into their table slice (no LDS pre-aggregation)
  int part_adapt;                          // the chunk size follows the group estimate (more groups than one chunk's LDS tables hold)
  int part_validate, part_validated;       // the next partitioned chunk is a 4 M-row check of an estimate extrapolated from a mostly-distinct probe
  int64_t rows_seen;                       // input rows of add_block so far (cardinality estimate)
  uint32_t* part_meta; size_t part_meta_cap;   // tot[PT_PMAX] | base[PT_PMAX + 8] | pcount[PT_PMAX] | mat[nwg][P]
  uint32_t* spill_idx; size_t spill_idx_cap;
  uint64_t* spill_rows; size_t spill_rows_cap;
  uint8_t* arena; size_t arena_cap;        // bytes of the long (> 12 B) string keys of the groups; cursor = ctrl[8]
  int has_long;                            // a long string key was met: the LDS / partitioned paths decline, the row path runs
  uint64_t* xcur;                          // exchange partitioning: cursor[4096] | base[4097]
  int fagg_disabled;                       // the fused few-groups kernel (k_fagg.hip) gave up on this table's keys / shape
  // compact-row kernels (gb_compact.h, round 4)
  int gbc_off;                             // test hook / fallback: never use them for this table
  int gbc_active;                          // the add_block call being worked on goes through them (layout AND columns qualify)
  int gbc_lcap;                            // LDS table slots of the no-partition path (0 = not chosen yet)
  int gbc_nodirect;                        // a partition outgrew its fixed region once (heavy keys): histogram path from now on
  int gbc_part_lcap_max;                   // largest partition table of the compact kernels, chosen with the partitioning (0 = default)
  uint32_t gbc_part_cap;                   // rows of a partition's region when the last scatter was the direct one, else 0
  uint64_t* gbc_spill; size_t gbc_spill_cap;   // rows (table layout) that did not fit an LDS table
  uint64_t arena_pinned, arena_live;           // bytes pinned since the arena's live bytes were last counted; that count
  int gbc_skip;                                // this chunk goes through the generic kernels (its spill list covers every row)
  uint32_t* gbc_split; size_t gbc_split_cap;   // heavy partitions (gbc_split_map_kernel): nsp[P] | cursor, extra workgroups | map[extra]
  void* fa_pipe;                               // pipelined fused aggregation (k_fagg.hip, dbhip_groupby_set_pipelined): blocks queued, not yet checked
};

// k_fagg.hip: a pipelined table owes the host a checkpoint before anything else looks at (or changes) its groups
int32_t dbhip_fagg_pipe_drain_internal(dbhip_groupby* g, void* pipe, hipStream_t s);
void dbhip_fagg_pipe_destroy_internal(void* pipe);
int32_t dbhip_fagg_pipe_reset_internal(void* pipe, hipStream_t s);
#define GB_DRAIN(g, s)                                                                   \
  do {                                                                                   \
    if ((g) && (g)->fa_pipe) {                                                           \
      const int32_t _rc = dbhip_fagg_pipe_drain_internal((g), (g)->fa_pipe, (s));        \
      if (_rc) return _rc;                                                               \
    }                                                                                    \
  } while (0)

namespace {

// ---------------------------------------------------------------------------
// dbhip_group_hash
// ---------------------------------------------------------------------------
struct HashCols {
  GbCol c[GB_MAX_KEYS];
  int n;
};

// Four rows per lane (rows base + u T + t), column by column: the four loads of a column are independent instructions
// in one basic block (gb_load_words_n); one row per lane leaves 8 bytes per lane in flight, which is latency bound.
__global__ __launch_bounds__(256) void group_hash_kernel(HashCols hc, int64_t n, uint64_t* out,
                                                         unsigned long long* bad) {
  constexpr int U = 4;
  const int64_t T = (int64_t)gridDim.x * blockDim.x;
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (int64_t base = 0; base < n; base += U * T) {
    int64_t row[U];
    bool in[U];
    uint64_t h[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      row[u] = base + u * T + t;
      in[u] = row[u] < n;
      if (!in[u]) row[u] = n - 1;
      h[u] = 0;
    }
    for (int k = 0; k < hc.n; ++k) {
      if (hc.c[k].type == DBHIP_T_STRING) {
        // general strings (any length): hash the bytes where they live
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int64_t j = hc.c[k].is_scalar ? 0 : row[u];
          const bool valid = !hc.c[k].validity || bit_get(hc.c[k].validity, hc.c[k].voff + j);
          const uint32_t* v = (const uint32_t*)hc.c[k].data + 4 * j;
          const uint32_t len = v[0];
          const uint8_t* p = len <= 12 ? (const uint8_t*)(v + 1) : (const uint8_t*)hc.c[k].buffers[v[2]] + v[3];
          const uint64_t hk = valid ? agg_hash_bytes(p, len) : DBHIP_NULL_HASH_VAL;
          h[u] = (k == 0) ? hk : merge_hash(h[u], hk);
        }
      } else {
        uint64_t w0[U], w1[U];
        bool valid[U];
        if (!gb_load_words_n<U>(hc.c[k], row, w0, w1, valid)) atomicAdd(bad, 1ULL);
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const uint64_t w[2] = {w0[u], w1[u]};
          const uint64_t hk = gb_hash_words(hc.c[k].type, w, valid[u]);
          h[u] = (k == 0) ? hk : merge_hash(h[u], hk);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (in[u]) out[row[u]] = h[u];
  }
}

// ---------------------------------------------------------------------------
// serialize: columns -> rows_in
// ---------------------------------------------------------------------------
// Four rows per lane, column by column (gb_load_words_n): the loads of a column are in flight together.
__device__ __forceinline__ bool gb_row_passes(const GbCols& C, int64_t row) {
  return !C.filter || bit_get(C.filter, C.filter_off + row);
}

// With a predicate Bitmap (C.filter) the passing rows are written densely (wave ballot + one cursor atomic per wave,
// ctrl[7] = number of rows written; their order is not the input order, which no consumer depends on).
__global__ __launch_bounds__(256) void gb_serialize_kernel(GbLayout L, GbCols C, int64_t row0, int64_t n,
                                                           uint64_t* rows_in, uint64_t* ctrl) {
  constexpr int U = 4;
  const int64_t T = (int64_t)gridDim.x * blockDim.x;
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (int64_t base = 0; base < n; base += U * T) {   // (n, T: wave-uniform trip count)
    int64_t li[U], row[U];
    bool in[U];
    uint64_t h[U], vmask[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      li[u] = base + u * T + t;
      in[u] = li[u] < n;
      if (!in[u]) li[u] = n - 1;
      row[u] = row0 + li[u];
      h[u] = 0; vmask[u] = 0;
      if (C.filter) {
        in[u] = in[u] && gb_row_passes(C, row[u]);
        const uint64_t m = __ballot(in[u]);
        unsigned long long b0 = 0;
        if (m && lane_id() == 0) b0 = atomicAdd((unsigned long long*)&ctrl[7], (unsigned long long)__popcll(m));
        b0 = __shfl(b0, 0, 64);
        if (in[u]) li[u] = (int64_t)b0 + __popcll(m & ((1ULL << lane_id()) - 1));
      }
    }
    for (int k = 0; k < L.nkeys; ++k) {
      uint64_t w0[U], w1[U], hlong[U];
      bool valid[U], is_long[U];
#pragma unroll
      for (int u = 0; u < U; ++u) is_long[u] = false;
      if (L.key_type[k] == DBHIP_T_STRING) {
        // strings of ANY length: short ones as canonical inline words, long ones as (len | prefix, address of the bytes) with
        // the hash of the bytes (group_hash.rs:522-553); their sizes are summed so the host can make room in the arena
        const GbCol& kc = C.key[k];
        uint64_t long_bytes = 0;
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int64_t j = kc.is_scalar ? 0 : row[u];
          valid[u] = !kc.validity || bit_get(kc.validity, kc.voff + j);
          const uint32_t* v = (const uint32_t*)kc.data + 4 * j;
          const uint32_t len = v[0];
          uint64_t ww[2] = {0, 0};
          hlong[u] = 0;
          if (len <= 12 || !valid[u]) {
            bool vv;
            gb_load_words(kc, row[u], ww, &vv);
          } else {
            const uint8_t* p = (const uint8_t*)kc.buffers[v[2]] + v[3];
            ww[0] = ((uint64_t)v[1] << 32) | len;
            ww[1] = (uint64_t)p;
            hlong[u] = agg_hash_bytes(p, len);
            is_long[u] = true;
            if (in[u]) long_bytes += (len + 7) & ~7u;
          }
          w0[u] = ww[0]; w1[u] = ww[1];
        }
        long_bytes = wave_sum_u64(long_bytes);
        if (long_bytes && lane_id() == 0) atomicAdd((unsigned long long*)&ctrl[9], (unsigned long long)long_bytes);
      } else if (L.key_type[k] == DBHIP_T_DEC256) {
        // four little-endian words; the hash is AggHash for i256 = its 32 bytes through the byte hash (group_hash.rs:593-597)
        const GbCol& kc = C.key[k];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int64_t j = kc.is_scalar ? 0 : row[u];
          valid[u] = !kc.validity || bit_get(kc.validity, kc.voff + j);
          const uint64_t* p = (const uint64_t*)kc.data + 4 * j;
          uint64_t q[4] = {p[0], p[1], p[2], p[3]};
          if (!valid[u]) { q[0] = 0; q[1] = 0; q[2] = 0; q[3] = 0; }
          const uint64_t hk = valid[u] ? agg_hash_i256(q[0], q[1], q[2], q[3]) : DBHIP_NULL_HASH_VAL;
          h[u] = (k == 0) ? hk : merge_hash(h[u], hk);
          if (in[u]) {
            uint64_t* r = rows_in + li[u] * L.W + L.key_off[k];
            r[0] = q[0]; r[1] = q[1]; r[2] = q[2]; r[3] = q[3];
          }
          if (valid[u]) vmask[u] |= 1ULL << k;
        }
        continue;
      } else if (!gb_load_words_n<U>(C.key[k], row, w0, w1, valid)) atomicOr((unsigned long long*)&ctrl[3], 2ULL);
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const uint64_t w[2] = {w0[u], w1[u]};
        const uint64_t hk = is_long[u] ? hlong[u] : gb_hash_words(L.key_type[k], w, valid[u]);
        h[u] = (k == 0) ? hk : merge_hash(h[u], hk);
        if (in[u]) {
          uint64_t* r = rows_in + li[u] * L.W;
          r[L.key_off[k]] = w0[u];
          if (L.key_words[k] == 2) r[L.key_off[k] + 1] = w1[u];
        }
        if (valid[u]) vmask[u] |= 1ULL << k;
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (in[u]) {
        uint64_t* r = rows_in + li[u] * L.W;
        if (L.validity_word >= 0) r[L.validity_word] = vmask[u];
        r[L.hash_word] = h[u];
      }
    }
    for (int a = 0; a < L.naggs; ++a) {
      uint64_t w0[U], w1[U];
      bool valid[U];
#pragma unroll
      for (int u = 0; u < U; ++u) { w0[u] = 0; w1[u] = 0; valid[u] = true; }
      if (C.arg[a].data != nullptr && gb_sum256(L, a)) {
        // SUM over Decimal256: the row's contribution is the value's four words + its sign extension (+ the adaptor's flag)
        const GbCol& ac = C.arg[a];
        const int fw = L.agg_flag[a];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          if (!in[u]) continue;
          const int64_t j = ac.is_scalar ? 0 : row[u];
          const bool ok = !ac.validity || bit_get(ac.validity, ac.voff + j);
          const uint64_t* p = (const uint64_t*)ac.data + 4 * j;
          uint64_t* st = rows_in + li[u] * L.W + L.agg_off[a];
          st[0] = ok ? p[0] : 0; st[1] = ok ? p[1] : 0; st[2] = ok ? p[2] : 0; st[3] = ok ? p[3] : 0;
          st[4] = (ok && (p[3] >> 63)) ? ~0ULL : 0;
          if (fw) st[fw] = ok ? 1 : 0;
        }
        continue;
      }
      if (C.arg[a].data != nullptr && gb_minmax256(L, a)) {
        // MIN / MAX over Decimal256: (top word with the sign flipped, has, the three lower words from high to low)
        const GbCol& ac = C.arg[a];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          if (!in[u]) continue;
          const int64_t j = ac.is_scalar ? 0 : row[u];
          const bool ok = !ac.validity || bit_get(ac.validity, ac.voff + j);
          const uint64_t* p = (const uint64_t*)ac.data + 4 * j;
          uint64_t* st = rows_in + li[u] * L.W + L.agg_off[a];
          st[0] = ok ? (p[3] ^ (1ULL << 63)) : 0; st[1] = ok ? 1 : 0; st[2] = ok ? p[2] : 0; st[3] = ok ? p[1] : 0; st[4] = ok ? p[0] : 0;
        }
        continue;
      }
      if (C.arg[a].data != nullptr && C.arg[a].type == DBHIP_T_STRING) {
        // a String argument (min / max): short values as the canonical inline words, long ones as (len | prefix, ADDRESS of the bytes)
        const GbCol& ac = C.arg[a];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int64_t j = ac.is_scalar ? 0 : row[u];
          valid[u] = !ac.validity || bit_get(ac.validity, ac.voff + j);
          const uint32_t* v = (const uint32_t*)ac.data + 4 * j;
          const uint32_t len = v[0];
          uint64_t ww[2] = {0, 0};
          if (len <= 12 || !valid[u]) {
            bool vv;
            gb_load_words(ac, row[u], ww, &vv);
          } else if (ac.buffers) {
            ww[0] = ((uint64_t)v[1] << 32) | len;
            ww[1] = (uint64_t)((const uint8_t*)ac.buffers[v[2]] + v[3]);
          } else {
            atomicOr((unsigned long long*)&ctrl[3], 2ULL);   // a long view without data buffers
          }
          w0[u] = ww[0]; w1[u] = ww[1];
        }
      } else if (C.arg[a].data != nullptr) gb_load_words_n<U>(C.arg[a], row, w0, w1, valid);
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (!in[u]) continue;
        uint64_t* s = rows_in + li[u] * L.W + L.agg_off[a];
        uint64_t v[GB_MAX_STATE_WORDS];
        gb_row_contrib(L, a, w0[u], w1[u], valid[u], v);
        for (int k = 0; k < L.agg_words[a]; ++k) s[k] = v[k];
      }
    }
  }
}

// ---------------------------------------------------------------------------
// probe
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint64_t probe_word(uint64_t h, uint64_t mask) {
  uint64_t hw = h & mask;
  return hw == 0 ? 1 : hw;  // 0 is the empty marker
}
// Home slot = the TOP log2(cap) bits of the probe word: rows sorted by the top hash bits (the radix partitions of
// the scatter kernel) then walk the table front to back, slice by slice — with >= 10^6 groups the table is far
// larger than the L2 / Infinity Cache and random home slots cost one HBM sector each.
__device__ __forceinline__ uint64_t home_slot(uint64_t hw, int64_t cap) {
  return hw >> (__builtin_clzll((unsigned long long)cap) + 1);  // cap = 2^k: clz = 63 - k -> shift = 64 - k
}

// `n_dev` (optional): the row count lives on the device (rows produced by a kernel of the same stream whose
// count the host has not read yet); `abort_dev` (optional): non-zero low bits = the producer gave up, merge nothing.
struct DevCount {
  const uint64_t* n_dev;
  const uint64_t* abort_dev;
};
__device__ __forceinline__ int64_t dev_rows(const DevCount& dc, int64_t n) {
  if (dc.abort_dev && (*dc.abort_dev & 7)) return 0;   // 1: too many groups, 2: long string key, 4: row errors (sealed by the pipeline)
  if (dc.n_dev) { const int64_t m = (int64_t)*dc.n_dev; return m < n ? m : n; }
  return n;
}

__device__ __forceinline__ bool bytes_equal(const uint8_t* x, const uint8_t* y, uint32_t len) {
  for (uint32_t i = 0; i < len; ++i)
    if (x[i] != y[i]) return false;
  return true;
}
// `a`: an input / partial row (long strings by ADDRESS), `b`: a table row (long strings by arena OFFSET), row_match_entries
// (payload_row.rs:324+): fixed-width words compare as words, long strings by length + prefix (word 0) and then their bytes
__device__ __forceinline__ bool keys_equal(const GbLayout& L, const uint64_t* a, const uint64_t* b, const uint8_t* arena) {
  bool eq = true;
  for (int k = 0; k < L.nkey_words; ++k) {
    if (((L.str_w1_mask >> k) & 1) && (uint32_t)a[k - 1] > 12) {
      eq = eq && a[k - 1] == b[k - 1] && bytes_equal((const uint8_t*)a[k], arena + b[k], (uint32_t)a[k - 1]);
      continue;
    }
    eq &= (a[k] == b[k]);
  }
  return eq;
}
// a lane that claimed a slot writes the group's key words; long strings are copied into the arena (bump allocation: the
// host made room for every long byte of the chunk before the launch)
__device__ __forceinline__ void write_group_keys(const GbLayout& L, const uint64_t* r, uint64_t* d, uint8_t* arena, uint64_t* ctrl) {
  for (int k = 0; k < L.nkey_words; ++k) {
    if (((L.str_w1_mask >> k) & 1) && (uint32_t)r[k - 1] > 12) {
      const uint32_t len = (uint32_t)r[k - 1];
      const unsigned long long off = atomicAdd((unsigned long long*)&ctrl[8], (unsigned long long)((len + 7) & ~7u));
      const uint8_t* src = (const uint8_t*)r[k];
      for (uint32_t i = 0; i < len; ++i) arena[off + i] = src[i];
      d[k] = off;
      continue;
    }
    d[k] = r[k];
  }
}

__global__ __launch_bounds__(256) void gb_probe_kernel(GbLayout L, const uint64_t* rows_in, int64_t n,
                                                       uint64_t* slot_hash, uint64_t* rows, int64_t cap,
                                                       uint64_t hash_mask, uint32_t* gid, uint64_t* ctrl, DevCount dc, uint8_t* arena) {
  n = dev_rows(dc, n);
  const uint64_t cmask = (uint64_t)cap - 1;
  // the number of NEW groups is added to ctrl[0] ONCE PER WORKGROUP, after its last row (one atomic per new group on that
  // single address serialises: 10 M new groups cost ~15 ms; one per wave and iteration was still 17 K atomics on one word for
  // 1.1 M new groups — 0.2 ms of the 0.49 ms this kernel took in Q3, r03)
  __shared__ uint32_t wg_new;
  if (threadIdx.x == 0) wg_new = 0;
  __syncthreads();
  uint32_t my_new = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    bool claimed = false;
    {
      const uint64_t* r = rows_in + i * L.W;
      const uint64_t hw = probe_word(r[L.hash_word], hash_mask);
      uint64_t pos = home_slot(hw, cap);
      uint32_t found = GB_INVALID_SLOT;
      for (int64_t step = 0; step < cap; ++step) {
        unsigned long long cur = __hip_atomic_load((unsigned long long*)&slot_hash[pos], __ATOMIC_RELAXED,
                                                   __HIP_MEMORY_SCOPE_AGENT);
        if (cur == 0) {
          unsigned long long old = atomicCAS((unsigned long long*)&slot_hash[pos], 0ULL, (unsigned long long)hw);
          if (old == 0) {
            // this lane owns the new group: write keys, hash and identity states
            uint64_t* d = rows + pos * L.W;
            write_group_keys(L, r, d, arena, ctrl);
            d[L.hash_word] = r[L.hash_word];
            for (int a = 0; a < L.naggs; ++a) gb_state_identity(L, a, d + L.agg_off[a]);
            claimed = true;
            found = (uint32_t)pos;
            break;
          }
          cur = old;
        }
        if (cur == hw) {
          found = (uint32_t)pos;
          break;
        }
        pos = (pos + 1) & cmask;
      }
      if (found == GB_INVALID_SLOT) atomicOr((unsigned long long*)&ctrl[1], 1ULL);
      gid[i] = found;
    }
    my_new += claimed;
  }
  my_new = (uint32_t)wave_sum_u64(my_new);
  if (lane_id() == 0 && my_new) atomicAdd(&wg_new, my_new);
  __syncthreads();
  if (threadIdx.x == 0 && wg_new) atomicAdd((unsigned long long*)&ctrl[0], (unsigned long long)wg_new);
}

// ---------------------------------------------------------------------------
// accumulate — direct atomics (many groups)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gb_accum_kernel(GbLayout L, const uint64_t* rows_in, int64_t n,
                                                       uint64_t* rows, const uint32_t* gid,
                                                       uint32_t* retry, uint64_t* ctrl, DevCount dc, const uint8_t* arena) {
  n = dev_rows(dc, n);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const uint64_t* r = rows_in + i * L.W;
    uint32_t pos = gid[i];
    uint64_t* d = rows + (uint64_t)pos * L.W;
    if (!keys_equal(L, r, d, arena)) {
      unsigned long long k = atomicAdd((unsigned long long*)&ctrl[2], 1ULL);
      retry[k] = (uint32_t)i;
      continue;
    }
    for (int a = 0; a < L.naggs; ++a) gb_atomic_merge(L, a, d + L.agg_off[a], r + L.agg_off[a]);
  }
}

// ---------------------------------------------------------------------------
// accumulate — few groups: lanes of a wave that hit the same slot are combined
// with shuffles first, one lane issues the atomics (guide §6 G12).
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint64_t wave_min_u64(uint64_t v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    uint64_t o = __shfl_xor(v, off, 64);
    v = o < v ? o : v;
  }
  return v;
}
__device__ __forceinline__ uint64_t wave_max_u64(uint64_t v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    uint64_t o = __shfl_xor(v, off, 64);
    v = o > v ? o : v;
  }
  return v;
}
__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

__global__ __launch_bounds__(256) void gb_accum_lowcard_kernel(GbLayout L, const uint64_t* rows_in,
                                                               int64_t n, uint64_t* rows,
                                                               const uint32_t* gid, uint32_t* retry,
                                                               uint64_t* ctrl, DevCount dc, const uint8_t* arena) {
  n = dev_rows(dc, n);
  const int64_t n_pad = (n + 63) & ~63LL;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_pad;
       i += (int64_t)gridDim.x * blockDim.x) {
    bool active = i < n;
    const uint64_t* r = rows_in + (active ? i : 0) * L.W;
    uint32_t pos = active ? gid[i] : GB_INVALID_SLOT;
    if (active) {
      const uint64_t* d = rows + (uint64_t)pos * L.W;
      if (!keys_equal(L, r, d, arena)) {
        unsigned long long k = atomicAdd((unsigned long long*)&ctrl[2], 1ULL);
        retry[k] = (uint32_t)i;
        active = false;
      }
    }
    uint64_t todo = __ballot(active);
    while (todo) {
      int leader = __ffsll((long long)todo) - 1;
      uint32_t lpos = __shfl(pos, leader, 64);
      bool mine = active && pos == lpos;
      uint64_t m = __ballot(mine);
      uint64_t* d = rows + (uint64_t)lpos * L.W;
      for (int a = 0; a < L.naggs; ++a) {
        const uint64_t* v = r + L.agg_off[a];
        uint64_t out[GB_MAX_STATE_WORDS] = {0, 0, 0, 0};
        if (gb_minmax_str(L, a)) {   // no word-wise reduction exists for strings: every row of the group takes the state's lock in turn
          if (mine) gb_minmax_str_locked(L.agg_kind[a] == DBHIP_AGG_MIN, d + L.agg_off[a], v);
          continue;
        }
        if (gb_sum256(L, a) || gb_minmax256(L, a)) {   // five-word states: every row merges its own words (the wave reduction below is four words wide)
          if (mine) gb_atomic_merge(L, a, d + L.agg_off[a], v);
          continue;
        }
        switch (L.agg_kind[a]) {
          case DBHIP_AGG_COUNT:
            out[0] = wave_sum_u64(mine ? v[0] : 0);
            break;
          case DBHIP_AGG_SUM:
            if (L.agg_flag[a]) out[L.agg_flag[a]] = wave_max_u64(mine ? v[L.agg_flag[a]] : 0);
            if (L.agg_words[a] - (L.agg_flag[a] ? 1 : 0) == 3) {
              u128 t = mine ? (((u128)v[1] << 64) | v[0]) : (u128)0;
              uint64_t e = mine ? v[2] : 0;
#pragma unroll
              for (int off = 32; off >= 1; off >>= 1) {
                uint64_t olo = __shfl_xor((uint64_t)t, off, 64), ohi = __shfl_xor((uint64_t)(t >> 64), off, 64);
                uint64_t oe = __shfl_xor(e, off, 64);
                u128 o = ((u128)ohi << 64) | olo;
                u128 r = t + o;
                e += oe + (r < t ? 1 : 0);
                t = r;
              }
              out[0] = (uint64_t)t;
              out[1] = (uint64_t)(t >> 64);
              out[2] = e;
            } else if (L.agg_type[a] == DBHIP_T_F32 || L.agg_type[a] == DBHIP_T_F64) {
              out[0] = (uint64_t)__double_as_longlong(
                  wave_sum_f64(mine ? __longlong_as_double((long long)v[0]) : 0.0));
            } else {
              out[0] = wave_sum_u64(mine ? v[0] : 0);
            }
            break;
          case DBHIP_AGG_MIN:
            out[0] = wave_min_u64((mine && v[1]) ? v[0] : ~0ULL);
            out[1] = wave_max_u64(mine ? v[1] : 0);
            if (L.agg_words[a] == 3) out[2] = wave_min_u64((mine && v[1] && v[0] == out[0]) ? v[2] : ~0ULL);   // low word among the rows that hold the best high word
            break;
          default:
            out[0] = wave_max_u64((mine && v[1]) ? v[0] : 0ULL);
            out[1] = wave_max_u64(mine ? v[1] : 0);
            if (L.agg_words[a] == 3) out[2] = wave_max_u64((mine && v[1] && v[0] == out[0]) ? v[2] : 0ULL);
            break;
        }
        if (lane_id() == leader) gb_atomic_merge(L, a, d + L.agg_off[a], out);
      }
      todo &= ~m;
    }
  }
}

// ---------------------------------------------------------------------------
// retry — serial continuation of the probe for true hash collisions
// ---------------------------------------------------------------------------
__global__ void gb_retry_kernel(GbLayout L, const uint64_t* rows_in, uint64_t* slot_hash, uint64_t* rows,
                                int64_t cap, uint64_t hash_mask, const uint32_t* gid,
                                const uint32_t* retry, uint64_t* ctrl, uint8_t* arena) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const uint64_t cmask = (uint64_t)cap - 1;
  const uint64_t nretry = ctrl[2];
  for (uint64_t t = 0; t < nretry; ++t) {
    const uint32_t i = retry[t];
    const uint64_t* r = rows_in + (uint64_t)i * L.W;
    const uint64_t hw = probe_word(r[L.hash_word], hash_mask);
    uint64_t pos = ((uint64_t)gid[i] + 1) & cmask;
    bool done = false;
    for (int64_t step = 0; step < cap && !done; ++step) {
      uint64_t cur = slot_hash[pos];
      uint64_t* d = rows + pos * L.W;
      if (cur == 0) {
        if ((int64_t)(ctrl[0] + 1) * 135 > cap * 100) break;  // would exceed the load factor
        slot_hash[pos] = hw;
        write_group_keys(L, r, d, arena, ctrl);
        d[L.hash_word] = r[L.hash_word];
        for (int a = 0; a < L.naggs; ++a) gb_state_identity(L, a, d + L.agg_off[a]);
        ctrl[0] += 1;
        cur = hw;
      }
      if (cur == hw && keys_equal(L, r, d, arena)) {
        for (int a = 0; a < L.naggs; ++a) gb_atomic_merge(L, a, d + L.agg_off[a], r + L.agg_off[a]);
        done = true;
      }
      pos = (pos + 1) & cmask;
    }
    if (!done) ctrl[1] |= 2;  // table full inside retry: host grows and replays the leftovers
  }
}

// ---------------------------------------------------------------------------
// rehash (grow)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gb_rehash_kernel(GbLayout L, const uint64_t* old_hash,
                                                        const uint64_t* old_rows, int64_t old_cap,
                                                        uint64_t* new_hash, uint64_t* new_rows,
                                                        int64_t new_cap, uint64_t hash_mask) {
  const uint64_t cmask = (uint64_t)new_cap - 1;
  for (int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s < old_cap;
       s += (int64_t)gridDim.x * blockDim.x) {
    uint64_t hw = old_hash[s];
    if (hw == 0) continue;
    const uint64_t* r = old_rows + s * L.W;
    uint64_t pos = home_slot(probe_word(r[L.hash_word], hash_mask), new_cap);
    // all old entries are distinct groups: take the first EMPTY slot
    while (true) {
      unsigned long long old = atomicCAS((unsigned long long*)&new_hash[pos], 0ULL, (unsigned long long)hw);
      if (old == 0) break;
      pos = (pos + 1) & cmask;
    }
    uint64_t* d = new_rows + pos * L.W;
    for (int k = 0; k < L.W; ++k) d[k] = r[k];
  }
}

// ---------------------------------------------------------------------------
// flush
// ---------------------------------------------------------------------------
// One returning atomic per 2048 slots (a workgroup counts its chunk first): a wave-level atomic per 64 slots serialised on the one
// counter word — 131 K returning atomics for Q3's 8 M-slot table took 1.5 of the kernel's 1.6 ms (r03).
__global__ __launch_bounds__(256) void gb_flush_kernel(GbLayout L, const uint64_t* slot_hash,
                                                       const uint64_t* rows, int64_t cap,
                                                       uint64_t* out_rows, int64_t max_rows, uint64_t* ctrl) {
  __shared__ uint32_t wtot[4];
  __shared__ unsigned long long base_s;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t nchunks = (cap + 2047) / 2048;
  for (int64_t c = blockIdx.x; c < nchunks; c += gridDim.x) {
    // thread t owns the 8 CONSECUTIVE slots c * 2048 + 8 t .. + 7 (one 64-byte read of slot_hash per thread)
    const int64_t s0 = c * 2048 + (int64_t)tid * 8;
    uint32_t occ = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if (s0 + k < cap && slot_hash[s0 + k] != 0) occ |= 1u << k;
    const uint32_t mine = (uint32_t)__popc(occ);
    uint32_t incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(incl, d, 64); if (lane >= d) incl += o; }
    if (lane == 63) wtot[wave] = incl;
    __syncthreads();
    uint32_t wbase = 0, all = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) { if (w < wave) wbase += wtot[w]; all += wtot[w]; }
    if (tid == 0 && all) base_s = atomicAdd((unsigned long long*)&ctrl[4], (unsigned long long)all);
    __syncthreads();
    if (all) {
      uint64_t idx = base_s + wbase + incl - mine;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        if (!((occ >> k) & 1)) continue;
        if ((int64_t)idx < max_rows) {
          const uint64_t* r = rows + (s0 + k) * L.W;
          uint64_t* d = out_rows + idx * L.W;
          for (int q = 0; q < L.W; ++q) d[q] = r[q];
        }
        ++idx;
      }
    }
    __syncthreads();
  }
}

// Fixed-size exchange block (multi-GPU partial-state exchange): row 0 of a block is its header
// (word 0 = number of rows that follow, ~0 = the table held more than max_rows groups), rows 1.. are serialized rows.
__global__ __launch_bounds__(64) void gb_block_header_kernel(uint64_t* block, int W, int64_t max_rows, const uint64_t* ctrl) {
  const int t = threadIdx.x;
  if (t < W) block[t] = t == 0 ? ((int64_t)ctrl[4] > max_rows ? ~0ULL : ctrl[4]) : 0;
}

// one workgroup per source block: append its rows behind those of the earlier blocks (`skip` = the caller's own block)
__global__ __launch_bounds__(256) void gb_compact_blocks_kernel(const uint64_t* __restrict__ blocks, int64_t stride_words, int W,
                                                                int skip, uint64_t* __restrict__ out) {
  const int b = blockIdx.x;
  if (b == skip) return;
  const uint64_t cnt = blocks[(int64_t)b * stride_words];
  uint64_t off = 0;
  for (int p = 0; p < b; ++p)
    if (p != skip) off += blocks[(int64_t)p * stride_words];
  const uint64_t* src = blocks + (int64_t)b * stride_words + W;
  uint64_t* dst = out + off * W;
  for (uint64_t i = threadIdx.x; i < cnt * (uint64_t)W; i += blockDim.x) dst[i] = src[i];
}

struct ResultPtrs {
  void* keys[GB_MAX_KEYS];
  uint32_t* key_validity[GB_MAX_KEYS];
  void* aggs[GB_MAX_AGGS];
  uint32_t* agg_validity[GB_MAX_AGGS];   // nullable-argument SUM / MIN / MAX: bit = the group saw a non-NULL row
  uint64_t* hashes;
  const uint8_t* arena;                  // min / max over String: a long value's state holds the ADDRESS of its bytes inside the arena
};

// DecimalSumState<true, i256>::add: outside [DECIMAL_MIN, DECIMAL_MAX] (precision 76) is an Overflow error — decided on the exact 320-bit
// total s[0..5): the fifth word must be the sign extension and |total| <= 10^76 - 1
__device__ __forceinline__ bool gb_sum256_out_of_range(const uint64_t* s) {
  const bool neg = (s[3] >> 63) != 0;
  if (s[4] != (neg ? ~0ULL : 0ULL)) return true;
  uint64_t m[4] = {s[0], s[1], s[2], s[3]};
  if (neg) {   // magnitude
    uint64_t c = 1;
    for (int q = 0; q < 4; ++q) { const uint64_t t = ~m[q] + c; c = (c && t == 0) ? 1 : 0; m[q] = t; }
  }
  // 10^76 - 1 = 0x161BCCA7119915B5_0764B4ABE8652979_7775A5F171950FFF_FFFFFFFFFFFFFFFF (little-endian words)
  const uint64_t mx[4] = {0xFFFFFFFFFFFFFFFFULL, 0x7775A5F171950FFFULL, 0x0764B4ABE8652979ULL, 0x161BCCA7119915B5ULL};
  for (int q = 3; q >= 0; --q)
    if (m[q] != mx[q]) return m[q] > mx[q];
  return false;
}

// rows -> result columns (merge_result, aggregate_hashtable.rs:382-408)
__global__ __launch_bounds__(256) void gb_result_kernel(GbLayout L, const uint64_t* rows_out, int64_t n,
                                                        ResultPtrs P, uint64_t* ctrl) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const uint64_t* r = rows_out + i * L.W;
    uint64_t vmask = L.validity_word >= 0 ? r[L.validity_word] : ~0ULL;
    for (int k = 0; k < L.nkeys; ++k) {
      uint64_t w0 = r[L.key_off[k]];
      void* o = P.keys[k];
      if (o) {
        switch (L.key_type[k]) {
          case DBHIP_T_BOOL: case DBHIP_T_I8: case DBHIP_T_U8: ((uint8_t*)o)[i] = (uint8_t)w0; break;
          case DBHIP_T_I16: case DBHIP_T_U16: ((uint16_t*)o)[i] = (uint16_t)w0; break;
          case DBHIP_T_I32: case DBHIP_T_U32: case DBHIP_T_F32: case DBHIP_T_DATE:
            ((uint32_t*)o)[i] = (uint32_t)w0; break;
          case DBHIP_T_DEC256:
            for (int q = 0; q < 4; ++q) ((uint64_t*)o)[4 * i + q] = r[L.key_off[k] + q];
            break;
          case DBHIP_T_DEC128: case DBHIP_T_STRING: {
            uint64_t w1 = r[L.key_off[k] + 1];
            if (L.key_type[k] == DBHIP_T_STRING) {
              // words -> 16-byte view: {len, bytes[12]} inline, or {len, prefix, buffer 0, offset} into the table's arena
              uint32_t* v = (uint32_t*)o + 4 * i;
              if ((uint32_t)w0 > 12) {
                if (w1 >> 32) atomicOr((unsigned long long*)&ctrl[3], 8ULL);   // a view's offset is 32 bits
                v[0] = (uint32_t)w0; v[1] = (uint32_t)(w0 >> 32); v[2] = 0; v[3] = (uint32_t)w1;
              } else {
                v[0] = (uint32_t)w0; v[1] = (uint32_t)(w0 >> 32); v[2] = (uint32_t)w1; v[3] = (uint32_t)(w1 >> 32);
              }
            } else {
              ((uint64_t*)o)[2 * i] = w0;
              ((uint64_t*)o)[2 * i + 1] = w1;
            }
          } break;
          default: ((uint64_t*)o)[i] = w0; break;
        }
      }
      if (P.key_validity[k] && ((vmask >> k) & 1)) atomicOr(&P.key_validity[k][i >> 5], 1u << (i & 31));
    }
    if (P.hashes) P.hashes[i] = r[L.hash_word];
    for (int a = 0; a < L.naggs; ++a) {
      const uint64_t* s = r + L.agg_off[a];
      void* o = P.aggs[a];
      if (P.agg_validity[a]) {
        // AggregateNullUnaryAdaptor<true>::merge_result (aggregate_null_adaptor.rs): NULL unless the flag is set
        bool seen = true;
        if (L.agg_kind[a] == DBHIP_AGG_SUM) seen = L.agg_flag[a] ? s[L.agg_flag[a]] != 0 : true;
        else if (L.agg_kind[a] == DBHIP_AGG_MIN || L.agg_kind[a] == DBHIP_AGG_MAX) seen = s[1] != 0;
        if (seen) atomicOr(&P.agg_validity[a][i >> 5], 1u << (i & 31));
      }
      if (!o) continue;
      switch (L.agg_kind[a]) {
        case DBHIP_AGG_COUNT:
          ((uint64_t*)o)[i] = s[0];
          break;
        case DBHIP_AGG_SUM:
          if (L.agg_type[a] == DBHIP_T_DEC256) {
            if (gb_sum256_out_of_range(s)) atomicOr((unsigned long long*)&ctrl[3], 1ULL);
            for (int q = 0; q < 4; ++q) ((uint64_t*)o)[4 * i + q] = s[q];
            break;
          }
          if (L.agg_words[a] - (L.agg_flag[a] ? 1 : 0) == 3) {
            i128 v = (i128)(((u128)s[1] << 64) | s[0]);
            // DecimalSumState<true,_>::add (aggregate_sum.rs:203-216): outside
            // [DECIMAL_MIN, DECIMAL_MAX] is an Overflow error. Decided on the exact
            // 192-bit total: ext must be the sign extension of the low 128 bits.
            i128 mx = pow10_i128(38) - 1;
            bool fits128 = s[2] == ((s[1] >> 63) ? ~0ULL : 0ULL);
            if (L.agg_precision[a] > 18 && (!fits128 || v > mx || v < -mx)) atomicOr((unsigned long long*)&ctrl[3], 1ULL);
            ((uint64_t*)o)[2 * i] = s[0];
            ((uint64_t*)o)[2 * i + 1] = s[1];
          } else {
            ((uint64_t*)o)[i] = s[0];
          }
          break;
        default: {  // MIN / MAX (no value seen: the type's default, MinMaxAnyState::merge_result push_default)
          if (L.agg_type[a] == DBHIP_T_STRING) {
            // -> 16-byte view: {len, bytes[12]} inline, or {len, prefix, buffer 0, offset into the table's arena} (dbhip_groupby_arena)
            uint32_t* v = (uint32_t*)o + 4 * i;
            const uint32_t len = s[1] ? (uint32_t)s[0] : 0;
            if (len > 12) {
              const uint64_t off = s[2] - (uint64_t)P.arena;
              if (off >> 32) atomicOr((unsigned long long*)&ctrl[3], 8ULL);   // a view's offset is 32 bits
              v[0] = len; v[1] = (uint32_t)(s[0] >> 32); v[2] = 0; v[3] = (uint32_t)off;
            } else {
              v[0] = len; v[1] = s[1] ? (uint32_t)(s[0] >> 32) : 0; v[2] = s[1] ? (uint32_t)s[2] : 0; v[3] = s[1] ? (uint32_t)(s[2] >> 32) : 0;
            }
            break;
          }
          if (L.agg_type[a] == DBHIP_T_DEC256) {
            uint64_t* q = (uint64_t*)o + 4 * i;
            q[0] = s[1] ? s[4] : 0; q[1] = s[1] ? s[3] : 0; q[2] = s[1] ? s[2] : 0; q[3] = s[1] ? (s[0] ^ (1ULL << 63)) : 0;
            break;
          }
          if (L.agg_words[a] == 3) {   // Decimal128
            ((uint64_t*)o)[2 * i] = s[1] ? s[2] : 0;
            ((uint64_t*)o)[2 * i + 1] = s[1] ? (s[0] ^ (1ULL << 63)) : 0;
            break;
          }
          uint64_t raw = s[1] ? ord_decode(s[0], L.agg_type[a]) : 0;
          switch (L.agg_type[a]) {
            case DBHIP_T_I8: case DBHIP_T_U8: case DBHIP_T_BOOL: ((uint8_t*)o)[i] = (uint8_t)raw; break;
            case DBHIP_T_I16: case DBHIP_T_U16: ((uint16_t*)o)[i] = (uint16_t)raw; break;
            case DBHIP_T_I32: case DBHIP_T_U32: case DBHIP_T_F32: case DBHIP_T_DATE:
              ((uint32_t*)o)[i] = (uint32_t)raw; break;
            default: ((uint64_t*)o)[i] = raw; break;
          }
        } break;
      }
    }
  }
}

// ---------------------------------------------------------------------------
// a12: hash partitioning of the table's group rows for the exchange / final merge — the device twin of
// Payload::scan_hash_partition_transfer (payload.rs:548-589: bucket = group hash % bucket count) and
// PartitionedPayload::repartition. Every occupied slot's row (keys, hash, states: the unit of exchange) is copied
// to its bucket's output region; lanes of a wave that share a bucket take ONE cursor atomic together.
//   blocks mode  (bucket_base == nullptr): bucket b -> out + b * stride_words, rows from row 1 on (row 0 = header),
//                at most max_rows rows are written, the cursor keeps counting (overflow is seen in the header)
//   ranges mode  (bucket_base != nullptr): bucket b -> rows [bucket_base[b], bucket_base[b + 1]) of `out`
//   count only   (out == nullptr)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gb_partition_rows_kernel(GbLayout L, const uint64_t* slot_hash, const uint64_t* rows,
                                                                int64_t cap, uint32_t n_buckets, int64_t max_rows,
                                                                int64_t stride_words, const uint64_t* bucket_base,
                                                                uint64_t* out, unsigned long long* cursor) {
  const int64_t cap_pad = (cap + 63) & ~63LL;
  for (int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s < cap_pad; s += (int64_t)gridDim.x * blockDim.x) {
    const bool occ = s < cap && slot_hash[s] != 0;
    const uint64_t* r = rows + s * L.W;
    const uint32_t bucket = occ ? (uint32_t)(r[L.hash_word] % (uint64_t)n_buckets) : 0xFFFFFFFFu;
    uint64_t todo = __ballot(occ);
    while (todo) {
      const int leader = __ffsll((long long)todo) - 1;
      const uint32_t lb = __shfl(bucket, leader, 64);
      const bool mine = occ && bucket == lb;
      const uint64_t m = __ballot(mine);
      unsigned long long b0 = 0;
      if (lane_id() == leader) b0 = atomicAdd(&cursor[lb], (unsigned long long)__popcll(m));
      b0 = __shfl(b0, leader, 64);
      if (mine && out) {
        const uint64_t idx = b0 + __popcll(m & ((1ULL << lane_id()) - 1));
        uint64_t* d = nullptr;
        if (bucket_base) d = out + (bucket_base[lb] + idx) * L.W;
        else if ((int64_t)idx < max_rows) d = out + (int64_t)lb * stride_words + (idx + 1) * L.W;
        if (d) for (int k = 0; k < L.W; ++k) d[k] = r[k];
      }
      todo &= ~m;
    }
  }
}

// headers of the n_buckets blocks: word 0 = rows that follow (~0: more than max_rows), word 1 = 1 when ANY block of this
// sender overflowed — every receiver of an all-to-all gets one block from every sender, so all ranks see the same
// flags and take the variable-length path together
__global__ __launch_bounds__(256) void gb_partition_headers_kernel(uint64_t* blocks, int W, int64_t stride_words, int64_t max_rows,
                                                                   uint32_t n_buckets, const unsigned long long* cursor) {
  __shared__ int any;
  if (threadIdx.x == 0) any = 0;
  __syncthreads();
  for (uint32_t b = threadIdx.x; b < n_buckets; b += blockDim.x)
    if ((int64_t)cursor[b] > max_rows) atomicOr(&any, 1);
  __syncthreads();
  for (uint32_t b = threadIdx.x; b < n_buckets; b += blockDim.x) {
    uint64_t* h = blocks + (int64_t)b * stride_words;
    for (int k = 0; k < W; ++k) h[k] = 0;
    h[0] = (int64_t)cursor[b] > max_rows ? ~0ULL : (uint64_t)cursor[b];
    h[1] = (uint64_t)any;
  }
}

// ---------------------------------------------------------------------------
// §8f-1: the serialized-state block of Payload::aggregate_flush (payload_flush.rs:151-181): per aggregate the fields of
// its serialize_type(), then the group columns.
//   count                        (UInt64)                                            aggregate_count.rs:170-186
//   sum  -> its result type      (Int64 / UInt64 / Float64 / Decimal)                aggregate_sum.rs:155-168,281-298
//   min / max                    (Boolean has-value, T value; default value if none) aggregate_min_max_any.rs:315-346
//   nullable argument (sum/min/max): the nested fields + a trailing Boolean flag      aggregate_null_adaptor.rs:508-540
// Field columns are flattened in aggregate order; Boolean fields are LSB-first bitmaps.
// ---------------------------------------------------------------------------
struct StateFieldPtrs {
  void* f[GB_MAX_AGGS * 3];
};

__device__ __forceinline__ void store_typed(void* o, int64_t i, int type, uint64_t raw) {
  switch (type) {
    case DBHIP_T_I8: case DBHIP_T_U8: ((uint8_t*)o)[i] = (uint8_t)raw; break;
    case DBHIP_T_I16: case DBHIP_T_U16: ((uint16_t*)o)[i] = (uint16_t)raw; break;
    case DBHIP_T_I32: case DBHIP_T_U32: case DBHIP_T_F32: case DBHIP_T_DATE: ((uint32_t*)o)[i] = (uint32_t)raw; break;
    default: ((uint64_t*)o)[i] = raw; break;
  }
}
__device__ __forceinline__ void set_bit32(void* bm, int64_t i) { atomicOr((uint32_t*)bm + (i >> 5), 1u << (i & 31)); }

// serialized rows -> state field columns (the key columns are written by gb_result_kernel)
__global__ __launch_bounds__(256) void gb_state_fields_kernel(GbLayout L, const uint64_t* rows_out, int64_t n, StateFieldPtrs P,
                                                              uint64_t* ctrl, const uint8_t* arena) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const uint64_t* r = rows_out + i * L.W;
    int f = 0;
    for (int a = 0; a < L.naggs; ++a) {
      const uint64_t* s = r + L.agg_off[a];
      switch (L.agg_kind[a]) {
        case DBHIP_AGG_COUNT:
          if (P.f[f]) ((uint64_t*)P.f[f])[i] = s[0];
          ++f;
          break;
        case DBHIP_AGG_SUM: {
          const int fw = L.agg_flag[a];
          if (L.agg_type[a] == DBHIP_T_DEC256) {
            // DecimalSumState<true, i256>: the state IS the running total; outside +-(10^76 - 1) is the Overflow error of add()
            if (gb_sum256_out_of_range(s)) atomicOr((unsigned long long*)&ctrl[3], 1ULL);
            if (P.f[f]) for (int q = 0; q < 4; ++q) ((uint64_t*)P.f[f])[4 * i + q] = s[q];
          } else if (L.agg_words[a] - (fw ? 1 : 0) == 3) {
            const i128 v = (i128)(((u128)s[1] << 64) | s[0]);
            const i128 mx = pow10_i128(38) - 1;
            const bool fits128 = s[2] == ((s[1] >> 63) ? ~0ULL : 0ULL);
            if (L.agg_precision[a] > 18 && (!fits128 || v > mx || v < -mx)) atomicOr((unsigned long long*)&ctrl[3], 1ULL);
            if (P.f[f]) { ((uint64_t*)P.f[f])[2 * i] = s[0]; ((uint64_t*)P.f[f])[2 * i + 1] = s[1]; }
          } else if (P.f[f]) {
            ((uint64_t*)P.f[f])[i] = s[0];
          }
          ++f;
          if (fw) { if (P.f[f] && s[fw]) set_bit32(P.f[f], i); ++f; }
        } break;
        default: {  // MIN / MAX
          if (P.f[f] && s[1]) set_bit32(P.f[f], i);
          ++f;
          if (P.f[f]) {
            if (L.agg_type[a] == DBHIP_T_STRING) {
              // the value column of the Nullable(String) state: a 16-byte view, long strings by offset into the table's arena (buffer 0)
              uint32_t* v = (uint32_t*)P.f[f] + 4 * i;
              const uint32_t len = s[1] ? (uint32_t)s[0] : 0;
              if (len > 12) {
                const uint64_t off = s[2] - (uint64_t)arena;
                if (off >> 32) atomicOr((unsigned long long*)&ctrl[3], 8ULL);
                v[0] = len; v[1] = (uint32_t)(s[0] >> 32); v[2] = 0; v[3] = (uint32_t)off;
              } else {
                v[0] = len; v[1] = s[1] ? (uint32_t)(s[0] >> 32) : 0; v[2] = s[1] ? (uint32_t)s[2] : 0; v[3] = s[1] ? (uint32_t)(s[2] >> 32) : 0;
              }
            } else if (L.agg_type[a] == DBHIP_T_DEC256) {
              uint64_t* q = (uint64_t*)P.f[f] + 4 * i;
              q[0] = s[1] ? s[4] : 0; q[1] = s[1] ? s[3] : 0; q[2] = s[1] ? s[2] : 0; q[3] = s[1] ? (s[0] ^ (1ULL << 63)) : 0;
            }
            else if (L.agg_words[a] == 3) { ((uint64_t*)P.f[f])[2 * i] = s[1] ? s[2] : 0; ((uint64_t*)P.f[f])[2 * i + 1] = s[1] ? (s[0] ^ (1ULL << 63)) : 0; }
            else store_typed(P.f[f], i, L.agg_type[a], s[1] ? ord_decode(s[0], L.agg_type[a]) : 0);
          }
          ++f;
          if (L.agg_nullable[a]) { if (P.f[f] && s[1]) set_bit32(P.f[f], i); ++f; }
        } break;
      }
    }
  }
}

struct StateFieldCols {
  GbCol f[GB_MAX_AGGS * 3];
};

// state field columns -> the state words of rows_in (the keys were serialized by gb_serialize_kernel with no
// aggregate arguments): what TransformDeserializer + AggregateFunction::batch_merge consume
// (aggregator/serde/transform_deserializer.rs; batch_merge of each function, cited above)
__global__ __launch_bounds__(256) void gb_states_from_fields_kernel(GbLayout L, StateFieldCols F, int64_t n, uint64_t* rows_in, uint64_t* ctrl) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    uint64_t* r = rows_in + i * L.W;
    int f = 0;
    for (int a = 0; a < L.naggs; ++a) {
      uint64_t* s = r + L.agg_off[a];
      uint64_t w[2];
      bool valid;
      switch (L.agg_kind[a]) {
        case DBHIP_AGG_COUNT:
          gb_load_words(F.f[f], i, w, &valid);
          s[0] = w[0];
          ++f;
          break;
        case DBHIP_AGG_SUM: {
          const int fw = L.agg_flag[a];
          if (L.agg_type[a] == DBHIP_T_DEC256) {
            const uint64_t* p = (const uint64_t*)F.f[f].data + 4 * (F.f[f].is_scalar ? 0 : i);
            ++f;
            bool seen256 = true;
            if (fw) { seen256 = bit_get((const uint8_t*)F.f[f].data, F.f[f].is_scalar ? 0 : i); ++f; }
            for (int q = 0; q < 4; ++q) s[q] = seen256 ? p[q] : 0;
            s[4] = (seen256 && (p[3] >> 63)) ? ~0ULL : 0;
            if (fw) s[fw] = seen256 ? 1 : 0;
            break;
          }
          gb_load_words(F.f[f], i, w, &valid);
          ++f;
          bool seen = true;
          if (fw) { seen = bit_get((const uint8_t*)F.f[f].data, F.f[f].is_scalar ? 0 : i); ++f; }
          // a state whose flag is clear contributes nothing (the adaptor's batch_merge filters on the flag)
          s[0] = seen ? w[0] : 0;
          if (L.agg_words[a] - (fw ? 1 : 0) == 3) {
            s[1] = seen ? w[1] : 0;
            s[2] = (seen && (w[1] >> 63)) ? ~0ULL : 0;
          }
          if (fw) s[fw] = seen ? 1 : 0;
        } break;
        default: {
          bool has = bit_get((const uint8_t*)F.f[f].data, F.f[f].is_scalar ? 0 : i);
          ++f;
          const GbCol& vc = F.f[f];
          ++f;
          if (L.agg_nullable[a]) { has = has && bit_get((const uint8_t*)F.f[f].data, F.f[f].is_scalar ? 0 : i); ++f; }
          if (L.agg_type[a] == DBHIP_T_STRING) {
            // the Nullable(String) state column: short values as the canonical inline words, long ones as (len | prefix, ADDRESS)
            const int64_t j = vc.is_scalar ? 0 : i;
            const uint32_t* v = (const uint32_t*)vc.data + 4 * j;
            const uint32_t len = v[0];
            uint64_t ww[2] = {0, 0};
            if (has) {
              if (len <= 12) { bool vv; gb_load_words(vc, i, ww, &vv); }
              else if (vc.buffers) { ww[0] = ((uint64_t)v[1] << 32) | len; ww[1] = (uint64_t)((const uint8_t*)vc.buffers[v[2]] + v[3]); }
              else { has = false; atomicOr((unsigned long long*)&ctrl[3], 2ULL); }   // a long view without data buffers: the merge reports it
            }
            s[0] = has ? ww[0] : 0; s[2] = has ? ww[1] : 0;
          } else if (L.agg_type[a] == DBHIP_T_DEC256) {
            const uint64_t* p = (const uint64_t*)vc.data + 4 * (vc.is_scalar ? 0 : i);
            s[0] = p[3] ^ (1ULL << 63); s[2] = p[2]; s[3] = p[1]; s[4] = p[0];
          } else {
            gb_load_words(vc, i, w, &valid);
            if (L.agg_words[a] == 3) { s[0] = w[1] ^ (1ULL << 63); s[2] = w[0]; }
            else s[0] = ord_encode(w[0], L.agg_type[a]);
          }
          s[1] = has ? 1 : 0;
        } break;
      }
    }
  }
}

// fields of the serialized-state block for this layout, in order; returns their number
int state_fields(const GbLayout& L, int32_t* types, int32_t* agg_of) {
  int f = 0;
  for (int a = 0; a < L.naggs; ++a) {
    dbhip_agg_desc d = {L.agg_kind[a], L.agg_type[a], (uint8_t)L.agg_precision[a], (uint8_t)L.agg_scale[a], (uint8_t)L.agg_nullable[a], 0};
    int32_t rt = 0;
    uint8_t p, sc;
    (void)dbhip_groupby_result_type(&d, &rt, &p, &sc);
    auto put = [&](int t) { if (types) types[f] = t; if (agg_of) agg_of[f] = a; ++f; };
    switch (L.agg_kind[a]) {
      case DBHIP_AGG_COUNT: put(DBHIP_T_U64); break;
      case DBHIP_AGG_SUM: put(rt); if (L.agg_flag[a]) put(DBHIP_T_BOOL); break;
      default: put(DBHIP_T_BOOL); put(rt); if (L.agg_nullable[a]) put(DBHIP_T_BOOL); break;
    }
  }
  return f;
}

GbCol to_gbcol(const dbhip_col& c) {
  GbCol g;
  g.data = c.data; g.validity = c.validity; g.voff = c.validity_offset;
  g.buffers = c.buffers; g.type = c.type; g.is_scalar = c.is_scalar;
  return g;
}

bool key_type_ok(int t) { return t >= DBHIP_T_BOOL && t <= DBHIP_T_DEC256; }

int32_t build_layout(const int32_t* key_types, const uint8_t* key_nullable, int nkeys,
                     const dbhip_agg_desc* aggs, int naggs, GbLayout* L) {
  if (nkeys < 1 || nkeys > GB_MAX_KEYS || naggs < 0 || naggs > GB_MAX_AGGS) {
    set_error("groupby: %d keys / %d aggregates outside the supported range (1..%d / 0..%d)", nkeys, naggs,
              GB_MAX_KEYS, GB_MAX_AGGS);
    return DBHIP_ERR_INVALID;
  }
  memset(L, 0, sizeof(*L));
  L->nkeys = nkeys; L->naggs = naggs;
  int w = 0;
  bool any_nullable = false;
  for (int k = 0; k < nkeys; ++k) {
    if (!key_type_ok(key_types[k])) {
      set_error("groupby: unsupported key type %d", key_types[k]);
      return DBHIP_ERR_INVALID;
    }
    L->key_type[k] = key_types[k];
    L->key_off[k] = w;
    L->key_words[k] = key_types[k] == DBHIP_T_DEC256 ? 4 : ((key_types[k] == DBHIP_T_DEC128 || key_types[k] == DBHIP_T_STRING) ? 2 : 1);
    if (key_types[k] == DBHIP_T_STRING) L->str_w1_mask |= 1u << (w + 1);
    L->key_nullable[k] = key_nullable ? key_nullable[k] : 0;
    any_nullable |= L->key_nullable[k] != 0;
    w += L->key_words[k];
  }
  L->validity_word = any_nullable ? w++ : -1;
  L->nkey_words = w;
  L->hash_word = w++;
  for (int a = 0; a < naggs; ++a) {
    const dbhip_agg_desc& d = aggs[a];
    L->agg_kind[a] = d.kind; L->agg_type[a] = d.arg_type; L->agg_nullable[a] = d.arg_nullable;
    L->agg_precision[a] = d.arg_precision; L->agg_scale[a] = d.arg_scale;
    L->agg_off[a] = w;
    int words = 1;
    switch (d.kind) {
      case DBHIP_AGG_COUNT: break;
      case DBHIP_AGG_SUM:
        if (d.arg_type == DBHIP_T_DEC128) words = 3;
        else if (d.arg_type == DBHIP_T_DEC256) words = GB_SUM256_WORDS;   // exact 320-bit total (gb_device.h)
        else if (!(d.arg_type >= DBHIP_T_I8 && d.arg_type <= DBHIP_T_F64) && d.arg_type != DBHIP_T_DEC64) {
          set_error("groupby: sum() does not support type %d", d.arg_type);
          return DBHIP_ERR_INVALID;
        }
        if (d.arg_nullable) L->agg_flag[a] = words++;   // "seen a non-NULL row" (AggregateNullUnaryAdaptor<true>)
        break;
      case DBHIP_AGG_MIN: case DBHIP_AGG_MAX:
        if (!key_type_ok(d.arg_type)) {
          set_error("groupby: min/max on type %d stays on the CPU operator", d.arg_type);
          return DBHIP_ERR_UNSUPPORTED;
        }
        // (value, has) — Decimal128: (high word, has, low word); String: (len | prefix, has, tail or address of the bytes); Decimal256:
        // (top word, has, three lower words), gb_device.h
        words = d.arg_type == DBHIP_T_DEC256 ? GB_MM256_WORDS : (d.arg_type == DBHIP_T_DEC128 || d.arg_type == DBHIP_T_STRING) ? 3 : 2;
        break;
      default:
        set_error("groupby: unknown aggregate kind %d", d.kind);
        return DBHIP_ERR_INVALID;
    }
    L->agg_words[a] = words;
    w += words;
  }
  L->W = w;
  return DBHIP_OK;
}

}  // namespace
int32_t dbhip_groupby_build_layout_internal(const int32_t* key_types, const uint8_t* key_nullable, int nkeys, const dbhip_agg_desc* aggs,
                                            int naggs, GbLayout* L) {
  return build_layout(key_types, key_nullable, nkeys, aggs, naggs, L);
}
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
                            bool deferred = false);
int32_t merge_rows(dbhip_groupby* g, const uint64_t* rows_in, int64_t n, hipStream_t s,
                   const uint64_t* n_dev = nullptr, const uint64_t* abort_dev = nullptr) {
  int32_t rc = merge_rows_unpinned(g, rows_in, n, s, n_dev, abort_dev);
  if (rc == DBHIP_OK && n > 0) rc = pin_string_states(g, s);
  return rc;
}
// `deferred` (the pipelined fused aggregation): the caller has made sure that the table cannot outgrow its load factor whatever the
// rows hold; the three kernels are queued and NOTHING is read back — the table's count_host is stale until the caller's checkpoint
int32_t merge_rows_unpinned(dbhip_groupby* g, const uint64_t* rows_in, int64_t n, hipStream_t s,
                            const uint64_t* n_dev, const uint64_t* abort_dev, bool deferred) {
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
  if (deferred || (g->count_host + n) * 135 <= g->cap * 100) {
    DBHIP_CHECK(hipMemsetAsync(&g->ctrl[1], 0, 16, s));
    hipLaunchKernelGGL(gb_probe_kernel, dim3(grid), dim3(256), 0, s, g->L, cur_rows, cur_n, g->slot_hash, g->rows, g->cap,
                       g->hash_mask, g->gid, g->ctrl, dc, g->arena);
    // (the wave-combining kernel only where the table is known to hold a handful of groups: on an empty table the first
    // merge may bring 50 K groups, r02y: 0.11 ms there against 0.02 ms for the plain kernel)
    // (a Decimal128 min / max state is merged under a lock: always combine the rows of a wave first, one acquisition per wave and state)
    if (deferred || (g->count_host > 0 && g->count_host <= 32 && n <= 65536) || n <= 2048 || layout_has_wide_minmax(g->L))
      hipLaunchKernelGGL(gb_accum_lowcard_kernel, dim3(grid), dim3(256), 0, s, g->L, cur_rows, cur_n, g->rows, g->gid, g->retry,
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
int32_t dbhip_fagg_add_columns_internal(dbhip_groupby* g, const GbCols& C, int64_t row0, int64_t n, bool may_compile, hipStream_t s);  // k_fagg.hip
bool dbhip_fagg_last_refusal_is_pending_internal();
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

// Used by k_q1.hip: merge `n` device rows (table layout) produced by a fused kernel.
int32_t dbhip_groupby_merge_rows_internal(dbhip_groupby* g, const uint64_t* rows, int64_t n, hipStream_t s) {
  return merge_rows(g, rows, n, s);
}
// same, with the row count (and the producer's give-up flag) still on the device: `n_max` bounds the count
int32_t dbhip_groupby_merge_rows_dev_internal(dbhip_groupby* g, const uint64_t* rows, int64_t n_max, const uint64_t* n_dev,
                                              const uint64_t* abort_dev, hipStream_t s) {
  return merge_rows(g, rows, n_max, s, n_dev, abort_dev);
}
// allocate the merge scratch for up to n rows NOW (callers that queue a merge behind a running kernel: no hipMalloc may
// fall between the kernel's launch and the merge's launches)
int32_t dbhip_groupby_reserve_merge_internal(dbhip_groupby* g, int64_t n) {
  int32_t rc;
  if ((rc = ensure((void**)&g->gid, &g->gid_cap, (size_t)n * 4))) return rc;
  return ensure((void**)&g->retry, &g->retry_cap, (size_t)n * 4);
}
// the pipelined fused aggregation (k_fagg.hip): merge with nothing read back (see merge_rows_unpinned)
int32_t dbhip_groupby_merge_rows_deferred_internal(dbhip_groupby* g, const uint64_t* rows, int64_t n_max, const uint64_t* n_dev,
                                                   const uint64_t* abort_dev, hipStream_t s) {
  return merge_rows_unpinned(g, rows, n_max, s, n_dev, abort_dev, true);
}
// drains the stream, reads the exact number of groups and grows the table until `extra` more groups cannot push it past its load factor
int32_t dbhip_groupby_ensure_room_internal(dbhip_groupby* g, int64_t extra, hipStream_t s) {
  uint64_t cnt = 0;
  DBHIP_CHECK(hipMemcpyAsync(&cnt, &g->ctrl[0], 8, hipMemcpyDeviceToHost, s));
  DBHIP_CHECK(hipStreamSynchronize(s));
  g->count_host = (int64_t)cnt;
  while ((g->count_host + extra) * 135 > g->cap * 100) {
    const int32_t rc = grow(g, s);
    if (rc) return rc;
  }
  return DBHIP_OK;
}
uint64_t* dbhip_groupby_ctrl_internal(dbhip_groupby* g) { return g->ctrl; }
void dbhip_groupby_set_count_internal(dbhip_groupby* g, int64_t count) { g->count_host = count; }
void** dbhip_groupby_pipe_slot_internal(dbhip_groupby* g) { return &g->fa_pipe; }
int64_t dbhip_groupby_capacity_internal(dbhip_groupby* g) { return g->cap; }
int64_t dbhip_groupby_count_internal(dbhip_groupby* g) { return g->count_host; }
const GbLayout* dbhip_groupby_layout_internal(dbhip_groupby* g) { return &g->L; }

extern "C" {

int32_t dbhip_group_hash(const dbhip_col* cols, int32_t ncols, int64_t n, uint64_t* out_hashes,
                         void* stream) {
  DBHIP_REQUIRE(cols && ncols >= 1 && ncols <= GB_MAX_KEYS, "dbhip_group_hash: bad column list");
  if (n == 0) return DBHIP_OK;
  DBHIP_REQUIRE(out_hashes, "dbhip_group_hash: NULL out");
  HashCols hc;
  hc.n = ncols;
  for (int k = 0; k < ncols; ++k) {
    if (!key_type_ok(cols[k].type)) {
      set_error("dbhip_group_hash: unsupported type %d", cols[k].type);
      return DBHIP_ERR_INVALID;
    }
    hc.c[k] = to_gbcol(cols[k]);
  }
  hipStream_t s = resolve_stream(stream);
  unsigned long long* bad = (unsigned long long*)scratch(8, 2, s);
  if (!bad) return DBHIP_ERR_HIP;
  DBHIP_CHECK(hipMemsetAsync(bad, 0, 8, s));
  hipLaunchKernelGGL(group_hash_kernel, dim3(grid_for(ceil_div(n, 4), 256, 1024)), dim3(256), 0, s, hc, n, out_hashes, bad);
  DBHIP_LAUNCH_CHECK();
  return DBHIP_OK;
}

int32_t dbhip_groupby_create(const int32_t* key_types_host, const uint8_t* key_nullable_host,
                             int32_t nkeys, const dbhip_agg_desc* aggs_host, int32_t naggs,
                             int64_t initial_capacity, dbhip_groupby** out_host) {
  DBHIP_REQUIRE(out_host && key_types_host, "dbhip_groupby_create: NULL argument");
  dbhip_groupby* g = new (std::nothrow) dbhip_groupby();
  DBHIP_REQUIRE(g, "dbhip_groupby_create: out of host memory");
  memset(g, 0, sizeof(*g));
  int32_t rc = build_layout(key_types_host, key_nullable_host, nkeys, aggs_host, naggs, &g->L);
  if (rc) { delete g; return rc; }
  int64_t cap = 1024;
  while (cap < initial_capacity) cap <<= 1;
  g->hash_mask = ~0ULL;
  g->part_min_rows = 262144;
  g->hint_groups = initial_capacity;
  if (layout_has_wide_minmax(g->L)) { g->part_forbidden = 1; g->part_bits = -1; }   // row path only (see gb_minmax_wide_locked)
  hipStream_t s = resolve_stream(nullptr);
  if ((rc = alloc_table(g, cap, s))) { delete g; return rc; }
  hipError_t e = hipMalloc((void**)&g->ctrl, 128);   // [0..7] see above, [8] arena cursor, [9] long-string bytes of the current chunk
  if (e == hipSuccess) e = hipMemsetAsync(g->ctrl, 0, 128, s);
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  if (e != hipSuccess) {  // nothing half-built is handed out or leaked
    (void)dbhip_groupby_destroy(g);
    return hip_fail(e, "dbhip_groupby_create");
  }
  *out_host = g;
  return DBHIP_OK;
}

// test hook (not part of the drop-in surface): restrict the probe hash to `mask`
// so that distinct keys share a hash word and the collision path is exercised
// (the reference tests the same situation with hand-made tags, hash_index/index.rs:385-404).
int32_t dbhip_groupby_debug_set_hash_mask(dbhip_groupby* g, uint64_t mask) {
  DBHIP_REQUIRE(g && g->count_host == 0, "dbhip_groupby_debug_set_hash_mask: table must be empty");
  g->hash_mask = mask;
  return DBHIP_OK;
}

// test hook: force the radix-partitioned path with 2^bits partitions for every block size
// (bits = 0: back to adaptive; bits < 0: never partition)
int32_t dbhip_groupby_debug_set_partition_bits(dbhip_groupby* g, int32_t bits) {
  DBHIP_REQUIRE(g && bits <= PT_MAX_BITS, "dbhip_groupby_debug_set_partition_bits: bad argument");
  if (bits > 0) { g->part_bits = bits; g->part_min_rows = 1; g->part_forbidden = 0; }
  else if (bits == 0) { g->part_bits = 0; g->part_min_rows = 262144; g->part_forbidden = 0; }
  else { g->part_bits = -1; g->part_forbidden = 1; g->part_min_rows = 262144; }
  return DBHIP_OK;
}

// test hook: keep this table off (0) / on (1, the default) the compact-row kernels (gb_compact.h), so that both the generic and the
// compact kernels can be driven through the same cases
int32_t dbhip_groupby_debug_set_compact(dbhip_groupby* g, int32_t on) {
  DBHIP_REQUIRE(g, "dbhip_groupby_debug_set_compact: NULL argument");
  g->gbc_off = on ? 0 : 1;
  return DBHIP_OK;
}

int32_t dbhip_groupby_add_block(dbhip_groupby* g, const dbhip_col* keys, const dbhip_col* args,
                                int64_t n, void* stream) {
  return dbhip_groupby_add_block_filtered(g, keys, args, n, nullptr, 0, stream);
}

int32_t dbhip_groupby_add_block_filtered(dbhip_groupby* g, const dbhip_col* keys, const dbhip_col* args, int64_t n,
                                         const uint8_t* filter_bitmap, int64_t filter_bit_offset, void* stream) {
  GB_DRAIN(g, resolve_stream(stream));
  DBHIP_REQUIRE(g && keys, "dbhip_groupby_add_block: NULL argument");
  if (n == 0) return DBHIP_OK;
  hipStream_t s = resolve_stream(stream);
  GbCols C;
  memset(&C, 0, sizeof(C));
  C.filter = filter_bitmap;
  C.filter_off = filter_bit_offset;
  for (int k = 0; k < g->L.nkeys; ++k) {
    if (keys[k].type != g->L.key_type[k]) {
      set_error("dbhip_groupby_add_block: key %d has type %d, table expects %d", k, keys[k].type, g->L.key_type[k]);
      return DBHIP_ERR_INVALID;
    }
    if (keys[k].validity && !g->L.key_nullable[k]) {
      set_error("dbhip_groupby_add_block: key %d carries validity but was declared NOT NULL", k);
      return DBHIP_ERR_INVALID;
    }
    C.key[k] = to_gbcol(keys[k]);
  }
  for (int a = 0; a < g->L.naggs; ++a) {
    bool count_star = g->L.agg_kind[a] == DBHIP_AGG_COUNT && (!args || !args[a].data);
    if (count_star) continue;
    DBHIP_REQUIRE(args && args[a].data, "dbhip_groupby_add_block: missing aggregate argument column");
    if (g->L.agg_kind[a] != DBHIP_AGG_COUNT && args[a].type != g->L.agg_type[a]) {
      set_error("dbhip_groupby_add_block: aggregate %d argument has type %d, table expects %d", a, args[a].type,
                g->L.agg_type[a]);
      return DBHIP_ERR_INVALID;
    }
    C.arg[a] = to_gbcol(args[a]);
  }
  int32_t rc;
  int64_t done = 0;
  g->gbc_active = 0;   // (add_block_fast decides per call whether layout and columns qualify for the compact-row kernels)
  // The caller sized the table for about as many groups as this first block has rows (a join's output grouped by the join key,
  // TPC-H Q3: 3 M rows, 1.1 M groups): pre-aggregation has nothing to combine and costs more than the rows it saves
  // (r03: LDS pre-aggregation 0.59 ms + merge against 0.3 ms for the row path alone), the block goes straight to the row path.
  const bool expect_distinct = g->rows_seen == 0 && g->count_host == 0 && n >= (1 << 20) && g->hint_groups * 2 >= n && g->part_bits == 0;
  // (layouts past the generic LDS kernel's limits — e.g. eight aggregates — may still qualify for the compact-row kernels: add_block_fast decides)
  if ((fast_layout_ok(g->L) || !layout_has_wide_minmax(g->L)) && !g->has_long && !expect_distinct) {
    rc = add_block_fast(g, C, n, s, &done);
    if (rc >= 0) return rc;
  }
  // generic row path (any layout; high-cardinality continuation of the fast path), in bounded chunks
  const int64_t CHUNK = 32 << 20;
  const bool probe_here = !fast_layout_ok(g->L);  // wide layouts learn their cardinality on the row path
  while (done < n) {
    DBHIP_POLL_CANCEL(s, "dbhip_groupby_add_block");
    if (g->part_bits > 0 && n - done >= g->part_min_rows) {
      if ((rc = partitioned_step(g, C, n, s, &done))) return rc;
      continue;
    }
    int64_t cn = n - done < CHUNK ? n - done : CHUNK;
    if (probe_here && g->part_bits == 0 && cn > (1 << 20)) cn = 1 << 20;
    // (Serializing the rows in the order of their top hash bits — partition_scatter, so that probe and accumulate
    // walk the table slice by slice — was measured and does not pay: at 10^6..10^7 groups the row path is bound by
    // the two device-scope atomics per row, not by the random sectors. r01y: 17.9 ms vs 16.4 ms at 10 M groups.)
    if ((rc = ensure((void**)&g->rows_in, &g->rows_in_cap, (size_t)cn * g->L.W * 8))) return rc;
    if (C.filter) DBHIP_CHECK(hipMemsetAsync(&g->ctrl[7], 0, 8, s));
    if (layout_has_strings(g->L)) DBHIP_CHECK(hipMemsetAsync(&g->ctrl[9], 0, 8, s));
    hipLaunchKernelGGL(gb_serialize_kernel, dim3(grid_for(ceil_div(cn, 4), 256)), dim3(256), 0, s, g->L, C, done, cn, g->rows_in,
                       g->ctrl);
    DBHIP_LAUNCH_CHECK();
    if (layout_has_strings(g->L) && (rc = reserve_arena_for_chunk(g, s))) return rc;
    int64_t kept = cn;
    if (C.filter) {  // the passing rows were written densely: their number comes back with one small copy
      uint64_t k7 = 0;
      DBHIP_CHECK(hipMemcpyAsync(&k7, &g->ctrl[7], 8, hipMemcpyDeviceToHost, s));
      DBHIP_CHECK(hipStreamSynchronize(s));
      kept = (int64_t)k7;
    }
    if ((rc = merge_rows(g, g->rows_in, kept, s))) return rc;
    done += cn;
    g->rows_seen += cn;
    if (probe_here && g->part_bits == 0 && g->rows_seen >= (1 << 20)) {
      if (g->count_host > 32) decide_partitioning(g, g->count_host, g->rows_seen, n);
      else g->part_bits = -1;  // a handful of groups: the wave-combining accumulate kernel is the right tool
    }
  }
  return DBHIP_OK;
}

// Deserializing side of the reference's state exchange (TransformDeserializer -> AggregateFunction::batch_merge,
// aggregator/serde/transform_deserializer.rs): a block [state fields..., group columns...] as
// Payload::aggregate_flush produces it (payload_flush.rs:151-181) is merged into the table. The keys are serialized
// like an input block, the state words are filled from the field columns (gb_states_from_fields_kernel) and the rows
// go through the row merge path — the table's layout is never modified.
int32_t dbhip_groupby_merge_state_block(dbhip_groupby* g, const dbhip_col* keys, const dbhip_col* states,
                                        int64_t n, void* stream) {
  GB_DRAIN(g, resolve_stream(stream));
  DBHIP_REQUIRE(g && keys && (states || g->L.naggs == 0), "dbhip_groupby_merge_state_block: NULL argument");
  if (int32_t rs = refuse_str_minmax_state(g->L, "dbhip_groupby_merge_state_block")) return rs;
  const GbLayout& L = g->L;
  int32_t ftype[GB_MAX_AGGS * 3], fagg[GB_MAX_AGGS * 3];
  const int nf = state_fields(L, ftype, fagg);
  // validate everything BEFORE anything is queued
  StateFieldCols F;
  memset(&F, 0, sizeof(F));
  for (int f = 0; f < nf; ++f) {
    if (!states[f].data) {
      set_error("dbhip_groupby_merge_state_block: missing state field %d (aggregate %d)", f, fagg[f]);
      return DBHIP_ERR_INVALID;
    }
    if (states[f].type != ftype[f]) {
      set_error("dbhip_groupby_merge_state_block: state field %d (aggregate %d) has type %d, the serialized state has type %d", f,
                fagg[f], states[f].type, ftype[f]);
      return DBHIP_ERR_INVALID;
    }
    F.f[f] = to_gbcol(states[f]);
    F.f[f].validity = nullptr;  // state fields are never NULL (MinMax: the has-value field says it)
  }
  GbCols C;
  memset(&C, 0, sizeof(C));
  for (int k = 0; k < L.nkeys; ++k) {
    if (keys[k].type != L.key_type[k]) {
      set_error("dbhip_groupby_merge_state_block: key %d has type %d, table expects %d", k, keys[k].type, L.key_type[k]);
      return DBHIP_ERR_INVALID;
    }
    if (keys[k].validity && !L.key_nullable[k]) {
      set_error("dbhip_groupby_merge_state_block: key %d carries validity but was declared NOT NULL", k);
      return DBHIP_ERR_INVALID;
    }
    C.key[k] = to_gbcol(keys[k]);
  }
  if (n == 0) return DBHIP_OK;
  hipStream_t s = resolve_stream(stream);
  int32_t rc;
  const int64_t CHUNK = 32 << 20;
  for (int64_t done = 0; done < n; done += CHUNK) {
    const int64_t cn = n - done < CHUNK ? n - done : CHUNK;
    if ((rc = ensure((void**)&g->rows_in, &g->rows_in_cap, (size_t)cn * L.W * 8))) return rc;
    // keys + hash (argument pointers are NULL: the state words are overwritten by the next kernel)
    hipLaunchKernelGGL(gb_serialize_kernel, dim3(grid_for(ceil_div(cn, 4), 256)), dim3(256), 0, s, L, C, done, cn, g->rows_in, g->ctrl);
    StateFieldCols Fc = F;
    for (int f = 0; f < nf; ++f) {
      if (Fc.f[f].is_scalar || done == 0) continue;
      const int es = Fc.f[f].type == DBHIP_T_BOOL ? 0 : type_size(Fc.f[f].type);
      if (es) Fc.f[f].data = (const uint8_t*)Fc.f[f].data + (size_t)done * es;
      else Fc.f[f].data = (const uint8_t*)Fc.f[f].data + (done >> 3);   // CHUNK is a multiple of 8 bits
    }
    hipLaunchKernelGGL(gb_states_from_fields_kernel, dim3(grid_for(cn, 256)), dim3(256), 0, s, L, Fc, cn, g->rows_in, g->ctrl);
    DBHIP_LAUNCH_CHECK();
    if ((rc = merge_rows(g, g->rows_in, cn, s))) return rc;
  }
  return DBHIP_OK;
}

int32_t dbhip_groupby_state_fields(dbhip_groupby* g, int32_t* out_types_host, int32_t* out_agg_index_host, int32_t max_fields,
                                   int32_t* out_n_fields_host) {
  DBHIP_REQUIRE(g && out_n_fields_host, "dbhip_groupby_state_fields: NULL argument");
  if (int32_t rs = refuse_str_minmax_state(g->L, "dbhip_groupby_state_fields")) return rs;
  int32_t ftype[GB_MAX_AGGS * 3], fagg[GB_MAX_AGGS * 3];
  const int nf = state_fields(g->L, ftype, fagg);
  *out_n_fields_host = nf;
  for (int f = 0; f < nf && f < max_fields; ++f) {
    if (out_types_host) out_types_host[f] = ftype[f];
    if (out_agg_index_host) out_agg_index_host[f] = fagg[f];
  }
  return DBHIP_OK;
}

int32_t dbhip_groupby_merge_serialized(dbhip_groupby* g, const void* rows_dev, int64_t n_rows, void* stream) {
  GB_DRAIN(g, resolve_stream(stream));
  DBHIP_REQUIRE(g && (rows_dev || n_rows == 0), "dbhip_groupby_merge_serialized: NULL argument");
  return merge_rows(g, (const uint64_t*)rows_dev, n_rows, resolve_stream(stream));
}

int32_t dbhip_groupby_num_groups(dbhip_groupby* g, int64_t* out_host, void* stream) {
  GB_DRAIN(g, resolve_stream(stream));
  DBHIP_REQUIRE(g && out_host, "dbhip_groupby_num_groups: NULL argument");
  hipStream_t s = resolve_stream(stream);
  uint64_t c = 0;
  DBHIP_CHECK(hipMemcpyAsync(&c, g->ctrl, 8, hipMemcpyDeviceToHost, s));
  DBHIP_CHECK(hipStreamSynchronize(s));
  g->count_host = (int64_t)c;
  *out_host = (int64_t)c;
  return DBHIP_OK;
}

int32_t dbhip_groupby_row_bytes(dbhip_groupby* g, int64_t* out_host) {
  DBHIP_REQUIRE(g && out_host, "dbhip_groupby_row_bytes: NULL argument");
  *out_host = (int64_t)g->L.W * 8;
  return DBHIP_OK;
}

int32_t dbhip_groupby_flush_serialized(dbhip_groupby* g, void* out_rows_dev, int64_t max_rows,
                                       int64_t* out_n_rows_host, void* stream) {
  GB_DRAIN(g, resolve_stream(stream));
  DBHIP_REQUIRE(g && out_n_rows_host && (out_rows_dev || max_rows == 0), "dbhip_groupby_flush_serialized: NULL argument");
  hipStream_t s = resolve_stream(stream);
  DBHIP_CHECK(hipMemsetAsync(&g->ctrl[4], 0, 8, s));
  hipLaunchKernelGGL(gb_flush_kernel, dim3(grid_for(g->cap, 256)), dim3(256), 0, s, g->L, g->slot_hash,
                     g->rows, g->cap, (uint64_t*)out_rows_dev, max_rows, g->ctrl);
  DBHIP_LAUNCH_CHECK();
  uint64_t nflush = 0;
  DBHIP_CHECK(hipMemcpyAsync(&nflush, &g->ctrl[4], 8, hipMemcpyDeviceToHost, s));
  DBHIP_CHECK(hipStreamSynchronize(s));
  *out_n_rows_host = (int64_t)nflush;
  if ((int64_t)nflush > max_rows) {
    set_error("dbhip_groupby_flush_serialized: %lld groups do not fit max_rows=%lld", (long long)nflush, (long long)max_rows);
    return DBHIP_ERR_CAPACITY;
  }
  return DBHIP_OK;
}

// Exchange entry points move serialized rows WITHOUT the arena: a long (> 12 byte) string key in such a row is an offset into
// the SENDER's arena, which the receiving table would read as an address. Tables that hold long strings exchange through
// dbhip_groupby_flush_serialized + dbhip_groupby_arena -> dbhip_groupby_merge_serialized_arena (which rebases the offsets).
static int32_t refuse_long_strings(const dbhip_groupby* g, const char* fn) {
  if (layout_has_str_minmax(g->L)) {
    set_error("%s: a min / max over String state refers to bytes in this table's arena; such tables are merged in process "
              "(dbhip_groupby_merge_serialized from a live table) and do not travel", fn);
    return DBHIP_ERR_UNSUPPORTED;
  }
  if (!g->has_long) return DBHIP_OK;
  set_error("%s: the table holds string keys longer than 12 bytes; exchange it with dbhip_groupby_flush_serialized + dbhip_groupby_arena "
            "-> dbhip_groupby_merge_serialized_arena", fn);
  return DBHIP_ERR_UNSUPPORTED;
}

int32_t dbhip_groupby_flush_block(dbhip_groupby* g, void* out_block_dev, int64_t max_rows, void* stream) {
  GB_DRAIN(g, resolve_stream(stream));
  DBHIP_REQUIRE(g && out_block_dev && max_rows >= 1, "dbhip_groupby_flush_block: bad argument");
  if (int32_t rl = refuse_long_strings(g, "dbhip_groupby_flush_block")) return rl;
  hipStream_t s = resolve_stream(stream);
  uint64_t* block = (uint64_t*)out_block_dev;
  DBHIP_CHECK(hipMemsetAsync(&g->ctrl[4], 0, 8, s));
  hipLaunchKernelGGL(gb_flush_kernel, dim3(grid_for(g->cap, 256)), dim3(256), 0, s, g->L, g->slot_hash, g->rows, g->cap,
                     block + g->L.W, max_rows, g->ctrl);
  hipLaunchKernelGGL(gb_block_header_kernel, dim3(1), dim3(64), 0, s, block, g->L.W, max_rows, g->ctrl);
  DBHIP_LAUNCH_CHECK();
  return DBHIP_OK;  // nothing is read back: the header travels with the block
}

int32_t dbhip_groupby_merge_blocks(dbhip_groupby* g, const void* blocks_dev, int32_t n_blocks, int64_t max_rows,
                                   int32_t skip_block, void* stream) {
  GB_DRAIN(g, resolve_stream(stream));
  DBHIP_REQUIRE(g && blocks_dev && n_blocks >= 1 && n_blocks <= 4096 && max_rows >= 1, "dbhip_groupby_merge_blocks: bad argument");
  if (int32_t rl = refuse_long_strings(g, "dbhip_groupby_merge_blocks")) return rl;
  hipStream_t s = resolve_stream(stream);
  const int W = g->L.W;
  const int64_t stride = (max_rows + 1) * W;
  const uint64_t* blocks = (const uint64_t*)blocks_dev;
  std::vector<uint64_t> head((size_t)n_blocks);
  DBHIP_CHECK(hipMemcpy2DAsync(head.data(), 8, blocks, (size_t)stride * 8, 8, (size_t)n_blocks, hipMemcpyDeviceToHost, s));
  DBHIP_CHECK(hipStreamSynchronize(s));
  int64_t total = 0;
  // EVERY header is checked, the caller's own block included: the owner of an overflowed block must take the
  // variable-length path together with the ranks that see the overflow in the gathered headers (otherwise the owner
  // would merge and return while the others enter a collective). Decided BEFORE the table is touched.
  for (int b = 0; b < n_blocks; ++b) {
    if (head[b] == ~0ULL || (int64_t)head[b] > max_rows) {
      set_error("dbhip_groupby_merge_blocks: block %d overflowed max_rows=%lld (exchange the rows with "
                "dbhip_groupby_flush_serialized / merge_serialized instead)", b, (long long)max_rows);
      return DBHIP_ERR_CAPACITY;
    }
    if (b != skip_block) total += (int64_t)head[b];
  }
  if (total == 0) return DBHIP_OK;
  int32_t rc;
  if ((rc = ensure((void**)&g->rows_in, &g->rows_in_cap, (size_t)total * W * 8))) return rc;
  hipLaunchKernelGGL(gb_compact_blocks_kernel, dim3(n_blocks), dim3(256), 0, s, blocks, stride, W, skip_block, g->rows_in);
  DBHIP_LAUNCH_CHECK();
  return merge_rows(g, g->rows_in, total, s);
}

int32_t dbhip_groupby_result_type(const dbhip_agg_desc* agg, int32_t* out_type, uint8_t* out_precision,
                                  uint8_t* out_scale) {
  DBHIP_REQUIRE(agg && out_type, "dbhip_groupby_result_type: NULL argument");
  uint8_t p = 0, sc = 0;
  int t = -1;
  switch (agg->kind) {
    case DBHIP_AGG_COUNT: t = DBHIP_T_U64; break;
    case DBHIP_AGG_SUM:
      switch (agg->arg_type) {  // ResultTypeOfUnary::Sum, sum_coercion
        case DBHIP_T_I8: case DBHIP_T_I16: case DBHIP_T_I32: case DBHIP_T_I64: t = DBHIP_T_I64; break;
        case DBHIP_T_U8: case DBHIP_T_U16: case DBHIP_T_U32: case DBHIP_T_U64: t = DBHIP_T_U64; break;
        case DBHIP_T_F32: case DBHIP_T_F64: t = DBHIP_T_F64; break;
        case DBHIP_T_DEC64: t = DBHIP_T_DEC64; p = 18; sc = agg->arg_scale; break;   // aggregate_sum.rs:404-406
        case DBHIP_T_DEC128: t = DBHIP_T_DEC128; p = 38; sc = agg->arg_scale; break;
        case DBHIP_T_DEC256: t = DBHIP_T_DEC256; p = 76; sc = agg->arg_scale; break;
      }
      break;
    case DBHIP_AGG_MIN: case DBHIP_AGG_MAX:
      t = agg->arg_type; p = agg->arg_precision; sc = agg->arg_scale;
      break;
  }
  if (t < 0) {
    set_error("dbhip_groupby_result_type: unsupported aggregate (%d on type %d)", agg->kind, agg->arg_type);
    return DBHIP_ERR_INVALID;
  }
  *out_type = t;
  if (out_precision) *out_precision = p;
  if (out_scale) *out_scale = sc;
  return DBHIP_OK;
}

static int32_t flush_columns(dbhip_groupby* g, void* const* out_keys_host, uint8_t* const* out_key_validity_host,
                             void* const* out_aggs_host, uint8_t* const* out_agg_validity_host, void* const* out_fields_host,
                             uint64_t* out_hashes, int64_t max_rows, int64_t* out_n_rows_host, void* stream) {
  hipStream_t s = resolve_stream(stream);
  uint64_t* tmp = (uint64_t*)scratch((size_t)(max_rows > 0 ? max_rows : 1) * g->L.W * 8, 3, s);
  if (!tmp) return DBHIP_ERR_HIP;
  int32_t rc = dbhip_groupby_flush_serialized(g, tmp, max_rows, out_n_rows_host, stream);
  if (rc) return rc;
  int64_t n = *out_n_rows_host;
  if (n == 0) return DBHIP_OK;
  ResultPtrs P;
  memset(&P, 0, sizeof(P));
  const size_t bm_bytes = (size_t)ceil_div(max_rows, 64) * 8;
  for (int k = 0; k < g->L.nkeys; ++k) {
    P.keys[k] = out_keys_host ? out_keys_host[k] : nullptr;
    P.key_validity[k] = out_key_validity_host ? (uint32_t*)out_key_validity_host[k] : nullptr;
    if (P.key_validity[k]) DBHIP_CHECK(hipMemsetAsync(P.key_validity[k], 0, bm_bytes, s));
  }
  for (int a = 0; a < g->L.naggs; ++a) {
    P.aggs[a] = out_aggs_host ? out_aggs_host[a] : nullptr;
    P.agg_validity[a] = out_agg_validity_host ? (uint32_t*)out_agg_validity_host[a] : nullptr;
    if (P.agg_validity[a]) DBHIP_CHECK(hipMemsetAsync(P.agg_validity[a], 0, bm_bytes, s));
  }
  P.hashes = out_hashes;
  P.arena = g->arena;
  DBHIP_CHECK(hipMemsetAsync(&g->ctrl[3], 0, 8, s));
  hipLaunchKernelGGL(gb_result_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, g->L, tmp, n, P, g->ctrl);
  if (out_fields_host) {
    int32_t ftype[GB_MAX_AGGS * 3];
    const int nf = state_fields(g->L, ftype, nullptr);
    StateFieldPtrs SP;
    memset(&SP, 0, sizeof(SP));
    for (int f = 0; f < nf; ++f) {
      SP.f[f] = out_fields_host[f];
      if (SP.f[f] && ftype[f] == DBHIP_T_BOOL) DBHIP_CHECK(hipMemsetAsync(SP.f[f], 0, bm_bytes, s));
    }
    hipLaunchKernelGGL(gb_state_fields_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, g->L, tmp, n, SP, g->ctrl, (const uint8_t*)g->arena);
  }
  DBHIP_LAUNCH_CHECK();
  uint64_t err = 0;
  DBHIP_CHECK(hipMemcpyAsync(&err, &g->ctrl[3], 8, hipMemcpyDeviceToHost, s));
  DBHIP_CHECK(hipStreamSynchronize(s));
  if (err & 1) {
    set_error("Decimal overflow: sum state not in [DECIMAL_MIN, DECIMAL_MAX]");
    return DBHIP_ERR_OVERFLOW;
  }
  if (err & 8) {
    set_error("groupby: more than 4 GiB of long string keys: a BinaryView offset is 32 bits; flush the table in pieces");
    return DBHIP_ERR_CAPACITY;
  }
  return DBHIP_OK;
}

int32_t dbhip_groupby_arena(dbhip_groupby* g, const void** out_ptr_host, int64_t* out_bytes_host, void* stream) {
  GB_DRAIN(g, resolve_stream(stream));
  DBHIP_REQUIRE(g && out_ptr_host && out_bytes_host, "dbhip_groupby_arena: NULL argument");
  hipStream_t s = resolve_stream(stream);
  uint64_t used = 0;
  DBHIP_CHECK(hipMemcpyAsync(&used, &g->ctrl[8], 8, hipMemcpyDeviceToHost, s));
  DBHIP_CHECK(hipStreamSynchronize(s));
  *out_ptr_host = g->arena;
  *out_bytes_host = (int64_t)used;
  return DBHIP_OK;
}

namespace {
// serialized rows of ANOTHER table (long strings by offset into that table's arena, which the caller shipped along) ->
// input rows (long strings by address); sums the long bytes for the arena reservation
__global__ __launch_bounds__(256) void gb_rebase_rows_kernel(GbLayout L, const uint64_t* rows, int64_t n, const uint8_t* arena,
                                                             uint64_t* out, uint64_t* ctrl) {
  const int64_t n_pad = (n + 63) & ~63LL;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_pad; i += (int64_t)gridDim.x * blockDim.x) {
    uint64_t lb = 0;
    if (i < n) {
      const uint64_t* r = rows + i * L.W;
      uint64_t* o = out + i * L.W;
      for (int k = 0; k < L.W; ++k) {
        uint64_t v = r[k];
        if (k < L.nkey_words && ((L.str_w1_mask >> k) & 1) && (uint32_t)r[k - 1] > 12) {
          v = (uint64_t)(arena + v);
          lb += ((uint32_t)r[k - 1] + 7) & ~7u;
        }
        o[k] = v;
      }
    }
    lb = wave_sum_u64(lb);
    if (lb && lane_id() == 0) atomicAdd((unsigned long long*)&ctrl[9], (unsigned long long)lb);
  }
}
}  // namespace

int32_t dbhip_groupby_merge_serialized_arena(dbhip_groupby* g, const void* rows_dev, int64_t n_rows, const void* arena_dev, void* stream) {
  GB_DRAIN(g, resolve_stream(stream));
  DBHIP_REQUIRE(g && (rows_dev || n_rows == 0), "dbhip_groupby_merge_serialized_arena: NULL argument");
  if (n_rows == 0) return DBHIP_OK;
  hipStream_t s = resolve_stream(stream);
  if (!layout_has_strings(g->L)) return merge_rows(g, (const uint64_t*)rows_dev, n_rows, s);
  int32_t rc;
  // (its own scratch: merge_rows may be handed g->rows_in by other callers, not by this one)
  if ((rc = ensure((void**)&g->spill_rows, &g->spill_rows_cap, (size_t)n_rows * g->L.W * 8))) return rc;
  DBHIP_CHECK(hipMemsetAsync(&g->ctrl[9], 0, 8, s));
  hipLaunchKernelGGL(gb_rebase_rows_kernel, dim3(grid_for(n_rows, 256)), dim3(256), 0, s, g->L, (const uint64_t*)rows_dev, n_rows,
                     (const uint8_t*)arena_dev, g->spill_rows, g->ctrl);
  DBHIP_LAUNCH_CHECK();
  if ((rc = reserve_arena_for_chunk(g, s))) return rc;
  return merge_rows(g, g->spill_rows, n_rows, s);
}

int32_t dbhip_groupby_flush_result(dbhip_groupby* g, void* const* out_keys_host,
                                   uint8_t* const* out_key_validity_host, void* const* out_aggs_host,
                                   uint64_t* out_hashes, int64_t max_rows, int64_t* out_n_rows_host,
                                   void* stream) {
  GB_DRAIN(g, resolve_stream(stream));
  DBHIP_REQUIRE(g && out_n_rows_host, "dbhip_groupby_flush_result: NULL argument");
  return flush_columns(g, out_keys_host, out_key_validity_host, out_aggs_host, nullptr, nullptr, out_hashes, max_rows,
                       out_n_rows_host, stream);
}

int32_t dbhip_groupby_flush_result_nullable(dbhip_groupby* g, void* const* out_keys_host,
                                            uint8_t* const* out_key_validity_host, void* const* out_aggs_host,
                                            uint8_t* const* out_agg_validity_host, uint64_t* out_hashes, int64_t max_rows,
                                            int64_t* out_n_rows_host, void* stream) {
  GB_DRAIN(g, resolve_stream(stream));
  DBHIP_REQUIRE(g && out_n_rows_host, "dbhip_groupby_flush_result_nullable: NULL argument");
  return flush_columns(g, out_keys_host, out_key_validity_host, out_aggs_host, out_agg_validity_host, nullptr, out_hashes,
                       max_rows, out_n_rows_host, stream);
}

int32_t dbhip_groupby_flush_state_block(dbhip_groupby* g, void* const* out_keys_host, uint8_t* const* out_key_validity_host,
                                        void* const* out_state_fields_host, uint64_t* out_hashes, int64_t max_rows,
                                        int64_t* out_n_rows_host, void* stream) {
  GB_DRAIN(g, resolve_stream(stream));
  DBHIP_REQUIRE(g && out_n_rows_host && (out_state_fields_host || g->L.naggs == 0), "dbhip_groupby_flush_state_block: NULL argument");
  if (int32_t rs = refuse_str_minmax_state(g->L, "dbhip_groupby_flush_state_block")) return rs;
  return flush_columns(g, out_keys_host, out_key_validity_host, nullptr, nullptr, out_state_fields_host, out_hashes, max_rows,
                       out_n_rows_host, stream);
}

// ---- a12: hash partitioning of the group rows (payload.rs:548-589) ------------------------------------------
static int32_t ensure_xcur(dbhip_groupby* g) {
  if (g->xcur) return DBHIP_OK;
  DBHIP_CHECK(hipMalloc((void**)&g->xcur, (size_t)(2 * 4096 + 2) * 8));
  return DBHIP_OK;
}

int32_t dbhip_groupby_partition_blocks(dbhip_groupby* g, int32_t n_buckets, void* out_blocks_dev, int64_t max_rows, void* stream) {
  GB_DRAIN(g, resolve_stream(stream));
  DBHIP_REQUIRE(g && out_blocks_dev && n_buckets >= 1 && n_buckets <= 4096 && max_rows >= 1, "dbhip_groupby_partition_blocks: bad argument");
  if (int32_t rl = refuse_long_strings(g, "dbhip_groupby_partition_blocks")) return rl;
  hipStream_t s = resolve_stream(stream);
  int32_t rc = ensure_xcur(g);
  if (rc) return rc;
  const int W = g->L.W;
  const int64_t stride = (max_rows + 1) * W;
  DBHIP_CHECK(hipMemsetAsync(g->xcur, 0, (size_t)n_buckets * 8, s));
  hipLaunchKernelGGL(gb_partition_rows_kernel, dim3(grid_for(g->cap, 256)), dim3(256), 0, s, g->L, g->slot_hash, g->rows, g->cap,
                     (uint32_t)n_buckets, max_rows, stride, (const uint64_t*)nullptr, (uint64_t*)out_blocks_dev,
                     (unsigned long long*)g->xcur);
  hipLaunchKernelGGL(gb_partition_headers_kernel, dim3(1), dim3(256), 0, s, (uint64_t*)out_blocks_dev, W, stride, max_rows,
                     (uint32_t)n_buckets, (const unsigned long long*)g->xcur);
  DBHIP_LAUNCH_CHECK();
  return DBHIP_OK;  // nothing is read back: the headers travel with the blocks
}

int32_t dbhip_groupby_flush_partitioned(dbhip_groupby* g, int32_t n_buckets, void* out_rows_dev, int64_t max_rows,
                                        int64_t* out_counts_host, void* stream) {
  GB_DRAIN(g, resolve_stream(stream));
  DBHIP_REQUIRE(g && out_counts_host && n_buckets >= 1 && n_buckets <= 4096 && (out_rows_dev || max_rows == 0),
                "dbhip_groupby_flush_partitioned: bad argument");
  if (int32_t rl = refuse_long_strings(g, "dbhip_groupby_flush_partitioned")) return rl;
  hipStream_t s = resolve_stream(stream);
  int32_t rc = ensure_xcur(g);
  if (rc) return rc;
  uint64_t* cur = g->xcur;
  uint64_t* base = g->xcur + 4096;
  DBHIP_CHECK(hipMemsetAsync(cur, 0, (size_t)n_buckets * 8, s));
  const int grid = grid_for(g->cap, 256);
  hipLaunchKernelGGL(gb_partition_rows_kernel, dim3(grid), dim3(256), 0, s, g->L, g->slot_hash, g->rows, g->cap, (uint32_t)n_buckets,
                     (int64_t)0, (int64_t)0, (const uint64_t*)nullptr, (uint64_t*)nullptr, (unsigned long long*)cur);
  DBHIP_LAUNCH_CHECK();
  std::vector<uint64_t> cnt((size_t)n_buckets), off((size_t)n_buckets + 1);
  DBHIP_CHECK(hipMemcpyAsync(cnt.data(), cur, (size_t)n_buckets * 8, hipMemcpyDeviceToHost, s));
  DBHIP_CHECK(hipStreamSynchronize(s));
  uint64_t total = 0;
  for (int b = 0; b < n_buckets; ++b) { off[b] = total; total += cnt[b]; out_counts_host[b] = (int64_t)cnt[b]; }
  off[n_buckets] = total;
  if ((int64_t)total > max_rows) {
    set_error("dbhip_groupby_flush_partitioned: %llu groups do not fit max_rows=%lld", (unsigned long long)total, (long long)max_rows);
    return DBHIP_ERR_CAPACITY;
  }
  if (total == 0) return DBHIP_OK;
  DBHIP_CHECK(hipMemcpyAsync(base, off.data(), (size_t)(n_buckets + 1) * 8, hipMemcpyHostToDevice, s));
  DBHIP_CHECK(hipMemsetAsync(cur, 0, (size_t)n_buckets * 8, s));
  hipLaunchKernelGGL(gb_partition_rows_kernel, dim3(grid), dim3(256), 0, s, g->L, g->slot_hash, g->rows, g->cap, (uint32_t)n_buckets,
                     (int64_t)0, (int64_t)0, (const uint64_t*)base, (uint64_t*)out_rows_dev, (unsigned long long*)cur);
  DBHIP_LAUNCH_CHECK();
  DBHIP_CHECK(hipStreamSynchronize(s));  // `off` (pageable host memory) was the source of an async copy
  return DBHIP_OK;
}

int32_t dbhip_groupby_replace_with_blocks(dbhip_groupby* g, const void* blocks_dev, int32_t n_blocks, int64_t max_rows, void* stream) {
  GB_DRAIN(g, resolve_stream(stream));
  DBHIP_REQUIRE(g && blocks_dev && n_blocks >= 1 && n_blocks <= 4096 && max_rows >= 1, "dbhip_groupby_replace_with_blocks: bad argument");
  if (int32_t rl = refuse_long_strings(g, "dbhip_groupby_replace_with_blocks")) return rl;
  hipStream_t s = resolve_stream(stream);
  const int W = g->L.W;
  const int64_t stride = (max_rows + 1) * W;
  const uint64_t* blocks = (const uint64_t*)blocks_dev;
  std::vector<uint64_t> head((size_t)n_blocks * 2);
  DBHIP_CHECK(hipMemcpy2DAsync(head.data(), 16, blocks, (size_t)stride * 8, 16, (size_t)n_blocks, hipMemcpyDeviceToHost, s));
  DBHIP_CHECK(hipStreamSynchronize(s));
  int64_t total = 0;
  for (int b = 0; b < n_blocks; ++b) {
    if (head[2 * b] == ~0ULL || (int64_t)head[2 * b] > max_rows || head[2 * b + 1] != 0) {  // decided BEFORE the table is touched
      set_error("dbhip_groupby_replace_with_blocks: the sender of block %d overflowed max_rows=%lld (exchange the rows with "
                "dbhip_groupby_flush_partitioned / merge_serialized instead)", b, (long long)max_rows);
      return DBHIP_ERR_CAPACITY;
    }
    total += (int64_t)head[2 * b];
  }
  int32_t rc;
  if ((rc = dbhip_groupby_reset(g, stream))) return rc;
  if (total == 0) return DBHIP_OK;
  if ((rc = ensure((void**)&g->rows_in, &g->rows_in_cap, (size_t)total * W * 8))) return rc;
  hipLaunchKernelGGL(gb_compact_blocks_kernel, dim3(n_blocks), dim3(256), 0, s, blocks, stride, W, -1, g->rows_in);
  DBHIP_LAUNCH_CHECK();
  return merge_rows(g, g->rows_in, total, s);
}

int32_t dbhip_groupby_reset(dbhip_groupby* g, void* stream) {
  DBHIP_REQUIRE(g, "dbhip_groupby_reset: NULL argument");
  hipStream_t s = resolve_stream(stream);
  if (g->fa_pipe) { const int32_t rc = dbhip_fagg_pipe_reset_internal(g->fa_pipe, s); if (rc) return rc; }   // queued blocks are dropped with the groups
  DBHIP_CHECK(hipMemsetAsync(g->slot_hash, 0, (size_t)g->cap * 8, s));
  DBHIP_CHECK(hipMemsetAsync(g->ctrl, 0, 128, s));
  g->count_host = 0;
  g->has_long = 0;
  g->fast_disabled = 0;
  g->fast_trusted = 0;
  g->lds_big = 0;
  g->fagg_disabled = 0;
  g->gbc_lcap = 0;
  g->gbc_active = 0;
  g->gbc_part_lcap_max = 0;
  g->part_validate = 0; g->part_validated = 0;
  if (g->part_min_rows > 1) g->part_bits = 0;  // (a forced partitioning — test hook — survives reset)
  g->rows_seen = 0;
  return DBHIP_OK;
}

int32_t dbhip_groupby_destroy(dbhip_groupby* g) {
  if (!g) return DBHIP_OK;
  (void)hipDeviceSynchronize();
  if (g->fa_pipe) dbhip_fagg_pipe_destroy_internal(g->fa_pipe);
  if (g->slot_hash) (void)dbhip_free(g->slot_hash);
  if (g->rows) (void)dbhip_free(g->rows);
  if (g->ctrl) (void)hipFree(g->ctrl);
  if (g->rows_in) (void)dbhip_free(g->rows_in);
  if (g->gid) (void)dbhip_free(g->gid);
  if (g->retry) (void)dbhip_free(g->retry);
  if (g->partial) (void)dbhip_free(g->partial);
  if (g->part_meta) (void)dbhip_free(g->part_meta);
  if (g->spill_idx) (void)dbhip_free(g->spill_idx);
  if (g->spill_rows) (void)dbhip_free(g->spill_rows);
  if (g->gbc_spill) (void)dbhip_free(g->gbc_spill);
  if (g->gbc_split) (void)dbhip_free(g->gbc_split);
  if (g->xcur) (void)hipFree(g->xcur);
  if (g->arena) (void)dbhip_free(g->arena);
  delete g;
  return DBHIP_OK;
}

}  // extern "C"

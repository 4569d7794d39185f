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
  uint64_t* xcur;                          // exchange partitioning: cursor[4096] | base[4097] | status of the queued exchange
  int xcur_dirty;                          // the cursors are not known to be zero
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
struct GbCols;
int32_t dbhip_fagg_pipe_add_columns_internal(dbhip_groupby* g, void* pipe, const GbCols& C, int64_t n, hipStream_t s);
#define GB_DRAIN(g, s)                                                                   \
  do {                                                                                   \
    if ((g) && (g)->fa_pipe) {                                                           \
      const int32_t _rc = dbhip_fagg_pipe_drain_internal((g), (g)->fa_pipe, (s));        \
      if (_rc) return _rc;                                                               \
    }                                                                                    \
  } while (0)

#include "gbk_rows.h"
int32_t dbhip_groupby_build_layout_internal(const int32_t* key_types, const uint8_t* key_nullable, int nkeys, const dbhip_agg_desc* aggs,
                                            int naggs, GbLayout* L) {
  return build_layout(key_types, key_nullable, nkeys, aggs, naggs, L);
}
#include "gbk_merge_lds.h"
int32_t dbhip_fagg_add_columns_internal(dbhip_groupby* g, const GbCols& C, int64_t row0, int64_t n, bool may_compile, hipStream_t s);  // k_fagg.hip
bool dbhip_fagg_last_refusal_is_pending_internal();
#include "gbk_partitioned.h"

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
  return merge_rows_unpinned(g, rows, n_max, s, n_dev, abort_dev, 1);
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

#include "gbk_api.h"

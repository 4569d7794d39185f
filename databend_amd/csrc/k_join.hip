// k_join.hip — packed fixed-width join/group keys (a14) and the hash-join table (a15).
//
// Reference: HashJoinHashTable<K> (hash_join_table/hashjoin_hashtable.rs:26-137): bucket array of
// u64 = 48-bit entry pointer | 16-bit one-hot tag (1 << (48 + (hash & 15))), capacity
// max(2*rows -> pow2, 1024) (:95-108), idx = hash >> (64 - log2 cap), lock-free CAS prepend
// (:110-137); probe: tag early-reject, then walk the chain comparing keys
// (new_hash_join/hashtable/fixed_keys.rs:139-142,209-269) and emit (probe_idx, RowPtr).
// Device geometry (same ideas, 32-bit row ids instead of pointers):
//   head[cap]   u64 = (build row + 1) | tag bits << 32; insertion = atomicExch on the low half
//               (prepend) + atomicOr on the high half; chains are only walked by later launches
//   ent[rows]   array of structures {key words, next, valid}: the key compare and the chain link
//               come from ONE 64-byte sector (16 B entries for keys <= 8 bytes, 32 B for 16-byte keys)
// NULL keys never match (validity-0 rows are neither inserted nor probed). The join hash itself is
// not part of the parity contract (FastHash is CRC32-C or a murmur mix depending on the host CPU,
// common/hashtable/src/traits.rs:199-211): a 64-bit multiply-xorshift is used.
// Pair order across threads is unspecified in the reference; here pairs come out sorted by
// (probe_idx, build_row): count per probe row -> exclusive scan -> ordered emit. The count pass
// remembers the matching build row, so for unique build keys (primary-key joins) the emit pass is a
// pure stream compaction and the table is walked once.
#include "dev_common.h"
#include "dev_scan.h"
#include "runtime.h"

#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <new>

using namespace dbhip;

#define DBHIP_TRY(x) do { int32_t _rc = (x); if (_rc) return _rc; } while (0)

struct dbhip_join {
  int kw;              // key words (u64): 1, 2 or 4 (KeysU256)
  int es;              // entry stride in u64 words: 2, 4 or 8
  uint64_t* ent;       // [cap_rows * es]: key words, then (next | valid << 32)
  int64_t nrows, cap_rows;
  uint64_t* head;      // [buckets]
  uint64_t* occ;       // [buckets / 64] bit b = bucket b is not empty (tables whose heads outgrow the L2 only), else NULL
  int64_t buckets;
  int shift;
  bool finalized;
  // probe scratch
  uint8_t* cnt; uint32_t* firstm; uint32_t* tsum; uint64_t* off; uint64_t* blk; size_t scratch_rows;   // per row: cnt, firstm; per tile: tsum, off
  uint64_t* total_dev;
  // the last counted probe block: dbhip_join_probe of the SAME block (count first, then emit into buffers of the
  // right size — the only way a caller can size its outputs) reuses the counts instead of walking the table again
  const void* prep_keys; const uint8_t* prep_valid; int64_t prep_n; uint64_t prep_total; hipStream_t prep_stream; bool prepared;
  // right / full outer, right semi / anti joins: bit r = build row r was matched by some probe row of ANY probe block so far
  // (the reference's per-row scan map, new_hash_join/memory/right_join*.rs; allocated on first use, after finalize)
  uint32_t* bmark; int64_t bmark_rows;
};

namespace {

template <int KW>
__device__ __forceinline__ uint64_t join_hash(const uint64_t* k) {
  if (KW == 1) return agg_hash_u64(k[0]);
  uint64_t h = agg_hash_u64(k[0] * 0x9E3779B97F4A7C15ULL ^ agg_hash_u64(k[1]));
#pragma unroll
  for (int w = 2; w < KW; ++w) h = agg_hash_u64(h * 0x9E3779B97F4A7C15ULL ^ agg_hash_u64(k[w]));   // KeysU256: four words
  return h;
}

template <int KW>
__global__ __launch_bounds__(256) void join_copy_kernel(const uint64_t* keys, const uint8_t* validity, int64_t n,
                                                        uint64_t* ent) {
  constexpr int ES = KW * 2;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    uint64_t* e = ent + i * ES;
#pragma unroll
    for (int w = 0; w < KW; ++w) e[w] = keys[i * KW + w];
    const uint64_t v = validity ? (uint64_t)bit_get(validity, i) : 1;
    e[KW] = v << 32;
#pragma unroll
    for (int w = KW + 1; w < ES; ++w) e[w] = 0;
  }
}

// Bucket and tag of a key. Keys wider than one word: an ordinary hash. ONE-word keys (every integer / date / decimal join key):
// the 64 keys that share key >> 6 take 64 CONSECUTIVE buckets starting at a hashed position. Sorted or clustered probe keys (a
// fact table stored in the order of its foreign key, TPC-H lineitem by l_orderkey) then read a handful of 64-byte head sectors
// per wave instead of one per row, and because build rows keep their arrival order the entries they reach are neighbours too;
// unrelated keys still scatter like a hash (the start of every run is hashed, runs overlap freely, so bucket loads are the sums
// of independent runs — the same occupancy statistics as hashing every key on its own). r03, SF100 lineitem probe, 600 M rows:
// 2.9 -> see DESIGN.md.
template <int KW>
__device__ __forceinline__ uint64_t join_bucket(const uint64_t* k, int cfg, uint32_t* tag) {
  const int shift = cfg & 255;   // cfg = shift | D << 8
  if (KW == 1) {
    const int D = cfg >> 8;      // 2^D neighbouring key values share a bucket (sparse clustered keys, see dbhip_join_finalize)
    // (round 6) ONE multiply: Fibonacci hashing of the run id — the bucket index below takes the TOP bits of the product, the ones a
    // multiplicative hash mixes best; the table's geometry is private to this file (set-equal results). (Also tried in round 6
    // and reverted: an occupancy filter of 16 positions per build row instead of one bit per bucket — 12 % instead of 39 % of the probes
    // of a selective join go on to fetch a head sector, but the filter (29 MB for TPC-H Q3's orders) no longer stays in one XCD's L2:
    // Q3 SF100 9.46 -> 10.22 ms.)
    const uint64_t hg = (k[0] >> (6 + D)) * 0x9E3779B97F4A7C15ULL;
    *tag = (uint32_t)((hg >> 40) ^ k[0]) & 15u;
    return ((hg >> shift) + ((k[0] >> D) & 63u)) & ((~0ULL) >> shift);
  }
  const uint64_t h = join_hash<KW>(k);
  *tag = (uint32_t)h & 15u;
  return h >> shift;
}

template <int KW>
__global__ __launch_bounds__(256) void join_build_kernel(uint64_t* ent, int64_t n, uint64_t* head, int shift) {
  constexpr int ES = KW * 2;
  uint32_t* head32 = (uint32_t*)head;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    uint64_t* e = ent + i * ES;
    if (!(e[KW] >> 32)) continue;
    uint32_t tag;
    const uint64_t idx = join_bucket<KW>(e, shift, &tag);
    const uint32_t old = atomicExch(&head32[2 * idx], (uint32_t)(i + 1));  // prepend
    atomicOr(&head32[2 * idx + 1], 1u << tag);
    ((uint32_t*)&e[KW])[0] = old;
  }
}

// walks the chain that starts at head word hd for probe key k; returns the number of matching build rows, *last = one of them
template <int KW>
__device__ __forceinline__ uint32_t join_walk_from(const uint64_t* ent, uint64_t hd, uint32_t tag, const uint64_t* k, uint32_t* last) {
  constexpr int ES = KW * 2;
  uint32_t c = 0;
  if (!((hd >> (32 + tag)) & 1)) return 0;  // empty bucket or tag miss (fixed_keys.rs:139-142)
  for (uint32_t e = (uint32_t)hd; e;) {
    const uint64_t* p = ent + (uint64_t)(e - 1) * ES;
    bool eq = true;
    uint64_t link;
    if (KW == 1) {
      const ulonglong2 v = *(const ulonglong2*)p;
      eq = v.x == k[0];
      link = v.y;
    } else {
      const ulonglong2 v0 = *(const ulonglong2*)p;
      eq = v0.x == k[0] && v0.y == k[1];
      if (KW == 4) {   // KeysU256: a 64-byte entry (4 key words, link, padding) = one 64-byte sector
        const ulonglong2 v1 = *(const ulonglong2*)(p + 2);
        eq = eq && v1.x == k[2] && v1.y == k[3];
      }
      link = p[KW];
    }
    if (eq) { ++c; *last = e - 1; }
    e = (uint32_t)link;
  }
  return c;
}
template <int KW>
__device__ __forceinline__ uint32_t join_walk(const uint64_t* ent, const uint64_t* head, int shift, const uint64_t* k,
                                              uint32_t* last) {
  uint32_t tag;
  const uint64_t idx = join_bucket<KW>(k, shift, &tag);
  return join_walk_from<KW>(ent, head[idx], tag, k, last);
}

// Probe geometry: a WAVE takes TILES of 256 consecutive probe rows and a lane takes FOUR CONSECUTIVE rows of the tile
// (row = tile * 256 + lane * 4 + r): its keys are 32 contiguous bytes (two 16-byte loads), its validity bits one nibble, its
// four saturated u8 counts one 32-bit store, and all four bucket heads are requested before the first chain is walked. The
// count pass leaves, per row, the u8 count (always) and the matching build row (only where there is one: the stores of a sparse
// join touch few lines), and per TILE the exact number of pairs; the exclusive scan runs over the tiles (n / 256 values, not n)
// and the emit pass rebuilds the row offsets inside a tile with one wave scan over the per-lane sums. Nothing in either pass
// crosses a wave: no LDS, no barrier (r03: with 1024-row workgroup tiles the emit pass of a 600 M-row probe was 586 k barriers
// long — 0.97 ms for 3 M pairs). (r02: per-row u32 counts + u32 build rows + u64 offsets and a device-wide scan over all n rows
// cost 28 bytes of traffic per probe row and ~3 ms of scan kernels per 600 M probe rows — more than the table walk itself at
// 1 % matches.)
constexpr int JOIN_TILE = 256;

__device__ __forceinline__ void join_tile_total(uint32_t tsum, uint32_t* tile_sum, int64_t t) {
  tsum = (uint32_t)wave_sum_u64(tsum);
  if (lane_id() == 0) tile_sum[t] = tsum;
}

// FULL tiles [0, nfull) of a key column that starts on a 16-byte boundary. The loop is bound by dependent round trips (keys ->
// occupancy bits -> bucket heads -> chain entries), not by bytes, so everything in it is branch-free straight-line code: the four
// heads are requested together, the keys and validity byte of the NEXT tile right after them (loads retire in order: the wait
// for the heads leaves the younger requests in flight — any branch around a load makes the compiler wait for everything), and
// the four chains are walked in step (a row whose chain has ended reads entry 0, one line for the whole wave).
template <int KW, bool HASV, bool OCC>
__global__ __launch_bounds__(256) void join_count_kernel(const uint64_t* __restrict__ ent, const uint64_t* __restrict__ head,
                                                         const uint64_t* __restrict__ occ, int shift,
                                                         const uint64_t* __restrict__ pkeys, const uint8_t* __restrict__ pvalid, int64_t nfull,
                                                         uint8_t* __restrict__ cnt8, uint32_t* __restrict__ firstm, uint32_t* __restrict__ tile_sum,
                                                         unsigned long long* total) {
  constexpr int ES = KW * 2;
  typedef uint64_t jk_u64x2 __attribute__((ext_vector_type(2)));
  uint64_t local = 0;
  jk_u64x2 kv[2 * KW], kvn[2 * KW];
  uint32_t vbyte = 0xFFu, vbyte_n = 0xFFu;   // the validity byte that holds this thread's nibble
  const int lane = lane_id();
  const int64_t tstride = (int64_t)gridDim.x * 4;
  int64_t t = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (t >= nfull) return;   // (wave-uniform; nothing below crosses a wave)
  {
    const int64_t i0 = t * JOIN_TILE + lane * 4;
#pragma unroll
    for (int q = 0; q < 2 * KW; ++q) kv[q] = __builtin_nontemporal_load((const jk_u64x2*)(pkeys + i0 * KW) + q);
    if (HASV) vbyte = pvalid[i0 >> 3];
  }
  for (; t < nfull; t += tstride) {
    const int64_t i0 = t * JOIN_TILE + lane * 4;
    const int64_t tn = t + tstride < nfull ? t + tstride : t;   // (past the end: this tile again, the lines are in the cache)
    const int64_t i0n = tn * JOIN_TILE + lane * 4;
    uint64_t k[4][KW];
#pragma unroll
    for (int q = 0; q < 2 * KW; ++q) { (&k[0][0])[2 * q] = kv[q].x; (&k[0][0])[2 * q + 1] = kv[q].y; }
    uint32_t vmask = HASV ? (vbyte >> (i0 & 4)) & 15u : 15u;
    uint64_t hd[4], idx[4];
    uint32_t tag[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) idx[r] = join_bucket<KW>(k[r], shift, &tag[r]);
    if (OCC) {   // one bit per bucket, resident in the L2: a random probe of an empty bucket never leaves the XCD
      uint64_t ow[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) ow[r] = occ[(vmask >> r) & 1 ? idx[r] >> 6 : 0];
#pragma unroll
      for (int r = 0; r < 4; ++r) if (!((ow[r] >> (idx[r] & 63)) & 1)) vmask &= ~(1u << r);
    }
    // (rows that are not probed read head 0: one line for the wave, no branch)
#pragma unroll
    for (int r = 0; r < 4; ++r) hd[r] = head[(vmask >> r) & 1 ? idx[r] : 0];
#pragma unroll
    for (int q = 0; q < 2 * KW; ++q) kvn[q] = __builtin_nontemporal_load((const jk_u64x2*)(pkeys + i0n * KW) + q);
    if (HASV) vbyte_n = pvalid[i0n >> 3];
    uint32_t e[4], c[4] = {0, 0, 0, 0}, last[4] = {0, 0, 0, 0};
#pragma unroll
    for (int r = 0; r < 4; ++r) e[r] = ((vmask >> r) & (uint32_t)(hd[r] >> (32 + tag[r])) & 1u) ? (uint32_t)hd[r] : 0u;   // not probed, empty bucket or tag miss (fixed_keys.rs:139-142)
    while (e[0] | e[1] | e[2] | e[3]) {
      const uint64_t* p[4];
      ulonglong2 v[4], v1[4];
      uint64_t link[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) { p[r] = ent + (uint64_t)(e[r] ? e[r] - 1 : 0u) * ES; v[r] = *(const ulonglong2*)p[r]; }
      if (KW == 4) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v1[r] = *(const ulonglong2*)(p[r] + 2);
      }
      if (KW > 1) {
#pragma unroll
        for (int r = 0; r < 4; ++r) link[r] = p[r][KW];
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        bool eq;
        if (KW == 1) { eq = v[r].x == k[r][0]; link[r] = v[r].y; }
        else {
          eq = v[r].x == k[r][0] && v[r].y == k[r][1];
          if (KW == 4) eq = eq && v1[r].x == k[r][2] && v1[r].y == k[r][3];
        }
        const bool live = e[r] != 0;
        c[r] += live && eq;
        last[r] = live && eq ? e[r] - 1 : last[r];
        e[r] = live ? (uint32_t)link[r] : 0u;
      }
    }
    uint32_t packed = 0, tsum = 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (c[r] && cnt8) firstm[i0 + r] = last[r];
      packed |= (c[r] < 255u ? c[r] : 255u) << (8 * r);
      tsum += c[r];
    }
    if (cnt8) *(uint32_t*)(cnt8 + i0) = packed;
    local += tsum;
    if (tile_sum) join_tile_total(tsum, tile_sum, t);
#pragma unroll
    for (int q = 0; q < 2 * KW; ++q) kv[q] = kvn[q];
    vbyte = vbyte_n;
  }
  local = wave_sum_u64(local);
  if (lane_id() == 0 && local) atomicAdd(total, (unsigned long long)local);
}

// any tile range [t0, ntiles) of any column (the ragged last tile; every tile of a key column that starts off a 16-byte boundary)
template <int KW>
__global__ __launch_bounds__(256) void join_count_any_kernel(const uint64_t* ent, const uint64_t* head, int shift, const uint64_t* pkeys,
                                                             const uint8_t* pvalid, int64_t n, int64_t t0, uint8_t* cnt8, uint32_t* firstm,
                                                             uint32_t* tile_sum, unsigned long long* total) {
  const int64_t ntiles = (n + JOIN_TILE - 1) / JOIN_TILE;
  uint64_t local = 0;
  for (int64_t t = t0 + (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); t < ntiles; t += (int64_t)gridDim.x * 4) {
    uint32_t tsum = 0;
    for (int r = 0; r < 4; ++r) {
      const int64_t i = t * JOIN_TILE + lane_id() * 4 + r;
      uint32_t c = 0, last = 0;
      if (i < n && (!pvalid || bit_get(pvalid, i))) {
        uint64_t k[KW];
#pragma unroll
        for (int w = 0; w < KW; ++w) k[w] = pkeys[i * KW + w];
        c = join_walk<KW>(ent, head, shift, k, &last);
      }
      if (i < n && cnt8) {
        cnt8[i] = (uint8_t)(c < 255u ? c : 255u);
        if (c) firstm[i] = last;
      }
      tsum += c;
    }
    local += tsum;
    if (tile_sum) join_tile_total(tsum, tile_sum, t);
  }
  local = wave_sum_u64(local);
  if (lane_id() == 0 && local) atomicAdd(total, (unsigned long long)local);
}

template <int KW>
__global__ __launch_bounds__(256) void join_emit_kernel(const uint64_t* ent, const uint64_t* head, int shift,
                                                        const uint64_t* pkeys, int64_t n, const uint8_t* cnt8,
                                                        const uint32_t* firstm, const uint32_t* tile_sum, const uint64_t* tile_off, uint32_t* out_p,
                                                        uint32_t* out_b, int64_t max_pairs, int G) {
  constexpr int ES = KW * 2;
  const int64_t ntiles = (n + JOIN_TILE - 1) / JOIN_TILE;
  const int lane = lane_id();
  const int64_t tstride = (int64_t)gridDim.x * 4;
  // (the counts of the NEXT tile are requested before this tile is scanned: the loop is a chain of dependent round trips)
  auto load_counts = [&](int64_t tt) -> uint32_t {
    const int64_t j0 = tt * JOIN_TILE + lane * 4;
    if (j0 + 4 <= n) return *(const uint32_t*)(cnt8 + j0);
    uint32_t pk = 0;
    for (int r = 0; r < 4 && j0 + r < n; ++r) pk |= (uint32_t)cnt8[j0 + r] << (8 * r);
    return pk;
  };
  // Round 6: a wave takes 64 CONSECUTIVE tiles at a time — one coalesced load brings their 64 pair counts, a ballot names the tiles that
  // have pairs, and only those are visited (their row counts and output offset requested one tile ahead). Rounds 3-5 walked every tile
  // with a dependent load each: a selective join (TPC-H Q3's lineitem probe: 3 M pairs in 2.3 M tiles of 600 M rows) spent its emit
  // pass — 0.8 ms — on tiles with nothing to emit. G (1 .. 64, chosen by the host) = consecutive tiles per wave and step: 64 when there
  // are tiles for every wave of the grid many times over, fewer for small probe blocks — a block of 196 tiles whose every row matches
  // thousands of build rows must still spread over 196 waves, not sit on four.
  const int64_t wave0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  (void)tstride;
  for (int64_t tb = wave0 * G; tb < ntiles; tb += (int64_t)gridDim.x * 4 * G) {
    const uint32_t my_ts = (lane < G && tb + lane < ntiles) ? tile_sum[tb + lane] : 0u;
    uint64_t todo = __ballot(my_ts != 0);
    if (todo == 0) continue;
    int nxt = __ffsll((long long)todo) - 1;
    uint32_t packed_next = load_counts(tb + nxt);
    uint64_t off_next = tile_off[tb + nxt];
   while (todo) {
    const int cur = nxt;
    const int64_t t = tb + cur;
    const uint32_t packed = packed_next;
    const uint64_t toff = off_next;
    todo &= todo - 1;
    if (todo) { nxt = __ffsll((long long)todo) - 1; packed_next = load_counts(tb + nxt); off_next = tile_off[tb + nxt]; }
    const int64_t i0 = t * JOIN_TILE + lane * 4;
    uint32_t c[4], mine = 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      c[r] = (packed >> (8 * r)) & 255u;
      if (c[r] == 255u) {   // saturated: count the chain again
        uint64_t k[KW];
#pragma unroll
        for (int w = 0; w < KW; ++w) k[w] = pkeys[(i0 + r) * KW + w];
        uint32_t last;
        c[r] = join_walk<KW>(ent, head, shift, k, &last);
      }
      mine += c[r];
    }
    // exclusive prefix of the per-lane sums over the wave = row order
    uint32_t incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(incl, d, 64); if (lane >= d) incl += o; }
    uint64_t o = toff + incl - mine;
    if (mine == 0) continue;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const uint32_t cr = c[r];
      const int64_t i = i0 + r;
      if (cr == 0) continue;
      if ((int64_t)(o + cr) > max_pairs) { o += cr; continue; }
      if (cr == 1) {  // unique build key: nothing to walk
        out_p[o] = (uint32_t)i;
        out_b[o] = firstm[i];
        o += 1;
        continue;
      }
      uint64_t k[KW];
#pragma unroll
      for (int w = 0; w < KW; ++w) k[w] = pkeys[i * KW + w];
      uint32_t tag;
      const uint64_t idx = join_bucket<KW>(k, shift, &tag);
      uint32_t m = 0;
      for (uint32_t e = (uint32_t)head[idx]; e;) {
        const uint64_t* p = ent + (uint64_t)(e - 1) * ES;
        bool eq = p[0] == k[0];
#pragma unroll
        for (int w = 1; w < KW; ++w) eq = eq && p[w] == k[w];
        if (eq && m < cr) {
          // the chain runs from the newest build row to the oldest (rows are prepended): written back to front the segment comes out
          // ascending, or nearly so (the build's atomics race within a wave) — the insertion pass below then moves almost nothing.
          // (Rounds 1-5 inserted from the front: a probe row with 4,500 matches cost 10 M moves — a 50 k x 50 k join on 10 keys took 74 s.)
          out_b[o + cr - 1 - m] = e - 1;
          out_p[o + m] = (uint32_t)i;
          ++m;
        }
        e = (uint32_t)p[KW];
      }
      for (uint32_t a = 1; a < cr; ++a) {   // keep this probe row's segment ascending by build row
        const uint32_t b = out_b[o + a];
        uint32_t j = a;
        while (j > 0 && out_b[o + j - 1] > b) { out_b[o + j] = out_b[o + j - 1]; --j; }
        out_b[o + j] = b;
      }
      o += cr;
    }
   }
  }
}

// matched[i] = probe row i has at least one build match (semi / anti / left-outer bookkeeping)
template <int KW>
__global__ __launch_bounds__(256) void join_mark_kernel(const uint64_t* ent, const uint64_t* head, int shift,
                                                        const uint64_t* pkeys, const uint8_t* pvalid, int64_t n,
                                                        uint8_t* out_bitmap, unsigned long long* total) {
  const int64_t n_pad = (n + 63) & ~63LL;
  uint64_t local = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_pad; i += (int64_t)gridDim.x * blockDim.x) {
    bool hit = false;
    if (i < n && (!pvalid || bit_get(pvalid, i))) {
      uint64_t k[KW];
#pragma unroll
      for (int w = 0; w < KW; ++w) k[w] = pkeys[i * KW + w];
      uint32_t last;
      hit = join_walk<KW>(ent, head, shift, k, &last) != 0;
    }
    const uint64_t m = __ballot(hit);
    const int l = lane_id();
    if ((l & 7) == 0 && i < n) out_bitmap[i >> 3] = (uint8_t)(m >> l);
    local += hit;
  }
  local = wave_sum_u64(local);
  if (lane_id() == 0 && local) atomicAdd(total, (unsigned long long)local);
}

__global__ __launch_bounds__(256) void join_occ_kernel(const uint64_t* head, int64_t buckets, uint64_t* occ) {
  for (int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; b < buckets; b += (int64_t)gridDim.x * blockDim.x) {
    const uint64_t m = __ballot((uint32_t)head[b] != 0u);
    if (lane_id() == 0) occ[b >> 6] = m;
  }
}

int32_t ensure_probe_scratch(dbhip_join* j, int64_t n) {
  if (j->scratch_rows >= (size_t)n) return DBHIP_OK;
  if (j->cnt) {
    DBHIP_CHECK(hipDeviceSynchronize());
    (void)dbhip_free(j->cnt); (void)dbhip_free(j->firstm); (void)dbhip_free(j->tsum); (void)dbhip_free(j->off); (void)dbhip_free(j->blk);
    j->cnt = nullptr; j->firstm = nullptr; j->tsum = nullptr; j->off = nullptr; j->blk = nullptr;   // (an allocation below may fail: no dangling pointers)
    j->scratch_rows = 0;
  }
  size_t cap = (size_t)n + (n >> 3) + 1024;
  const size_t tiles = cap / JOIN_TILE + 2;
  DBHIP_TRY(dbhip_alloc(cap, (void**)&j->cnt));
  DBHIP_TRY(dbhip_alloc(cap * 4, (void**)&j->firstm));
  DBHIP_TRY(dbhip_alloc(tiles * 4, (void**)&j->tsum));
  DBHIP_TRY(dbhip_alloc(tiles * 8, (void**)&j->off));
  DBHIP_TRY(dbhip_alloc((tiles / SCAN_TILE + 2) * 8, (void**)&j->blk));
  j->scratch_rows = cap;
  return DBHIP_OK;
}

// ---------------------------------------------------------------------------
// a14: packed fixed-width keys (HashMethodFixedKeys::build_keys_vec, method_fixed_keys.rs:58-78;
// KeysVec layout :310-403): columns stably sorted by byte width, values little endian back to
// back, then one null byte per nullable column (1 = NULL, the value bytes of a NULL stay zero).
// ---------------------------------------------------------------------------
constexpr int PK_MAX_COLS = 16;
struct PackCols {
  const void* data[PK_MAX_COLS];
  const uint8_t* validity[PK_MAX_COLS];
  int64_t voff[PK_MAX_COLS];
  int32_t type[PK_MAX_COLS];
  int32_t src_bytes[PK_MAX_COLS];   // bytes per element in the source buffer
  int32_t key_bytes[PK_MAX_COLS];   // bytes the value occupies in the key (numeric_byte_size)
  int32_t offset[PK_MAX_COLS];      // byte offset of the value in the key
  int32_t null_offset[PK_MAX_COLS]; // byte offset of the null flag, -1 = not nullable
  int32_t is_scalar[PK_MAX_COLS];
  int n;
};

__global__ __launch_bounds__(256) void pack_keys_kernel(PackCols P, int64_t n, int key_bytes, uint8_t* out,
                                                        uint8_t* all_valid) {
  const int64_t n_pad = (n + 63) & ~63LL;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_pad; i += (int64_t)gridDim.x * blockDim.x) {
    bool allv = true;
    if (i < n) {
      uint64_t w[4] = {0, 0, 0, 0};  // the key as little-endian words (up to 32 bytes)
      for (int c = 0; c < P.n; ++c) {
        const int64_t j = P.is_scalar[c] ? 0 : i;
        const bool valid = !P.validity[c] || bit_get(P.validity[c], P.voff[c] + j);
        allv &= valid;
        if (!valid) {
          const int o = P.null_offset[c];
          if (o >= 0) w[o >> 3] |= 1ULL << (8 * (o & 7));
          continue;
        }
        uint64_t v0 = 0, v1 = 0;
        const uint8_t* src = (const uint8_t*)P.data[c];
        switch (P.src_bytes[c]) {
          case 0: v0 = bit_get(src, j); break;  // Boolean
          case 1: v0 = src[j]; break;
          case 2: v0 = ((const uint16_t*)src)[j]; break;
          case 4: v0 = ((const uint32_t*)src)[j]; break;
          case 8: v0 = ((const uint64_t*)src)[j]; break;
          default: v0 = ((const uint64_t*)src)[2 * j]; v1 = ((const uint64_t*)src)[2 * j + 1]; break;
        }
        // DecimalView<FROM, TO>: a wider/narrower store than the precision's carrier is converted
        if (P.key_bytes[c] == 16 && P.src_bytes[c] == 8) v1 = (v0 >> 63) ? ~0ULL : 0ULL;  // sign extend i64 -> i128
        const int o = P.offset[c], kb = P.key_bytes[c];
        // place `kb` little-endian bytes at byte offset o (may straddle words)
        for (int part = 0; part < (kb + 7) / 8; ++part) {
          uint64_t v = part == 0 ? v0 : v1;
          const int nb = kb - 8 * part < 8 ? kb - 8 * part : 8;
          if (nb < 8) v &= (1ULL << (8 * nb)) - 1;
          const int bo = o + 8 * part;
          const int wi = bo >> 3, sh = 8 * (bo & 7);
          w[wi] |= v << sh;
          if (sh && wi + 1 < 4 && nb * 8 + sh > 64) w[wi + 1] |= v >> (64 - sh);
        }
      }
      switch (key_bytes) {
        case 1: out[i] = (uint8_t)w[0]; break;
        case 2: ((uint16_t*)out)[i] = (uint16_t)w[0]; break;
        case 4: ((uint32_t*)out)[i] = (uint32_t)w[0]; break;
        case 8: ((uint64_t*)out)[i] = w[0]; break;
        case 16: ((uint64_t*)out)[2 * i] = w[0]; ((uint64_t*)out)[2 * i + 1] = w[1]; break;
        default:
          for (int k = 0; k < 4; ++k) ((uint64_t*)out)[4 * i + k] = w[k];
          break;
      }
    } else {
      allv = false;
    }
    if (all_valid) {
      const uint64_t m = __ballot(allv);
      const int l = lane_id();
      if ((l & 7) == 0 && i < n) all_valid[i >> 3] = (uint8_t)(m >> l);
    }
  }
}

// numeric_byte_size (src/query/expression/src/types.rs:606-633); 0 = not a fixed-width key type
int key_type_bytes(const dbhip_col& c, int* src_bytes) {
  switch (c.type) {
    case DBHIP_T_I8: case DBHIP_T_U8: *src_bytes = 1; return 1;
    case DBHIP_T_I16: case DBHIP_T_U16: *src_bytes = 2; return 2;
    case DBHIP_T_I32: case DBHIP_T_U32: case DBHIP_T_F32: case DBHIP_T_DATE: *src_bytes = 4; return 4;
    case DBHIP_T_I64: case DBHIP_T_U64: case DBHIP_T_F64: case DBHIP_T_TIMESTAMP: *src_bytes = 8; return 8;
    case DBHIP_T_DEC64: *src_bytes = 8; return c.precision <= 18 ? 8 : 16;
    case DBHIP_T_DEC128: *src_bytes = 16; return c.precision <= 18 ? 8 : 16;
    default: *src_bytes = 0; return 0;
  }
}

int32_t plan_pack(const dbhip_col* cols, int ncols, PackCols* P, int* total_bytes) {
  if (ncols < 1 || ncols > PK_MAX_COLS) {
    set_error("pack_keys: 1..%d key columns", PK_MAX_COLS);
    return DBHIP_ERR_INVALID;
  }
  int order[PK_MAX_COLS], kb[PK_MAX_COLS], sb[PK_MAX_COLS];
  for (int c = 0; c < ncols; ++c) {
    order[c] = c;
    kb[c] = key_type_bytes(cols[c], &sb[c]);
    if (kb[c] == 0) {
      set_error("pack_keys: column %d of type %d is not a fixed-width key (HashMethodSerializer stays on the CPU)", c, cols[c].type);
      return DBHIP_ERR_UNSUPPORTED;
    }
  }
  std::stable_sort(order, order + ncols, [&](int a, int b) { return kb[a] < kb[b]; });  // sort_by_key is stable
  int off = 0;
  memset(P, 0, sizeof(*P));
  P->n = ncols;
  for (int s = 0; s < ncols; ++s) {
    const int c = order[s];
    P->data[s] = cols[c].data; P->validity[s] = cols[c].validity; P->voff[s] = cols[c].validity_offset;
    P->type[s] = cols[c].type; P->src_bytes[s] = sb[c]; P->key_bytes[s] = kb[c]; P->is_scalar[s] = cols[c].is_scalar;
    P->offset[s] = off;
    off += kb[c];
  }
  for (int s = 0; s < ncols; ++s) P->null_offset[s] = P->validity[s] ? off++ : -1;
  *total_bytes = off;
  return DBHIP_OK;
}

int method_bytes(int total) {  // choose_hash_method_with_types (kernels/group_by.rs:70-78)
  if (total <= 1) return 1;
  if (total <= 2) return 2;
  if (total <= 4) return 4;
  if (total <= 8) return 8;
  if (total <= 16) return 16;
  if (total <= 32) return 32;
  return 0;
}

}  // namespace

extern "C" {

int32_t dbhip_keys_method(const dbhip_col* cols, int32_t ncols, int32_t* out_key_bytes_host) {
  DBHIP_REQUIRE(cols && out_key_bytes_host, "dbhip_keys_method: NULL argument");
  PackCols P;
  int total = 0;
  int32_t rc = plan_pack(cols, ncols, &P, &total);
  if (rc == DBHIP_ERR_UNSUPPORTED) { *out_key_bytes_host = 0; return DBHIP_OK; }
  if (rc) return rc;
  *out_key_bytes_host = method_bytes(total);
  return DBHIP_OK;
}

int32_t dbhip_pack_keys(const dbhip_col* cols, int32_t ncols, int64_t n, int32_t key_bytes, void* out_keys,
                        uint8_t* out_all_valid, void* stream) {
  DBHIP_REQUIRE(cols, "dbhip_pack_keys: NULL columns");
  PackCols P;
  int total = 0;
  int32_t rc = plan_pack(cols, ncols, &P, &total);
  if (rc) return rc;
  if (!(key_bytes == 1 || key_bytes == 2 || key_bytes == 4 || key_bytes == 8 || key_bytes == 16 || key_bytes == 32) ||
      key_bytes < total) {
    set_error("dbhip_pack_keys: %d key bytes do not fit a %d-byte key", total, key_bytes);
    return DBHIP_ERR_INVALID;
  }
  if (n == 0) return DBHIP_OK;
  DBHIP_REQUIRE(out_keys, "dbhip_pack_keys: NULL out");
  hipStream_t s = resolve_stream(stream);
  hipLaunchKernelGGL(pack_keys_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, P, n, key_bytes, (uint8_t*)out_keys,
                     out_all_valid);
  DBHIP_LAUNCH_CHECK();
  return DBHIP_OK;
}

int32_t dbhip_join_create_keys(int64_t expected_build_rows, int32_t key_bytes, dbhip_join** out_host) {
  DBHIP_REQUIRE(out_host, "dbhip_join_create: NULL out");
  DBHIP_REQUIRE(key_bytes == 8 || key_bytes == 16 || key_bytes == 32, "dbhip_join_create: key width must be 8, 16 or 32 bytes (narrower keys are zero-extended by dbhip_pack_keys)");
  dbhip_join* j = new (std::nothrow) dbhip_join();
  DBHIP_REQUIRE(j, "dbhip_join_create: out of host memory");
  memset(j, 0, sizeof(*j));
  j->kw = key_bytes / 8;
  j->es = j->kw * 2;
  j->cap_rows = expected_build_rows > 1024 ? expected_build_rows : 1024;
  int32_t rc = dbhip_alloc((size_t)j->cap_rows * j->es * 8, (void**)&j->ent);
  if (!rc) rc = dbhip_alloc(8, (void**)&j->total_dev);
  if (rc) {  // nothing half-built is handed out or leaked
    if (j->ent) (void)dbhip_free(j->ent);
    delete j;
    return rc;
  }
  *out_host = j;
  return DBHIP_OK;
}

int32_t dbhip_join_create(int64_t expected_build_rows, dbhip_join** out_host) {
  return dbhip_join_create_keys(expected_build_rows, 8, out_host);
}

int32_t dbhip_join_add_build(dbhip_join* j, const void* keys, const uint8_t* validity, int64_t n,
                             void* stream) {
  DBHIP_REQUIRE(j && !j->finalized, "dbhip_join_add_build: table missing or already finalized");
  if (n == 0) return DBHIP_OK;
  DBHIP_REQUIRE(keys, "dbhip_join_add_build: NULL keys");
  DBHIP_REQUIRE(j->nrows + n < 0xFFFFFFFFLL, "dbhip_join_add_build: more than 2^32-1 build rows");
  hipStream_t s = resolve_stream(stream);
  if (j->nrows + n > j->cap_rows) {  // grow the chunk store (BasicHashJoin::add_block squashes chunks)
    int64_t nc = j->cap_rows * 2 > j->nrows + n ? j->cap_rows * 2 : j->nrows + n;
    uint64_t* ne;
    DBHIP_TRY(dbhip_alloc((size_t)nc * j->es * 8, (void**)&ne));
    DBHIP_CHECK(hipMemcpyAsync(ne, j->ent, (size_t)j->nrows * j->es * 8, hipMemcpyDeviceToDevice, s));
    DBHIP_CHECK(hipStreamSynchronize(s));
    (void)dbhip_free(j->ent);
    j->ent = ne; j->cap_rows = nc;
  }
  uint64_t* dst = j->ent + (size_t)j->nrows * j->es;
  if (j->kw == 1)
    hipLaunchKernelGGL(join_copy_kernel<1>, dim3(grid_for(n, 256)), dim3(256), 0, s, (const uint64_t*)keys, validity, n, dst);
  else if (j->kw == 2)
    hipLaunchKernelGGL(join_copy_kernel<2>, dim3(grid_for(n, 256)), dim3(256), 0, s, (const uint64_t*)keys, validity, n, dst);
  else
    hipLaunchKernelGGL(join_copy_kernel<4>, dim3(grid_for(n, 256)), dim3(256), 0, s, (const uint64_t*)keys, validity, n, dst);
  DBHIP_LAUNCH_CHECK();
  j->nrows += n;
  return DBHIP_OK;
}

int32_t dbhip_join_finalize(dbhip_join* j, void* stream) {
  DBHIP_REQUIRE(j && !j->finalized, "dbhip_join_finalize: table missing or already finalized");
  hipStream_t s = resolve_stream(stream);
  int64_t cap = 1024;
  while (cap < j->nrows * 2) cap <<= 1;  // hashjoin_hashtable.rs:95-108
  j->buckets = cap;
  j->shift = 64 - __builtin_ctzll((unsigned long long)cap);
  if (j->kw == 1 && exp_env("DBHIP_JOIN_D")) j->shift |= atoi(exp_env("DBHIP_JOIN_D")) << 8;
  DBHIP_TRY(dbhip_alloc((size_t)cap * 8, (void**)&j->head));
  DBHIP_CHECK(hipMemsetAsync(j->head, 0, (size_t)cap * 8, s));
  if (j->nrows) {
    if (j->kw == 1)
      hipLaunchKernelGGL(join_build_kernel<1>, dim3(grid_for(j->nrows, 256)), dim3(256), 0, s, j->ent, j->nrows, j->head, j->shift);
    else if (j->kw == 2)
      hipLaunchKernelGGL(join_build_kernel<2>, dim3(grid_for(j->nrows, 256)), dim3(256), 0, s, j->ent, j->nrows, j->head, j->shift);
    else
      hipLaunchKernelGGL(join_build_kernel<4>, dim3(grid_for(j->nrows, 256)), dim3(256), 0, s, j->ent, j->nrows, j->head, j->shift);
    DBHIP_LAUNCH_CHECK();
    // JOIN_OCC_MIN_BUCKETS: 2^20 heads = 8 MB, past what one XCD's L2 keeps; DBHIP_JOIN_OCC=0 turns the filter off (measurements)
    static const bool occ_on = !(exp_env("DBHIP_JOIN_OCC") && exp_env("DBHIP_JOIN_OCC")[0] == '0');
    if (occ_on && cap >= (1 << 20)) {
      DBHIP_TRY(dbhip_alloc((size_t)cap / 8, (void**)&j->occ));
      hipLaunchKernelGGL(join_occ_kernel, dim3(grid_for(cap, 256)), dim3(256), 0, s, j->head, cap, j->occ);
      DBHIP_LAUNCH_CHECK();
    }
  }
  j->finalized = true;
  return DBHIP_OK;
}

// count pass + exclusive scan of one probe block into the handle's scratch
static int32_t join_count_block(dbhip_join* j, const void* keys, const uint8_t* validity, int64_t n, hipStream_t s,
                                uint64_t* total_host) {
  int32_t rc = ensure_probe_scratch(j, n);
  if (rc) return rc;
  j->prepared = false;
  DBHIP_CHECK(hipMemsetAsync(j->total_dev, 0, 8, s));
  const int64_t ntiles = ceil_div(n, JOIN_TILE);
  const int64_t nfull = ((uintptr_t)keys & 15) == 0 ? n / JOIN_TILE : 0;
  unsigned long long* tot = (unsigned long long*)j->total_dev;
  if (nfull) {
    const int grid = (int)(nfull < 4 * 4096 ? ceil_div(nfull, 4) : 4096);   // four wave tiles per workgroup at a time
#define JOIN_COUNT_LAUNCH(KW, HASV, OCC)                                                                                         \
  hipLaunchKernelGGL((join_count_kernel<KW, HASV, OCC>), dim3(grid), dim3(256), 0, s, j->ent, j->head, j->occ, j->shift,       \
                     (const uint64_t*)keys, validity, nfull, j->cnt, j->firstm, j->tsum, tot)
#define JOIN_COUNT_KW(KW)                                                                                                       \
  do {                                                                                                                          \
    if (validity) { if (j->occ) JOIN_COUNT_LAUNCH(KW, true, true); else JOIN_COUNT_LAUNCH(KW, true, false); }                   \
    else { if (j->occ) JOIN_COUNT_LAUNCH(KW, false, true); else JOIN_COUNT_LAUNCH(KW, false, false); }                          \
  } while (0)
    if (j->kw == 1) JOIN_COUNT_KW(1);
    else if (j->kw == 2) JOIN_COUNT_KW(2);
    else JOIN_COUNT_KW(4);
#undef JOIN_COUNT_KW
#undef JOIN_COUNT_LAUNCH
    DBHIP_LAUNCH_CHECK();
  }
  if (nfull < ntiles) {
    const int64_t rest = ntiles - nfull;
    const int grid = (int)(rest < 4 * 4096 ? ceil_div(rest, 4) : 4096);
    if (j->kw == 1)
      hipLaunchKernelGGL(join_count_any_kernel<1>, dim3(grid), dim3(256), 0, s, j->ent, j->head, j->shift, (const uint64_t*)keys, validity, n, nfull, j->cnt, j->firstm, j->tsum, tot);
    else if (j->kw == 2)
      hipLaunchKernelGGL(join_count_any_kernel<2>, dim3(grid), dim3(256), 0, s, j->ent, j->head, j->shift, (const uint64_t*)keys, validity, n, nfull, j->cnt, j->firstm, j->tsum, tot);
    else
      hipLaunchKernelGGL(join_count_any_kernel<4>, dim3(grid), dim3(256), 0, s, j->ent, j->head, j->shift, (const uint64_t*)keys, validity, n, nfull, j->cnt, j->firstm, j->tsum, tot);
    DBHIP_LAUNCH_CHECK();
  }
  rc = dbscan::exclusive_scan_u32(j->tsum, ntiles, j->blk, j->off, s);
  if (rc) return rc;
  DBHIP_CHECK(hipMemcpyAsync(total_host, j->total_dev, 8, hipMemcpyDeviceToHost, s));
  DBHIP_CHECK(hipStreamSynchronize(s));
  j->prep_keys = keys; j->prep_valid = validity; j->prep_n = n; j->prep_total = *total_host; j->prep_stream = s;
  j->prepared = true;
  return DBHIP_OK;
}

int32_t dbhip_join_probe_count(dbhip_join* j, const void* keys, const uint8_t* validity, int64_t n,
                               uint64_t* out_total_host, void* stream) {
  DBHIP_REQUIRE(j && j->finalized && out_total_host, "dbhip_join_probe_count: table not finalized / NULL out");
  DBHIP_REQUIRE(n < 0xFFFFFFFFLL, "dbhip_join_probe_count: more than 2^32-1 probe rows in one block");
  *out_total_host = 0;
  if (n == 0) return DBHIP_OK;
  return join_count_block(j, keys, validity, n, resolve_stream(stream), out_total_host);
}

int32_t dbhip_join_probe_mark(dbhip_join* j, const void* keys, const uint8_t* validity, int64_t n,
                              uint8_t* out_matched_bitmap, uint64_t* out_n_matched_host, void* stream) {
  DBHIP_REQUIRE(j && j->finalized, "dbhip_join_probe_mark: table not finalized");
  hipStream_t s = resolve_stream(stream);
  if (out_n_matched_host) *out_n_matched_host = 0;
  if (n == 0) return DBHIP_OK;
  DBHIP_REQUIRE(keys && out_matched_bitmap, "dbhip_join_probe_mark: NULL argument");
  DBHIP_CHECK(hipMemsetAsync(j->total_dev, 0, 8, s));
  const int grid = grid_for(n, 256);
  if (j->kw == 1)
    hipLaunchKernelGGL(join_mark_kernel<1>, dim3(grid), dim3(256), 0, s, j->ent, j->head, j->shift, (const uint64_t*)keys,
                       validity, n, out_matched_bitmap, (unsigned long long*)j->total_dev);
  else if (j->kw == 2)
    hipLaunchKernelGGL(join_mark_kernel<2>, dim3(grid), dim3(256), 0, s, j->ent, j->head, j->shift, (const uint64_t*)keys,
                       validity, n, out_matched_bitmap, (unsigned long long*)j->total_dev);
  else
    hipLaunchKernelGGL(join_mark_kernel<4>, dim3(grid), dim3(256), 0, s, j->ent, j->head, j->shift, (const uint64_t*)keys,
                       validity, n, out_matched_bitmap, (unsigned long long*)j->total_dev);
  DBHIP_LAUNCH_CHECK();
  if (out_n_matched_host) {
    DBHIP_CHECK(hipMemcpyAsync(out_n_matched_host, j->total_dev, 8, hipMemcpyDeviceToHost, s));
    DBHIP_CHECK(hipStreamSynchronize(s));
  }
  return DBHIP_OK;
}

int32_t dbhip_join_probe(dbhip_join* j, const void* keys, const uint8_t* validity, int64_t n,
                         uint32_t* out_probe_idx, uint32_t* out_build_row, int64_t max_pairs,
                         uint64_t* out_n_pairs_host, void* stream) {
  DBHIP_REQUIRE(j && j->finalized && out_n_pairs_host, "dbhip_join_probe: table not finalized / NULL out");
  DBHIP_REQUIRE(n < 0xFFFFFFFFLL, "dbhip_join_probe: more than 2^32-1 probe rows in one block");
  hipStream_t s = resolve_stream(stream);
  *out_n_pairs_host = 0;
  if (n == 0) return DBHIP_OK;
  const int64_t ntiles_p = ceil_div(n, JOIN_TILE);
  const int grid = (int)(ntiles_p < 4 * 4096 ? ceil_div(ntiles_p, 4) : 4096);   // the emit pass walks tiles like the count pass
  int G = 1;   // tiles per wave and step (join_emit_kernel)
  while (G < 64 && (int64_t)grid * 4 * (G * 2) <= ntiles_p) G *= 2;
  uint64_t total = 0;
  int32_t rc;
  if (j->prepared && j->prep_keys == keys && j->prep_valid == validity && j->prep_n == n && j->prep_stream == s) {
    total = j->prep_total;  // counted by dbhip_join_probe_count just before (columns are immutable between the two)
  } else if ((rc = join_count_block(j, keys, validity, n, s, &total))) {
    return rc;
  }
  j->prepared = false;
  *out_n_pairs_host = total;
  if ((int64_t)total > max_pairs) {
    set_error("dbhip_join_probe: %llu pairs do not fit max_pairs=%lld (call dbhip_join_probe_count first)",
              (unsigned long long)total, (long long)max_pairs);
    return DBHIP_ERR_CAPACITY;
  }
  if (total) {
    DBHIP_REQUIRE(out_probe_idx && out_build_row, "dbhip_join_probe: NULL output");
    if (j->kw == 1)
      hipLaunchKernelGGL(join_emit_kernel<1>, dim3(grid), dim3(256), 0, s, j->ent, j->head, j->shift, (const uint64_t*)keys, n,
                         j->cnt, j->firstm, j->tsum, j->off, out_probe_idx, out_build_row, max_pairs, G);
    else if (j->kw == 2)
      hipLaunchKernelGGL(join_emit_kernel<2>, dim3(grid), dim3(256), 0, s, j->ent, j->head, j->shift, (const uint64_t*)keys, n,
                         j->cnt, j->firstm, j->tsum, j->off, out_probe_idx, out_build_row, max_pairs, G);
    else
      hipLaunchKernelGGL(join_emit_kernel<4>, dim3(grid), dim3(256), 0, s, j->ent, j->head, j->shift, (const uint64_t*)keys, n,
                         j->cnt, j->firstm, j->tsum, j->off, out_probe_idx, out_build_row, max_pairs, G);
    DBHIP_LAUNCH_CHECK();
  }
  return DBHIP_OK;
}

// mark[b >> 5] |= 1 << (b & 31) for every build row b of the pairs
static __global__ __launch_bounds__(256) void join_mark_build_kernel(const uint32_t* build_rows, int64_t n, int64_t nrows, uint32_t* mark) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const uint32_t b = build_rows[i];
    if ((int64_t)b < nrows) atomicOr(&mark[b >> 5], 1u << (b & 31));
  }
}

int32_t dbhip_join_mark_build(dbhip_join* j, const uint32_t* build_rows, int64_t n_pairs, void* stream) {
  DBHIP_REQUIRE(j && j->finalized && (build_rows || n_pairs == 0), "dbhip_join_mark_build: needs a finalized table and the pairs' build rows");
  hipStream_t s = resolve_stream(stream);
  if (!j->bmark) {
    const size_t bytes = (size_t)ceil_div(j->nrows > 0 ? j->nrows : 1, 64) * 8;
    int32_t rc = dbhip_alloc(bytes, (void**)&j->bmark);
    if (rc) return rc;
    DBHIP_CHECK(hipMemsetAsync(j->bmark, 0, bytes, s));
    j->bmark_rows = j->nrows;
  }
  if (n_pairs == 0) return DBHIP_OK;
  hipLaunchKernelGGL(join_mark_build_kernel, dim3(grid_for(n_pairs, 256)), dim3(256), 0, s, build_rows, n_pairs, j->nrows, j->bmark);
  DBHIP_LAUNCH_CHECK();
  return DBHIP_OK;
}

int32_t dbhip_join_build_matched(dbhip_join* j, uint8_t* out_bitmap, int64_t* out_build_rows_host, void* stream) {
  DBHIP_REQUIRE(j && j->finalized && out_build_rows_host, "dbhip_join_build_matched: needs a finalized table");
  *out_build_rows_host = j->nrows;
  if (!out_bitmap || j->nrows == 0) return DBHIP_OK;
  hipStream_t s = resolve_stream(stream);
  const size_t bytes = (size_t)ceil_div(j->nrows, 64) * 8;
  if (!j->bmark) DBHIP_CHECK(hipMemsetAsync(out_bitmap, 0, bytes, s));   // no probe block marked anything yet
  else DBHIP_CHECK(hipMemcpyAsync(out_bitmap, j->bmark, bytes, hipMemcpyDeviceToDevice, s));
  return DBHIP_OK;
}

int32_t dbhip_join_destroy(dbhip_join* j) {
  if (!j) return DBHIP_OK;
  (void)hipDeviceSynchronize();
  void* ptrs[] = {j->ent, j->head, j->occ, j->cnt, j->firstm, j->tsum, j->off, j->blk, j->total_dev, j->bmark};
  for (void* p : ptrs)
    if (p) (void)dbhip_free(p);
  delete j;
  return DBHIP_OK;
}

}  // extern "C"

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
#include "gb_device.h"
#include "runtime.h"

#include <string.h>

#include <new>
#include <vector>

using namespace dbhip;

#define GB_INVALID_SLOT 0xFFFFFFFFu

struct dbhip_groupby {
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
};

namespace {

// ---------------------------------------------------------------------------
// dbhip_group_hash
// ---------------------------------------------------------------------------
struct HashCols {
  GbCol c[GB_MAX_KEYS];
  int n;
};

__global__ __launch_bounds__(256) void group_hash_kernel(HashCols hc, int64_t n, uint64_t* out,
                                                         unsigned long long* bad) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    uint64_t h = 0;
    for (int k = 0; k < hc.n; ++k) {
      uint64_t w[2];
      bool valid;
      uint64_t hk;
      if (hc.c[k].type == DBHIP_T_STRING) {
        // general strings (any length): hash the bytes where they live
        int64_t j = hc.c[k].is_scalar ? 0 : i;
        valid = !hc.c[k].validity || bit_get(hc.c[k].validity, hc.c[k].voff + j);
        const uint32_t* v = (const uint32_t*)hc.c[k].data + 4 * j;
        uint32_t len = v[0];
        const uint8_t* p = len <= 12 ? (const uint8_t*)(v + 1)
                                     : (const uint8_t*)hc.c[k].buffers[v[2]] + v[3];
        hk = valid ? agg_hash_bytes(p, len) : DBHIP_NULL_HASH_VAL;
      } else {
        if (!gb_load_words(hc.c[k], i, w, &valid)) atomicAdd(bad, 1ULL);
        hk = gb_hash_words(hc.c[k].type, w, valid);
      }
      h = (k == 0) ? hk : merge_hash(h, hk);
    }
    out[i] = h;
  }
}

// ---------------------------------------------------------------------------
// serialize: columns -> rows_in
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gb_serialize_kernel(GbLayout L, GbCols C, int64_t n,
                                                           uint64_t* rows_in, uint64_t* ctrl) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    uint64_t* r = rows_in + i * L.W;
    uint64_t h = 0, vmask = 0;
    for (int k = 0; k < L.nkeys; ++k) {
      uint64_t w[2];
      bool valid;
      if (!gb_load_words(C.key[k], i, w, &valid)) atomicOr((unsigned long long*)&ctrl[3], 2ULL);
      uint64_t hk = gb_hash_words(L.key_type[k], w, valid);
      h = (k == 0) ? hk : merge_hash(h, hk);
      r[L.key_off[k]] = w[0];
      if (L.key_words[k] == 2) r[L.key_off[k] + 1] = w[1];
      if (valid) vmask |= 1ULL << k;
    }
    if (L.validity_word >= 0) r[L.validity_word] = vmask;
    r[L.hash_word] = h;
    for (int a = 0; a < L.naggs; ++a) {
      uint64_t* s = r + L.agg_off[a];
      uint64_t w[2] = {0, 0};
      bool valid = true;
      bool has_arg = C.arg[a].data != nullptr;
      if (has_arg) gb_load_words(C.arg[a], i, w, &valid);
      switch (L.agg_kind[a]) {
        case DBHIP_AGG_COUNT:
          s[0] = valid ? 1 : 0;
          break;
        case DBHIP_AGG_SUM:
          if (L.agg_type[a] == DBHIP_T_F32)
            w[0] = (uint64_t)__double_as_longlong((double)__uint_as_float((uint32_t)w[0]));
          s[0] = valid ? w[0] : 0;
          if (L.agg_words[a] == 3) {
            s[1] = valid ? w[1] : 0;
            s[2] = (valid && (w[1] >> 63)) ? ~0ULL : 0;  // sign extension to 192 bits
          }
          break;
        default:  // MIN / MAX
          s[0] = ord_encode(w[0], L.agg_type[a]);
          s[1] = valid ? 1 : 0;
          break;
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

__global__ __launch_bounds__(256) void gb_probe_kernel(GbLayout L, const uint64_t* rows_in, int64_t n,
                                                       uint64_t* slot_hash, uint64_t* rows, int64_t cap,
                                                       uint64_t hash_mask, uint32_t* gid, uint64_t* ctrl) {
  const uint64_t cmask = (uint64_t)cap - 1;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const uint64_t* r = rows_in + i * L.W;
    const uint64_t hw = probe_word(r[L.hash_word], hash_mask);
    uint64_t pos = hw & cmask;
    uint32_t found = GB_INVALID_SLOT;
    for (int64_t step = 0; step < cap; ++step) {
      unsigned long long cur = __hip_atomic_load((unsigned long long*)&slot_hash[pos], __ATOMIC_RELAXED,
                                                 __HIP_MEMORY_SCOPE_AGENT);
      if (cur == 0) {
        unsigned long long old = atomicCAS((unsigned long long*)&slot_hash[pos], 0ULL, (unsigned long long)hw);
        if (old == 0) {
          // this lane owns the new group: write keys, hash and identity states
          uint64_t* d = rows + pos * L.W;
          for (int k = 0; k < L.nkey_words; ++k) d[k] = r[k];
          d[L.hash_word] = r[L.hash_word];
          for (int a = 0; a < L.naggs; ++a) gb_state_identity(L, a, d + L.agg_off[a]);
          atomicAdd((unsigned long long*)&ctrl[0], 1ULL);
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
}

__device__ __forceinline__ bool keys_equal(const GbLayout& L, const uint64_t* a, const uint64_t* b) {
  bool eq = true;
  for (int k = 0; k < L.nkey_words; ++k) eq &= (a[k] == b[k]);
  return eq;
}

// ---------------------------------------------------------------------------
// accumulate — direct atomics (many groups)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gb_accum_kernel(GbLayout L, const uint64_t* rows_in, int64_t n,
                                                       uint64_t* rows, const uint32_t* gid,
                                                       uint32_t* retry, uint64_t* ctrl) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const uint64_t* r = rows_in + i * L.W;
    uint32_t pos = gid[i];
    uint64_t* d = rows + (uint64_t)pos * L.W;
    if (!keys_equal(L, r, d)) {
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
                                                               uint64_t* ctrl) {
  const int64_t n_pad = (n + 63) & ~63LL;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_pad;
       i += (int64_t)gridDim.x * blockDim.x) {
    bool active = i < n;
    const uint64_t* r = rows_in + (active ? i : 0) * L.W;
    uint32_t pos = active ? gid[i] : GB_INVALID_SLOT;
    if (active) {
      const uint64_t* d = rows + (uint64_t)pos * L.W;
      if (!keys_equal(L, r, d)) {
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
        uint64_t out[3] = {0, 0, 0};
        switch (L.agg_kind[a]) {
          case DBHIP_AGG_COUNT:
            out[0] = wave_sum_u64(mine ? v[0] : 0);
            break;
          case DBHIP_AGG_SUM:
            if (L.agg_words[a] == 3) {
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
            break;
          default:
            out[0] = wave_max_u64((mine && v[1]) ? v[0] : 0ULL);
            out[1] = wave_max_u64(mine ? v[1] : 0);
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
                                const uint32_t* retry, uint64_t* ctrl) {
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
        for (int k = 0; k < L.nkey_words; ++k) d[k] = r[k];
        d[L.hash_word] = r[L.hash_word];
        for (int a = 0; a < L.naggs; ++a) gb_state_identity(L, a, d + L.agg_off[a]);
        ctrl[0] += 1;
        cur = hw;
      }
      if (cur == hw && keys_equal(L, r, d)) {
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
    uint64_t pos = probe_word(r[L.hash_word], hash_mask) & cmask;
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
__global__ __launch_bounds__(256) void gb_flush_kernel(GbLayout L, const uint64_t* slot_hash,
                                                       const uint64_t* rows, int64_t cap,
                                                       uint64_t* out_rows, int64_t max_rows, uint64_t* ctrl) {
  const int64_t cap_pad = (cap + 63) & ~63LL;
  for (int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s < cap_pad;
       s += (int64_t)gridDim.x * blockDim.x) {
    bool occ = s < cap && slot_hash[s] != 0;
    uint64_t m = __ballot(occ);
    if (m == 0) continue;
    unsigned long long base = 0;
    if (lane_id() == 0) base = atomicAdd((unsigned long long*)&ctrl[4], (unsigned long long)__popcll(m));
    base = __shfl(base, 0, 64);
    if (occ) {
      uint64_t idx = base + __popcll(m & ((1ULL << lane_id()) - 1));
      if ((int64_t)idx < max_rows) {
        const uint64_t* r = rows + s * L.W;
        uint64_t* d = out_rows + idx * L.W;
        for (int k = 0; k < L.W; ++k) d[k] = r[k];
      }
    }
  }
}

struct ResultPtrs {
  void* keys[GB_MAX_KEYS];
  uint32_t* key_validity[GB_MAX_KEYS];
  void* aggs[GB_MAX_AGGS];
  uint64_t* hashes;
};

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
          case DBHIP_T_DEC128: case DBHIP_T_STRING: {
            uint64_t w1 = r[L.key_off[k] + 1];
            if (L.key_type[k] == DBHIP_T_STRING) {
              // words -> 16-byte view {len, bytes[12]}
              uint32_t* v = (uint32_t*)o + 4 * i;
              v[0] = (uint32_t)w0; v[1] = (uint32_t)(w0 >> 32); v[2] = (uint32_t)w1; v[3] = (uint32_t)(w1 >> 32);
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
      if (!o) continue;
      switch (L.agg_kind[a]) {
        case DBHIP_AGG_COUNT:
          ((uint64_t*)o)[i] = s[0];
          break;
        case DBHIP_AGG_SUM:
          if (L.agg_words[a] == 3) {
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
        default: {  // MIN / MAX
          uint64_t raw = ord_decode(s[0], L.agg_type[a]);
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

GbCol to_gbcol(const dbhip_col& c) {
  GbCol g;
  g.data = c.data; g.validity = c.validity; g.voff = c.validity_offset;
  g.buffers = c.buffers; g.type = c.type; g.is_scalar = c.is_scalar;
  return g;
}

bool key_type_ok(int t) { return t >= DBHIP_T_BOOL && t <= DBHIP_T_STRING; }

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
    L->key_words[k] = (key_types[k] == DBHIP_T_DEC128 || key_types[k] == DBHIP_T_STRING) ? 2 : 1;
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
        else if (!(d.arg_type >= DBHIP_T_I8 && d.arg_type <= DBHIP_T_F64) && d.arg_type != DBHIP_T_DEC64) {
          set_error("groupby: sum() does not support type %d", d.arg_type);
          return DBHIP_ERR_INVALID;
        }
        break;
      case DBHIP_AGG_MIN: case DBHIP_AGG_MAX:
        if (d.arg_type == DBHIP_T_DEC128 || d.arg_type == DBHIP_T_STRING || d.arg_nullable) {
          set_error("groupby: min/max on type %d (nullable=%d) stays on the CPU operator", d.arg_type, d.arg_nullable);
          return DBHIP_ERR_UNSUPPORTED;
        }
        words = 2;
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

int32_t ensure(void** p, size_t* cap, size_t bytes) {
  if (*cap >= bytes) return DBHIP_OK;
  if (*p) {
    DBHIP_CHECK(hipDeviceSynchronize());
    DBHIP_CHECK(hipFree(*p));
    *p = nullptr; *cap = 0;
  }
  size_t want = bytes + (bytes >> 3) + 256;
  DBHIP_CHECK(hipMalloc(p, want));
  *cap = want;
  return DBHIP_OK;
}

int32_t alloc_table(dbhip_groupby* g, int64_t cap, hipStream_t s) {
  DBHIP_CHECK(hipMalloc((void**)&g->slot_hash, (size_t)cap * 8));
  DBHIP_CHECK(hipMalloc((void**)&g->rows, (size_t)cap * g->L.W * 8));
  DBHIP_CHECK(hipMemsetAsync(g->slot_hash, 0, (size_t)cap * 8, s));
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
  DBHIP_CHECK(hipFree(old_hash));
  DBHIP_CHECK(hipFree(old_rows));
  return DBHIP_OK;
}

// probe + accumulate + retry over rows_in[n] (device rows in table layout)
int32_t merge_rows(dbhip_groupby* g, const uint64_t* rows_in, int64_t n, hipStream_t s) {
  if (n == 0) return DBHIP_OK;
  if (n > 0xFFFFFFF0LL) {
    set_error("groupby: more than 2^32 rows in one call");
    return DBHIP_ERR_INVALID;
  }
  int32_t rc;
  if ((rc = ensure((void**)&g->gid, &g->gid_cap, (size_t)n * 4))) return rc;
  if ((rc = ensure((void**)&g->retry, &g->retry_cap, (size_t)n * 4))) return rc;
  const int grid = grid_for(n, 256);
  uint64_t host_ctrl[5];
  const uint64_t* cur_rows = rows_in;
  int64_t cur_n = n;
  for (int attempt = 0; attempt < 40; ++attempt) {
    // ctrl[1] (overflow) and ctrl[2] (retry count) are per-attempt
    DBHIP_CHECK(hipMemsetAsync(&g->ctrl[1], 0, 16, s));
    hipLaunchKernelGGL(gb_probe_kernel, dim3(grid), dim3(256), 0, s, g->L, cur_rows, cur_n, g->slot_hash,
                       g->rows, g->cap, g->hash_mask, g->gid, g->ctrl);
    DBHIP_LAUNCH_CHECK();
    DBHIP_CHECK(hipMemcpyAsync(host_ctrl, g->ctrl, sizeof(host_ctrl), hipMemcpyDeviceToHost, s));
    DBHIP_CHECK(hipStreamSynchronize(s));
    bool too_full = host_ctrl[1] != 0 || (int64_t)host_ctrl[0] * 135 > g->cap * 100;
    if (too_full) {
      if ((rc = grow(g, s))) return rc;
      continue;  // redo the (idempotent) probe against the bigger table
    }
    g->count_host = (int64_t)host_ctrl[0];
    if (g->count_host <= 32) {
      hipLaunchKernelGGL(gb_accum_lowcard_kernel, dim3(grid), dim3(256), 0, s, g->L, cur_rows, cur_n,
                         g->rows, g->gid, g->retry, g->ctrl);
    } else {
      hipLaunchKernelGGL(gb_accum_kernel, dim3(grid), dim3(256), 0, s, g->L, cur_rows, cur_n, g->rows,
                         g->gid, g->retry, g->ctrl);
    }
    hipLaunchKernelGGL(gb_retry_kernel, dim3(1), dim3(64), 0, s, g->L, cur_rows, g->slot_hash, g->rows,
                       g->cap, g->hash_mask, g->gid, g->retry, g->ctrl);
    DBHIP_LAUNCH_CHECK();
    DBHIP_CHECK(hipMemcpyAsync(host_ctrl, g->ctrl, sizeof(host_ctrl), hipMemcpyDeviceToHost, s));
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

}  // namespace

// Used by k_q1.hip: merge `n` device rows (table layout) produced by a fused kernel.
int32_t dbhip_groupby_merge_rows_internal(dbhip_groupby* g, const uint64_t* rows, int64_t n, hipStream_t s) {
  return merge_rows(g, rows, n, s);
}
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
  unsigned long long* bad = (unsigned long long*)scratch(8, 2);
  if (!bad) return DBHIP_ERR_HIP;
  DBHIP_CHECK(hipMemsetAsync(bad, 0, 8, s));
  hipLaunchKernelGGL(group_hash_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, hc, n, out_hashes, bad);
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
  hipStream_t s = resolve_stream(nullptr);
  if ((rc = alloc_table(g, cap, s))) { delete g; return rc; }
  DBHIP_CHECK(hipMalloc((void**)&g->ctrl, 64));
  DBHIP_CHECK(hipMemsetAsync(g->ctrl, 0, 64, s));
  DBHIP_CHECK(hipStreamSynchronize(s));
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

int32_t dbhip_groupby_add_block(dbhip_groupby* g, const dbhip_col* keys, const dbhip_col* args,
                                int64_t n, void* stream) {
  DBHIP_REQUIRE(g && keys, "dbhip_groupby_add_block: NULL argument");
  if (n == 0) return DBHIP_OK;
  hipStream_t s = resolve_stream(stream);
  GbCols C;
  memset(&C, 0, sizeof(C));
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
  int32_t rc = ensure((void**)&g->rows_in, &g->rows_in_cap, (size_t)n * g->L.W * 8);
  if (rc) return rc;
  hipLaunchKernelGGL(gb_serialize_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, g->L, C, n, g->rows_in,
                     g->ctrl);
  DBHIP_LAUNCH_CHECK();
  return merge_rows(g, g->rows_in, n, s);
}

int32_t dbhip_groupby_merge_serialized(dbhip_groupby* g, const void* rows_dev, int64_t n_rows, void* stream) {
  DBHIP_REQUIRE(g && (rows_dev || n_rows == 0), "dbhip_groupby_merge_serialized: NULL argument");
  return merge_rows(g, (const uint64_t*)rows_dev, n_rows, resolve_stream(stream));
}

int32_t dbhip_groupby_num_groups(dbhip_groupby* g, int64_t* out_host, void* stream) {
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

int32_t dbhip_groupby_flush_result(dbhip_groupby* g, void* const* out_keys_host,
                                   uint8_t* const* out_key_validity_host, void* const* out_aggs_host,
                                   uint64_t* out_hashes, int64_t max_rows, int64_t* out_n_rows_host,
                                   void* stream) {
  DBHIP_REQUIRE(g && out_n_rows_host, "dbhip_groupby_flush_result: NULL argument");
  hipStream_t s = resolve_stream(stream);
  uint64_t* tmp = (uint64_t*)scratch((size_t)(max_rows > 0 ? max_rows : 1) * g->L.W * 8, 3);
  if (!tmp) return DBHIP_ERR_HIP;
  int32_t rc = dbhip_groupby_flush_serialized(g, tmp, max_rows, out_n_rows_host, stream);
  if (rc) return rc;
  int64_t n = *out_n_rows_host;
  if (n == 0) return DBHIP_OK;
  ResultPtrs P;
  memset(&P, 0, sizeof(P));
  for (int k = 0; k < g->L.nkeys; ++k) {
    P.keys[k] = out_keys_host ? out_keys_host[k] : nullptr;
    P.key_validity[k] = out_key_validity_host ? (uint32_t*)out_key_validity_host[k] : nullptr;
    if (P.key_validity[k]) DBHIP_CHECK(hipMemsetAsync(P.key_validity[k], 0, (size_t)ceil_div(max_rows, 32) * 4, s));
  }
  for (int a = 0; a < g->L.naggs; ++a) P.aggs[a] = out_aggs_host ? out_aggs_host[a] : nullptr;
  P.hashes = out_hashes;
  DBHIP_CHECK(hipMemsetAsync(&g->ctrl[3], 0, 8, s));
  hipLaunchKernelGGL(gb_result_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, g->L, tmp, n, P, g->ctrl);
  DBHIP_LAUNCH_CHECK();
  uint64_t err = 0;
  DBHIP_CHECK(hipMemcpyAsync(&err, &g->ctrl[3], 8, hipMemcpyDeviceToHost, s));
  DBHIP_CHECK(hipStreamSynchronize(s));
  if (err & 1) {
    set_error("Decimal overflow: sum state not in [DECIMAL_MIN, DECIMAL_MAX]");
    return DBHIP_ERR_OVERFLOW;
  }
  return DBHIP_OK;
}

int32_t dbhip_groupby_reset(dbhip_groupby* g, void* stream) {
  DBHIP_REQUIRE(g, "dbhip_groupby_reset: NULL argument");
  hipStream_t s = resolve_stream(stream);
  DBHIP_CHECK(hipMemsetAsync(g->slot_hash, 0, (size_t)g->cap * 8, s));
  DBHIP_CHECK(hipMemsetAsync(g->ctrl, 0, 64, s));
  g->count_host = 0;
  return DBHIP_OK;
}

int32_t dbhip_groupby_destroy(dbhip_groupby* g) {
  if (!g) return DBHIP_OK;
  (void)hipDeviceSynchronize();
  if (g->slot_hash) (void)hipFree(g->slot_hash);
  if (g->rows) (void)hipFree(g->rows);
  if (g->ctrl) (void)hipFree(g->ctrl);
  if (g->rows_in) (void)hipFree(g->rows_in);
  if (g->gid) (void)hipFree(g->gid);
  if (g->retry) (void)hipFree(g->retry);
  delete g;
  return DBHIP_OK;
}

}  // extern "C"

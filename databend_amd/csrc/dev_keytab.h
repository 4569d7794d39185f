// dev_keytab.h — per-workgroup table of a handful of distinct group keys (<= 8, four u64 words each), shared by the
// fused TPC-H Q1 kernel (k_q1.hip) and the generic few-groups aggregation kernel (k_groupby_few.hip).
#pragma once
#include "dev_common.h"

constexpr int MAX_SLOTS = 8;

// Per-block key table in LDS: append-only array of the distinct group keys seen by the
// block. Readers scan entries [0, count); a writer appends under `lock` and publishes
// by bumping `count` after a workgroup fence, so a reader never sees a half-written key.
struct KeyTable {
  uint32_t count;
  uint32_t lock;
  uint64_t key[MAX_SLOTS][4];
};

// slow path, ONE lane of a wave at a time: find or append under the lock. -1 when full.
template <int SLOTS>
__device__ __forceinline__ int tab_insert(KeyTable* T, uint64_t k0, uint64_t k1, uint64_t k2, uint64_t k3) {
  while (atomicCAS(&T->lock, 0u, 1u) != 0u) __builtin_amdgcn_s_sleep(1);
  volatile KeyTable* V = T;
  uint32_t nk = V->count;
  int slot = -1;
  for (uint32_t s = 0; s < nk; ++s)
    if (V->key[s][0] == k0 && V->key[s][1] == k1 && V->key[s][2] == k2 && V->key[s][3] == k3) slot = (int)s;
  if (slot < 0 && nk < (uint32_t)SLOTS) {
    V->key[nk][0] = k0; V->key[nk][1] = k1; V->key[nk][2] = k2; V->key[nk][3] = k3;
    __threadfence_block();
    V->count = nk + 1;
    slot = (int)nk;
  }
  __threadfence_block();
  atomicExch(&T->lock, 0u);
  return slot;
}

__device__ __forceinline__ uint64_t uniform_u64(uint64_t v) {
  uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
  uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  return ((uint64_t)hi << 32) | lo;
}

// Wave-private, scalar-register copy of the published part of the block's key table.
template <int SLOTS>
struct TabCache {
  uint32_t nk;
  uint64_t k[SLOTS][4];
  __device__ __forceinline__ void refresh(KeyTable* T) {
    volatile KeyTable* V = T;
    nk = __builtin_amdgcn_readfirstlane(V->count);
#pragma unroll
    for (int s = 0; s < SLOTS; ++s)
#pragma unroll
      for (int j = 0; j < 4; ++j) k[s][j] = uniform_u64(V->key[s][j]);
  }
  // branch-free compare against every published entry; -1 if absent
  __device__ __forceinline__ int lookup(uint64_t k0, uint64_t k1, uint64_t k2, uint64_t k3) const {
    int slot = -1;
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
      bool eq = ((uint32_t)s < nk) & (k[s][0] == k0) & (k[s][1] == k1) & (k[s][2] == k2) & (k[s][3] == k3);
      slot = eq ? s : slot;
    }
    return slot;
  }
};


// slot of one key row given its four canonical words (wave-convergent call): published entries are matched against
// the wave's scalar-register copy; a key nobody published yet is inserted by one lane at a time under the table's
// lock. 0xF = row not wanted, 0xE = dropped (table full: flags |= 1).
template <int SLOTS>
__device__ __forceinline__ int resolve_slot_words(KeyTable* T, TabCache<SLOTS>& C, bool want, uint64_t k0, uint64_t k1,
                                                  uint64_t k2, uint64_t k3, uint32_t& flags) {
  int slot = C.lookup(k0, k1, k2, k3);
  slot = want ? slot : 0xF;
  uint64_t miss = __ballot(slot < 0);
  while (miss) {  // rare: a key this wave has not seen published yet
    const int leader = __ffsll((long long)miss) - 1;
    if (lane_id() == leader) {
      int ls = tab_insert<SLOTS>(T, k0, k1, k2, k3);
      if (ls < 0) flags |= 1u;
    }
    C.refresh(T);
    int again = C.lookup(k0, k1, k2, k3);
    // after the leader's insert its key is published (or the table is full)
    const bool full = C.nk >= (uint32_t)SLOTS;
    if (slot < 0) slot = again >= 0 ? again : (full ? 0xE : -1);
    miss = __ballot(slot < 0);
  }
  return slot;
}

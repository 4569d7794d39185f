// gbk_rows.h — a fragment of k_groupby.hip (ONE translation unit: the kernels share the anonymous namespace's helpers and the table struct;
// split by kernel family in round 6, VERDICT r05 hygiene #18): the row path: group hash, serialize, probe / accumulate / retry, rehash, flush, the a12 partitioning of group rows, serialized-state blocks, table layout.
// Included by k_groupby.hip only, in this order: gbk_rows.h, gbk_merge_lds.h, gbk_partitioned.h, gbk_api.h.

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

// (bodies as device functions: the three kernels below, and ONE single-workgroup kernel that runs them back to back for small merges)
__device__ __forceinline__ void gb_probe_body(const GbLayout& L, const uint64_t* rows_in, int64_t n,
                                              uint64_t* slot_hash, uint64_t* rows, int64_t cap,
                                              uint64_t hash_mask, uint32_t* gid, uint64_t* ctrl, uint8_t* arena) {
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
__global__ __launch_bounds__(256) void gb_probe_kernel(GbLayout L, const uint64_t* rows_in, int64_t n,
                                                       uint64_t* slot_hash, uint64_t* rows, int64_t cap,
                                                       uint64_t hash_mask, uint32_t* gid, uint64_t* ctrl, DevCount dc, uint8_t* arena) {
  gb_probe_body(L, rows_in, dev_rows(dc, n), slot_hash, rows, cap, hash_mask, gid, ctrl, arena);
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

__device__ __forceinline__ void gb_accum_lowcard_body(const GbLayout& L, const uint64_t* rows_in,
                                                      int64_t n, uint64_t* rows,
                                                      const uint32_t* gid, uint32_t* retry,
                                                      uint64_t* ctrl, const uint8_t* arena) {
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
__global__ __launch_bounds__(256) void gb_accum_lowcard_kernel(GbLayout L, const uint64_t* rows_in,
                                                               int64_t n, uint64_t* rows,
                                                               const uint32_t* gid, uint32_t* retry,
                                                               uint64_t* ctrl, DevCount dc, const uint8_t* arena) {
  gb_accum_lowcard_body(L, rows_in, dev_rows(dc, n), rows, gid, retry, ctrl, arena);
}

// ---------------------------------------------------------------------------
// retry — serial continuation of the probe for true hash collisions
// ---------------------------------------------------------------------------
__device__ __forceinline__ void gb_retry_body(const GbLayout& L, const uint64_t* rows_in, uint64_t* slot_hash, uint64_t* rows,
                                              int64_t cap, uint64_t hash_mask, const uint32_t* gid,
                                              const uint32_t* retry, uint64_t* ctrl, uint8_t* arena) {
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
__global__ void gb_retry_kernel(GbLayout L, const uint64_t* rows_in, uint64_t* slot_hash, uint64_t* rows,
                                int64_t cap, uint64_t hash_mask, const uint32_t* gid,
                                const uint32_t* retry, uint64_t* ctrl, uint8_t* arena) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  gb_retry_body(L, rows_in, slot_hash, rows, cap, hash_mask, gid, retry, ctrl, arena);
}

// Small merges (the partial rows behind a fused kernel, the blocks of a fixed-slot exchange: tens to a few thousand rows): the three
// steps above in ONE launch of ONE workgroup, the per-merge control words cleared by the kernel itself. Such a merge is bound by what the
// host pays per queued operation (~5 us each on this stack), not by the device: one launch where rounds 1-5 queued a memset and three
// kernels. Between the steps: a device-scope fence and the workgroup barrier (the claimed rows' keys and identity states are plain
// stores that the accumulate step compares against and adds to with atomics).
__global__ __launch_bounds__(256) void gb_merge_small_kernel(GbLayout L, const uint64_t* rows_in, int64_t n, uint64_t* slot_hash, uint64_t* rows,
                                                             int64_t cap, uint64_t hash_mask, uint32_t* gid, uint32_t* retry, uint64_t* ctrl,
                                                             DevCount dc, uint8_t* arena) {
  n = dev_rows(dc, n);
  if (threadIdx.x < 2) ctrl[1 + threadIdx.x] = 0;   // overflow flags, retry count
  __threadfence();
  __syncthreads();
  gb_probe_body(L, rows_in, n, slot_hash, rows, cap, hash_mask, gid, ctrl, arena);
  __threadfence();
  __syncthreads();
  gb_accum_lowcard_body(L, rows_in, n, rows, gid, retry, ctrl, arena);
  __threadfence();
  __syncthreads();
  if (threadIdx.x == 0) gb_retry_body(L, rows_in, slot_hash, rows, cap, hash_mask, gid, retry, ctrl, arena);
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

// one workgroup per source block: append its rows behind those of the earlier blocks (`skip` = the caller's own block).
// `xst` (optional, the queued exchange): nothing was read back, so every workgroup judges ALL the headers itself (the same words on every rank)
// — a sender that overflowed (`check_w1`: or says that one of its other blocks did) => nothing is copied; workgroup 0 leaves
// xst[0] = the rows of all blocks, xst[1] = 1 if the exchange is off (the DevCount pair of the merge queued behind), xst[2] = the offending block + 1
__global__ __launch_bounds__(256) void gb_compact_blocks_kernel(const uint64_t* __restrict__ blocks, int64_t stride_words, int W,
                                                                int skip, uint64_t* __restrict__ out, uint64_t* __restrict__ xst = nullptr,
                                                                int n_blocks = 0, int64_t max_rows = 0, int check_w1 = 0) {
  const int b = blockIdx.x;
  if (xst) {
    uint64_t total = 0;
    int bad = 0;
    for (int p = 0; p < n_blocks; ++p) {
      const uint64_t c = blocks[(int64_t)p * stride_words];
      if (c == ~0ULL || (int64_t)c > max_rows || (check_w1 && blocks[(int64_t)p * stride_words + 1] != 0)) { if (!bad) bad = p + 1; }
      else if (p != skip) total += c;
    }
    if (b == 0 && threadIdx.x == 0) { xst[0] = bad ? 0 : total; xst[1] = bad ? 1 : 0; xst[2] = (uint64_t)bad; }
    if (bad) return;
  }
  if (b == skip) return;
  const uint64_t cnt = blocks[(int64_t)b * stride_words];
  uint64_t off = 0;
  for (int p = 0; p < b; ++p)
    if (p != skip) off += blocks[(int64_t)p * stride_words];
  const uint64_t* src = blocks + (int64_t)b * stride_words + W;
  uint64_t* dst = out + off * W;
  for (uint64_t i = threadIdx.x; i < cnt * (uint64_t)W; i += blockDim.x) dst[i] = src[i];
}

// the queued exchange's dbhip_groupby_reset: the table is emptied on the device unless the headers say that the exchange is off (then it
// must stay as it was: the caller falls back to the variable-length path with its groups intact)
__global__ __launch_bounds__(256) void gb_reset_unless_off_kernel(const uint64_t* __restrict__ blocks, int64_t stride_words, int n_blocks,
                                                                  int64_t max_rows, uint64_t* __restrict__ slot_hash, int64_t cap, uint64_t* __restrict__ ctrl) {
  for (int p = 0; p < n_blocks; ++p) {
    const uint64_t c = blocks[(int64_t)p * stride_words];
    if (c == ~0ULL || (int64_t)c > max_rows || blocks[(int64_t)p * stride_words + 1] != 0) return;
  }
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < cap; i += (int64_t)gridDim.x * blockDim.x) slot_hash[i] = 0;
  if (blockIdx.x == 0 && threadIdx.x < 16) ctrl[threadIdx.x] = 0;
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
                                                                   uint32_t n_buckets, unsigned long long* cursor) {
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
    cursor[b] = 0;   // (this thread is the only reader of cursor[b] after the barrier above)
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

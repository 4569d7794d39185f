// gb_compact.h — the COMPACT-ROW kernels of the hash aggregation (included by k_groupby.hip inside its anonymous namespace).
//
// Round 4. The generic partition pipeline (gb_part_hist / scatter / agg above) moves serialized rows [keys | hash | states]
// (32 B for one i64 key + sum + count) and decodes the table layout per row; r02w's counters had it instruction bound at ~300
// instructions per 64 rows and at ~100 B of traffic per 16 B row. For the layouts that dominate (1-2 fixed-width keys without NULLs,
// <= 4 count / sum / min / max aggregates over non-nullable fixed-width arguments) these kernels work on a COMPACT row instead:
//
//     [key word(s)] [one value word per aggregate that has an argument]            (16 B for i64 key + sum + count(*))
//
// the hash is recomputed where it is needed (two multiplies), count(*) has no word at all, and the layout is a template
// (KW key words, NV value words) plus a small uniform descriptor whose switches sit OUTSIDE the per-row loops:
//   gbc_hist_kernel     partition histogram from the key columns alone
//   gbc_scatter_kernel  compact rows into their partition (LDS cursors, LDS-staged batch, 16-byte copy-out per row)
//   gbc_agg_kernel      LDS hash table per workgroup — over a partition's compact rows, or (FROM_COLS) straight over the input
//                       columns when all groups fit one workgroup's table (no partitioning pass at all: up to ~2.9 K groups in a
//                       112 KB table, one 1024-thread workgroup per CU). States are pre-set, a one-word key is claimed with ONE
//                       CAS on the key word itself (two words: tag 0 -> 1 -> tag, keys published before the tag), the first probe
//                       of a lane's 4 rows is issued together, the next tile's loads are in flight while a tile is worked on; no
//                       barrier per tile.
// Outputs are what the generic kernels produce — partial rows in table layout (per workgroup, with per-workgroup counts for the
// partition-exclusive merge) and spilled rows in table layout — so everything downstream (gb_part_merge_kernel, merge_rows, the
// growth logic) is shared. Reference semantics: aggregate_hashtable.rs:168-333, partitioned_payload.rs:160-240.
#pragma once

// Round 5: the layouts plans actually produce — up to three keys in up to four key words (16-byte keys: Decimal128 / inline String
// views; nullable keys: the table row's validity word travels as one more key word), up to eight aggregates in up to eight value
// words (Decimal128 sums take two), arguments with validity (one more value word holds a "not NULL" bit per nullable argument).
// Value words hold the argument's canonical words AS LOADED — the order-preserving image of min / max and the f32 -> f64 widening of
// a float sum are applied where the row is merged — so aggregates over the same column share one value word (sum(x), min(x), max(x),
// avg(x) read x once and a compact row carries it once).
constexpr int GBC_MAX_AGGS = 8;
constexpr int GBC_MAX_SW = 16;
constexpr int GBC_MAX_KEYS = 3;
constexpr int GBC_MAX_KW = 4;
constexpr int GBC_MAX_NV = 8;
enum { GBC_COUNT = 0, GBC_SUM_INT = 1, GBC_SUM_F32 = 2, GBC_SUM_F64 = 3, GBC_MIN = 4, GBC_MAX = 5, GBC_SUM_I128 = 6 };
constexpr uint32_t GBC_FULL = 0xFFFFFFFFu;
constexpr uint32_t GBC_NONE = 0xFFFFFFFEu;

struct GbcDesc {
  int32_t kw, nv, naggs, sw;          // key words, value words of a compact row, aggregates, state words of a table row
  int32_t nkeys;
  int32_t vword;                      // key word that holds the keys' validity bits (the table row's validity word), -1 = none
  int32_t simple_keys;                // every key is one word and none is nullable: key word j IS key column j's value
  int32_t key_type[GBC_MAX_KEYS], key_off[GBC_MAX_KEYS], key_two[GBC_MAX_KEYS];
  int32_t op[GBC_MAX_AGGS];           // GBC_*
  int32_t val[GBC_MAX_AGGS];          // (first) value word of the compact row, -1 = none (count)
  int32_t off[GBC_MAX_AGGS];          // first state word, relative to the first state word of the table row
  int32_t type[GBC_MAX_AGGS];         // argument type (order-preserving image of min / max)
  int32_t abit[GBC_MAX_AGGS];         // bit of value word `vmw` that says "this row's argument is not NULL", -1 = it never is
  int32_t fword[GBC_MAX_AGGS];        // state word (relative to off[a]) that says "merged a non-NULL row" (sum: the adaptor's flag, min / max: has-value), -1 = none
  int32_t fset[GBC_MAX_AGGS];         // 1: rows have to SET that word (the argument can be NULL); 0: a fresh group starts with it set
  int32_t own[GBC_MAX_AGGS];          // this aggregate's column is loaded into val[a] here (0: an earlier aggregate over the same column did)
  int32_t vown[GBC_MAX_AGGS];         // this aggregate's validity is loaded into bit abit[a] here
  int32_t vmw;                        // value word of the not-NULL bits, -1 = no argument with validity
  int32_t W, hash_word, state_off;    // table row
  uint64_t ident[GBC_MAX_SW];         // the states of a fresh group
  // every key and argument column holds 8-byte values (i64 / u64 / f64 / timestamp / Decimal64), nothing is nullable: the loads of a
  // tile are then issued from ONE basic block with typed pointers — behind the generic per-column type switch the compiler gives the
  // loads of different columns the same destination registers, so the second column's loads wait for the first column's data (r04e
  // ISA: s_waitcnt vmcnt(0) between the key loads and the value loads of a tile)
  int32_t all8;
  const uint64_t* kcol[GBC_MAX_KEYS];
  const uint64_t* vcol[GBC_MAX_NV];   // by value word
  uint64_t* ctrl;                     // the table's control words ([3] |= 2: a String key longer than an inline view — row path)
};

inline bool gbc_same_col(const GbCol& a, const GbCol& b) {
  return a.data == b.data && a.validity == b.validity && a.voff == b.voff && a.type == b.type && a.is_scalar == b.is_scalar;
}

// Can this table + these columns go through the compact kernels?
inline bool gbc_describe(const GbLayout& L, const GbCols& C, GbcDesc* D) {
  memset(D, 0, sizeof(*D));
  if (L.nkeys < 1 || L.nkeys > GBC_MAX_KEYS || L.nkey_words > GBC_MAX_KW || L.naggs < 1 || L.naggs > GBC_MAX_AGGS) return false;
  if (L.hash_word != L.nkey_words || L.agg_off[0] != L.hash_word + 1) return false;
  D->nkeys = L.nkeys; D->kw = L.nkey_words; D->vword = L.validity_word;
  D->simple_keys = L.validity_word < 0 ? 1 : 0;
  for (int k = 0; k < L.nkeys; ++k) {
    const int t = L.key_type[k];
    if (L.key_words[k] > 2 || t == DBHIP_T_DEC256) return false;
    if (C.key[k].is_scalar || C.key[k].type != t || (C.key[k].validity && !L.key_nullable[k])) return false;
    D->key_type[k] = t; D->key_off[k] = L.key_off[k]; D->key_two[k] = L.key_words[k] == 2 ? 1 : 0;
    if (L.key_words[k] != 1 || L.key_off[k] != k) D->simple_keys = 0;
  }
  D->naggs = L.naggs;
  D->W = L.W; D->hash_word = L.hash_word; D->state_off = L.agg_off[0];
  D->sw = L.W - D->state_off;
  if (D->sw > GBC_MAX_SW) return false;
  D->vmw = -1;
  int at = D->state_off, nbits = 0;
  bool wide = false;
  for (int a = 0; a < L.naggs; ++a) {
    if (L.agg_off[a] != at) return false;   // states back to back
    at += L.agg_words[a];
    D->off[a] = L.agg_off[a] - D->state_off;
    D->type[a] = L.agg_type[a];
    D->val[a] = -1; D->abit[a] = -1; D->fword[a] = -1;
    const int t = L.agg_type[a];
    const GbCol& ac = C.arg[a];
    const bool has_arg = ac.data != nullptr;
    if (has_arg && L.agg_kind[a] != DBHIP_AGG_COUNT && ac.type != t) return false;
    const bool one_word = t == DBHIP_T_I8 || t == DBHIP_T_I16 || t == DBHIP_T_I32 || t == DBHIP_T_I64 || t == DBHIP_T_U8 || t == DBHIP_T_U16 ||
                          t == DBHIP_T_U32 || t == DBHIP_T_U64 || t == DBHIP_T_F32 || t == DBHIP_T_F64 || t == DBHIP_T_DATE ||
                          t == DBHIP_T_TIMESTAMP || t == DBHIP_T_DEC64;
    int need = 0;   // value words
    switch (L.agg_kind[a]) {
      case DBHIP_AGG_COUNT:
        if (L.agg_words[a] != 1) return false;
        D->op[a] = GBC_COUNT;
        D->ident[D->off[a]] = 0;
        break;
      case DBHIP_AGG_SUM: {
        const int body = L.agg_words[a] - (L.agg_flag[a] ? 1 : 0);
        if (!has_arg || (L.agg_flag[a] && L.agg_flag[a] != body)) return false;
        if (body == 1 && one_word) { D->op[a] = t == DBHIP_T_F32 ? GBC_SUM_F32 : (t == DBHIP_T_F64 ? GBC_SUM_F64 : GBC_SUM_INT); need = 1; }
        else if (body == 3 && t == DBHIP_T_DEC128) { D->op[a] = GBC_SUM_I128; need = 2; wide = true; }
        else return false;
        for (int w = 0; w < body; ++w) D->ident[D->off[a] + w] = 0;
        if (L.agg_flag[a]) D->fword[a] = L.agg_flag[a];
      } break;
      case DBHIP_AGG_MIN: case DBHIP_AGG_MAX:
        if (L.agg_words[a] != 2 || !has_arg || !one_word) return false;
        D->op[a] = L.agg_kind[a] == DBHIP_AGG_MIN ? GBC_MIN : GBC_MAX;
        need = 1;
        D->ident[D->off[a]] = L.agg_kind[a] == DBHIP_AGG_MIN ? ~0ULL : 0ULL;
        D->fword[a] = 1;
        break;
      default: return false;
    }
    if (has_arg && need) {   // the column's words: shared with an earlier aggregate over the same column
      for (int b = 0; b < a && D->val[a] < 0; ++b)
        if (D->val[b] >= 0 && C.arg[b].data && gbc_same_col(ac, C.arg[b]) && (D->op[b] == GBC_SUM_I128) == (need == 2)) D->val[a] = D->val[b];
      if (D->val[a] < 0) { D->val[a] = D->nv; D->nv += need; D->own[a] = 1; }
    }
    if (has_arg && ac.validity) {
      for (int b = 0; b < a && D->abit[a] < 0; ++b)
        if (D->abit[b] >= 0 && gbc_same_col(ac, C.arg[b])) D->abit[a] = D->abit[b];
      if (D->abit[a] < 0) { D->abit[a] = nbits++; D->vown[a] = 1; }
    }
    if (D->fword[a] >= 0) {   // a fresh group has met a row: with an argument that is never NULL the word starts set
      D->fset[a] = D->abit[a] >= 0 ? 1 : 0;
      D->ident[D->off[a] + D->fword[a]] = D->fset[a] ? 0 : 1;
    }
  }
  if (at != L.W) return false;
  if (nbits) D->vmw = D->nv++;
  if (D->nv > GBC_MAX_NV) return false;
  auto is8 = [](int t) { return t == DBHIP_T_I64 || t == DBHIP_T_U64 || t == DBHIP_T_F64 || t == DBHIP_T_TIMESTAMP || t == DBHIP_T_DEC64; };
  D->all8 = (D->simple_keys && !nbits && !wide) ? 1 : 0;
  for (int k = 0; k < L.nkeys; ++k) { D->all8 &= is8(L.key_type[k]) ? 1 : 0; D->kcol[k] = (const uint64_t*)C.key[k].data; }
  for (int a = 0; a < L.naggs; ++a)
    if (D->own[a]) { D->all8 &= (is8(L.agg_type[a]) && !C.arg[a].is_scalar) ? 1 : 0; D->vcol[D->val[a]] = (const uint64_t*)C.arg[a].data; }
  return true;
}

// word `idx` (uniform, run time) of a lane's register array: a scalar branch per candidate, not a chain of per-lane selects
template <int N, int R>
__device__ __forceinline__ void gbc_pick(const uint64_t (&v)[R][N], int idx, uint64_t (&out)[R]) {
#pragma unroll
  for (int x = 0; x < R; ++x) out[x] = 0;
  switch (idx) {
#define GBC_PICK_CASE(J) case J: if (J < N) {                        \
      _Pragma("unroll") for (int x = 0; x < R; ++x) out[x] = v[x][J < N ? J : 0]; } break;
    GBC_PICK_CASE(0) GBC_PICK_CASE(1) GBC_PICK_CASE(2) GBC_PICK_CASE(3) GBC_PICK_CASE(4) GBC_PICK_CASE(5) GBC_PICK_CASE(6) GBC_PICK_CASE(7)
#undef GBC_PICK_CASE
    default: break;
  }
}
template <int N>
__device__ __forceinline__ uint64_t gbc_pick1(const uint64_t (&v)[N], int idx) {
  uint64_t r = 0;
#pragma unroll
  for (int j = 0; j < N; ++j) r = (j == idx) ? v[j] : r;
  return r;
}

// ---- loads --------------------------------------------------------------------------------------------------------------------
template <int KW, int R>
__device__ __forceinline__ void gbc_load_keys(const GbcDesc& D, const GbCols& C, const int64_t (&row)[R], uint64_t (&k)[R][KW]) {
  if (D.all8) {
#pragma unroll
    for (int j = 0; j < KW; ++j)
#pragma unroll
      for (int x = 0; x < R; ++x) k[x][j] = D.kcol[j < GBC_MAX_KEYS ? j : 0][row[x]];
    return;
  }
  uint64_t vm[R];
#pragma unroll
  for (int x = 0; x < R; ++x) {
    vm[x] = 0;
#pragma unroll
    for (int j = 0; j < KW; ++j) k[x][j] = 0;
  }
#pragma unroll
  for (int kk = 0; kk < GBC_MAX_KEYS; ++kk) {
    if (kk < D.nkeys && kk < KW) {   // (uniform)
      uint64_t w0[R], w1[R];
      bool valid[R];
      if (!gb_load_words_n<R>(C.key[kk], row, w0, w1, valid)) atomicOr((unsigned long long*)&D.ctrl[3], 2ULL);
      const int off = D.key_off[kk], off1 = D.key_two[kk] ? off + 1 : -1;
#pragma unroll
      for (int x = 0; x < R; ++x) {
#pragma unroll
        for (int j = 0; j < KW; ++j) k[x][j] = (j == off) ? w0[x] : ((j == off1) ? w1[x] : k[x][j]);
        vm[x] |= valid[x] ? (1ULL << kk) : 0ULL;
      }
    }
  }
  if (D.vword >= 0) {
#pragma unroll
    for (int x = 0; x < R; ++x)
#pragma unroll
      for (int j = 0; j < KW; ++j) k[x][j] = (j == D.vword) ? vm[x] : k[x][j];
  }
}

// the value words of R rows: the argument columns' canonical words as loaded, and the not-NULL bits
template <int NV, int R>
__device__ __forceinline__ void gbc_load_values(const GbcDesc& D, const GbCols& C, const int64_t (&row)[R], uint64_t (&v)[R][NV]) {
#pragma unroll
  for (int x = 0; x < R; ++x)
#pragma unroll
    for (int j = 0; j < NV; ++j) v[x][j] = 0;
  if (D.all8) {
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      if (j < D.nv) {   // (uniform)
#pragma unroll
        for (int x = 0; x < R; ++x) v[x][j] = D.vcol[j][row[x]];
      }
    }
    return;
  }
  for (int a = 0; a < D.naggs; ++a) {
    if (!D.own[a] && !D.vown[a]) continue;
    uint64_t w0[R], w1[R];
    bool valid[R];
    const GbCol& ac = C.arg[a];
    if (D.own[a]) gb_load_words_n<R>(ac, row, w0, w1, valid);
    else {
#pragma unroll
      for (int x = 0; x < R; ++x) { w0[x] = 0; w1[x] = 0; valid[x] = bit_get(ac.validity, ac.voff + (ac.is_scalar ? 0 : row[x])); }
    }
    if (D.own[a]) {
      const int vi = D.val[a], vi1 = D.op[a] == GBC_SUM_I128 ? vi + 1 : -1;
#pragma unroll
      for (int x = 0; x < R; ++x)
#pragma unroll
        for (int j = 0; j < NV; ++j) v[x][j] = (j == vi) ? w0[x] : ((j == vi1) ? w1[x] : v[x][j]);
    }
    if (D.vown[a]) {
      const int bit = D.abit[a];
#pragma unroll
      for (int x = 0; x < R; ++x)
#pragma unroll
        for (int j = 0; j < NV; ++j) v[x][j] = (j == D.vmw) ? (v[x][j] | (valid[x] ? (1ULL << bit) : 0ULL)) : v[x][j];
    }
  }
}

template <int KW>
__device__ __forceinline__ uint64_t gbc_hash(const GbcDesc& D, const uint64_t (&k)[KW]) {
  uint64_t h = 0;
  if (KW == 1 || D.simple_keys) {
#pragma unroll
    for (int j = 0; j < KW; ++j) {
      const uint64_t w[2] = {k[j], 0};
      const uint64_t hk = gb_hash_words(D.key_type[j < GBC_MAX_KEYS ? j : 0], w, true);
      h = (j == 0) ? hk : merge_hash(h, hk);
    }
    return h;
  }
  const uint64_t vm = D.vword >= 0 ? gbc_pick1<KW>(k, D.vword) : ~0ULL;
#pragma unroll
  for (int kk = 0; kk < GBC_MAX_KEYS; ++kk) {
    if (kk < D.nkeys) {   // (uniform)
      const uint64_t w[2] = {gbc_pick1<KW>(k, D.key_off[kk]), D.key_two[kk] ? gbc_pick1<KW>(k, D.key_off[kk] + 1) : 0};
      const uint64_t hk = gb_hash_words(D.key_type[kk], w, ((vm >> kk) & 1ULL) != 0);
      h = (kk == 0) ? hk : merge_hash(h, hk);
    }
  }
  return h;
}

// the table-layout image of one compact row (spilled rows go through the row path like any serialized row)
template <int KW, int NV>
__device__ __forceinline__ void gbc_write_full_row(const GbcDesc& D, const uint64_t (&k)[KW], const uint64_t (&v)[NV], uint64_t h, uint64_t* out) {
#pragma unroll
  for (int j = 0; j < KW; ++j) out[j] = k[j];
  out[D.hash_word] = h;
  const uint64_t vm = D.vmw >= 0 ? gbc_pick1<NV>(v, D.vmw) : 0;
  for (int a = 0; a < D.naggs; ++a) {
    uint64_t* s = out + D.state_off + D.off[a];
    const uint64_t val = gbc_pick1<NV>(v, D.val[a]);
    const bool ok = D.abit[a] < 0 || ((vm >> D.abit[a]) & 1ULL);
    switch (D.op[a]) {
      case GBC_COUNT: s[0] = ok ? 1 : 0; break;
      case GBC_MIN: case GBC_MAX: s[0] = ord_encode(val, D.type[a]); break;
      case GBC_SUM_F32: s[0] = ok ? (uint64_t)__double_as_longlong((double)__uint_as_float((uint32_t)val)) : 0; break;
      case GBC_SUM_I128: {
        const uint64_t hi = gbc_pick1<NV>(v, D.val[a] + 1);
        s[0] = ok ? val : 0; s[1] = ok ? hi : 0; s[2] = (ok && (hi >> 63)) ? ~0ULL : 0;
      } break;
      default: s[0] = ok ? val : 0; break;
    }
    if (D.fword[a] >= 0) s[D.fword[a]] = ok ? 1 : 0;
  }
}

// rows per lane and tile of the aggregation kernel, and whether its deferred-row queue holds the rows themselves or their positions
// (2 x rows x row words x 2 registers are live across a tile, and a 1024-thread workgroup has 128 registers per lane: 4 rows up to 4
// words, 2 up to 8, 1 beyond — r05: with 4 rows the 5- and 6-word shapes spilled 15 ... 160 registers to scratch; a queue of 128
// twelve-word rows per wave would be 192 KB per workgroup)
constexpr int gbc_rows_per_lane(int rw) { return rw <= 4 ? 4 : (rw <= 8 ? 2 : 1); }
constexpr bool gbc_queue_of_positions(int rw) { return rw > 6; }

// ---- partition histogram ------------------------------------------------------------------------------------------------------------
constexpr int GBC_T = 1024;

template <int KW>
__global__ __launch_bounds__(GBC_T) void gbc_hist_kernel(GbcDesc D, GbCols C, int64_t row0, int64_t n, int pbits, int64_t rows_per_wg, uint32_t* mat) {
  extern __shared__ uint32_t gbc_hist_lds[];
  constexpr int R = 4;
  const int P = 1 << pbits;
  const int T = blockDim.x;
  for (int s = threadIdx.x; s < P; s += T) gbc_hist_lds[s] = 0;
  __syncthreads();
  const int64_t lo = (int64_t)blockIdx.x * rows_per_wg;
  const int64_t hi = lo + rows_per_wg < n ? lo + rows_per_wg : n;
  for (int64_t t0 = lo; t0 < hi; t0 += (int64_t)T * R) {
    int64_t row[R];
    bool in[R];
    uint64_t k[R][KW];
#pragma unroll
    for (int x = 0; x < R; ++x) {
      const int64_t li = t0 + (int64_t)x * T + threadIdx.x;
      in[x] = li < hi;
      row[x] = row0 + (in[x] ? li : lo);
      in[x] = in[x] && gb_row_passes(C, row[x]);   // (the pushed-down predicate Bitmap: rows that fail it do not take part)
    }
    gbc_load_keys<KW, R>(D, C, row, k);
#pragma unroll
    for (int x = 0; x < R; ++x)
      if (in[x]) atomicAdd(&gbc_hist_lds[part_of(gbc_hash<KW>(D, k[x]), pbits)], 1u);
  }
  __syncthreads();
  uint32_t* out = mat + (size_t)blockIdx.x * P;
  for (int s = threadIdx.x; s < P; s += T) out[s] = gbc_hist_lds[s];
}

// ---- scatter ------------------------------------------------------------------------------------------------------------------------
// Workgroup b walks the row range it counted; lcur[p] (LDS) = next output row of its run in partition p (one LDS atomic per row, no
// global atomic). A batch of SR rows per thread is staged in LDS as compact rows next to their output positions and copied out one
// ROW per thread (a 16-byte store for the i64 key + one value shape): rows that follow each other in a partition leave as one piece.
template <int KW, int NV>
__global__ __launch_bounds__(GBC_T) void gbc_scatter_kernel(GbcDesc D, GbCols C, int64_t row0, int64_t n, int pbits, int64_t rows_per_wg,
                                                            const uint32_t* mat, uint64_t* rows_out) {
  extern __shared__ uint32_t gbc_sc_lds[];
  constexpr int RW = KW + NV;
  constexpr int SR = RW <= 2 ? 4 : (RW <= 4 ? 2 : 1);
  const int P = 1 << pbits;
  const int T = blockDim.x;
  const int tid = threadIdx.x;
  const int BR = T * SR;
  uint32_t* lcur = gbc_sc_lds;                                   // [P]
  uint32_t* gpos = gbc_sc_lds + P;                               // [BR]
  uint64_t* stage = (uint64_t*)(gbc_sc_lds + P + BR);            // [BR][RW]   (P and BR are even: 8-byte aligned)
  const uint32_t* mine = mat + (size_t)blockIdx.x * P;
  for (int s = tid; s < P; s += T) lcur[s] = mine[s];
  __syncthreads();
  const int64_t lo = (int64_t)blockIdx.x * rows_per_wg;
  const int64_t hi = lo + rows_per_wg < n ? lo + rows_per_wg : n;
  // rows of the batch at t0 (the NEXT batch's loads go out before the copy-out of the current one: a 1024-thread workgroup is
  // alone on its CU, and without that nothing would be on its way from HBM while it stages and stores)
  auto load_batch = [&](int64_t t0, bool (&in)[SR], uint64_t (&k)[SR][KW], uint64_t (&v)[SR][NV]) {
    int64_t row[SR];
#pragma unroll
    for (int x = 0; x < SR; ++x) {
      const int64_t li = t0 + (int64_t)x * T + tid;
      in[x] = li < hi;
      row[x] = row0 + (in[x] ? li : lo);
      in[x] = in[x] && gb_row_passes(C, row[x]);
    }
    gbc_load_keys<KW, SR>(D, C, row, k);
    gbc_load_values<NV, SR>(D, C, row, v);
  };
  bool in[SR];
  uint64_t k[SR][KW];
  uint64_t v[SR][NV];
  if (lo < hi) load_batch(lo, in, k, v);
  for (int64_t t0 = lo; t0 < hi; t0 += BR) {
#pragma unroll
    for (int x = 0; x < SR; ++x) {
      uint32_t pos = 0xFFFFFFFFu;
      if (in[x]) pos = atomicAdd(&lcur[part_of(gbc_hash<KW>(D, k[x]), pbits)], 1u);
      const int sr = x * T + tid;
      gpos[sr] = pos;
      uint64_t* st = stage + (size_t)sr * RW;
#pragma unroll
      for (int j = 0; j < KW; ++j) st[j] = k[x][j];
#pragma unroll
      for (int j = 0; j < NV; ++j) st[KW + j] = v[x][j];
    }
    if (t0 + BR < hi) load_batch(t0 + BR, in, k, v);   // (uniform; the registers of the staged batch are free again)
    __syncthreads();
#pragma unroll
    for (int x = 0; x < SR; ++x) {
      const int sr = x * T + tid;
      const uint32_t g = gpos[sr];
      if (g != 0xFFFFFFFFu) {
        const uint64_t* st = stage + (size_t)sr * RW;
        uint64_t* o = rows_out + (uint64_t)g * RW;
#pragma unroll
        for (int j = 0; j < RW; ++j) o[j] = st[j];   // (plain stores: a partition's run is assembled line by line in the L2; r04c measured
                                                     //  nontemporal stores here at 3.1 ms instead of 0.55 ms per 60 M rows with 256 partitions)
      }
    }
    __syncthreads();
  }
}

// ---- scatter without a histogram pass ------------------------------------------------------------------------------------------------
// Up to 1024 partitions. Every partition owns a fixed region of `part_cap` rows (the rows a uniform hash gives it plus slack); a
// workgroup ranks the rows of a batch per partition in LDS (the LDS atomic's return value IS the rank), reserves each partition's
// run with ONE global atomic per (batch, partition) — 4096 rows per batch: at most a quarter of an atomic per row at 1024
// partitions, 1/256 at 16 — and copies the staged rows out behind the reserved position. No histogram kernel, no scans (r04g: hist
// 0.11 + scans 0.06 of 1.3 ms at 10^4 groups), and the key columns are read once instead of twice. A partition that outgrows its
// region (heavy keys) raises ctrl[3] bit 3: the host redoes the chunk through the exact histogram path.
template <int KW, int NV>
__global__ __launch_bounds__(GBC_T) void gbc_scatter_direct_kernel(GbcDesc D, GbCols C, int64_t row0, int64_t n, int pbits, int64_t rows_per_wg,
                                                                   uint32_t part_cap, uint32_t* gcursor, uint64_t* rows_out, uint64_t* ctrl) {
  extern __shared__ uint32_t gbc_sd_lds[];
  constexpr int RW = KW + NV;
  constexpr int SR = RW <= 2 ? 4 : (RW <= 4 ? 2 : 1);
  const int P = 1 << pbits;
  const int T = blockDim.x;
  const int tid = threadIdx.x;
  const int BR = T * SR;                                         // <= 4096 rows: a rank fits 12 bits
  uint32_t* lhist = gbc_sd_lds;                                  // [P] rows of the batch per partition
  uint32_t* lbase = gbc_sd_lds + P;                              // [P] reserved position of the batch's run
  uint32_t* gpos = gbc_sd_lds + 2 * P;                           // [BR] partition << 12 | rank, ~0 = no row
  uint64_t* stage = (uint64_t*)(gbc_sd_lds + 2 * P + BR);        // [BR][RW]
  for (int s = tid; s < P; s += T) lhist[s] = 0;
  __syncthreads();
  const int64_t lo = (int64_t)blockIdx.x * rows_per_wg;
  const int64_t hi = lo + rows_per_wg < n ? lo + rows_per_wg : n;
  auto load_batch = [&](int64_t t0, bool (&in)[SR], uint64_t (&k)[SR][KW], uint64_t (&v)[SR][NV]) {
    int64_t row[SR];
#pragma unroll
    for (int x = 0; x < SR; ++x) {
      const int64_t li = t0 + (int64_t)x * T + tid;
      in[x] = li < hi;
      row[x] = row0 + (in[x] ? li : lo);
      in[x] = in[x] && gb_row_passes(C, row[x]);
    }
    gbc_load_keys<KW, SR>(D, C, row, k);
    gbc_load_values<NV, SR>(D, C, row, v);
  };
  bool in[SR];
  uint64_t k[SR][KW];
  uint64_t v[SR][NV];
  if (lo < hi) load_batch(lo, in, k, v);
  bool overflow = false;
  for (int64_t t0 = lo; t0 < hi; t0 += BR) {
#pragma unroll
    for (int x = 0; x < SR; ++x) {
      uint32_t pr = 0xFFFFFFFFu;
      if (in[x]) {
        const uint32_t p = part_of(gbc_hash<KW>(D, k[x]), pbits);
        pr = (p << 12) | atomicAdd(&lhist[p], 1u);
      }
      const int sr = x * T + tid;
      gpos[sr] = pr;
      uint64_t* st = stage + (size_t)sr * RW;
#pragma unroll
      for (int j = 0; j < KW; ++j) st[j] = k[x][j];
#pragma unroll
      for (int j = 0; j < NV; ++j) st[KW + j] = v[x][j];
    }
    if (t0 + BR < hi) load_batch(t0 + BR, in, k, v);   // (the next batch's loads fly during the reservation and the copy-out)
    __syncthreads();
    for (int s = tid; s < P; s += T) {
      const uint32_t c = lhist[s];
      if (c) { lbase[s] = atomicAdd(&gcursor[s], c); lhist[s] = 0; }
    }
    __syncthreads();
#pragma unroll
    for (int x = 0; x < SR; ++x) {
      const int sr = x * T + tid;
      const uint32_t pr = gpos[sr];
      if (pr != 0xFFFFFFFFu) {
        const uint32_t p = pr >> 12, at = lbase[p] + (pr & 4095u);
        if (at < part_cap) {
          const uint64_t* st = stage + (size_t)sr * RW;
          uint64_t* o = rows_out + ((uint64_t)p * part_cap + at) * RW;
#pragma unroll
          for (int j = 0; j < RW; ++j) o[j] = st[j];
        } else overflow = true;
      }
    }
    __syncthreads();
  }
  if (__ballot(overflow) && lane_id() == 0) atomicOr((unsigned long long*)&ctrl[3], 8ULL);
}

// ---- LDS aggregation ------------------------------------------------------------------------------------------------------------------
struct GbcAggArgs {
  // FROM_COLS: rows [row0, row0 + n) of the columns, tiles of blockDim.x * 4 rows dealt round-robin to the workgroups
  int64_t row0, n;
  // partitions: compact rows grouped by partition, base[p] .. base[p + 1], `splits` workgroups per partition
  const uint64_t* rows;
  const uint32_t* base;
  const uint32_t* pcursor;   // non-NULL (gbc_scatter_direct_kernel's output): partition p = rows [p * part_cap, p * part_cap + min(pcursor[p], part_cap))
  uint32_t part_cap;
  int splits;
  int lcap;                // LDS table slots (power of two)
  uint32_t llimit;         // most slots a table may hold
  uint64_t* partial;       // [gridDim.x * lcap][W] partial rows in table layout
  uint32_t* pcount;        // non-NULL: workgroup b keeps its partial rows at partial[b * lcap ...] and their number here; NULL: one packed list (ctrl[5])
  uint64_t* spill;         // [spill_cap][W] rows that did not fit, table layout
  uint64_t spill_cap;
  uint64_t* ctrl;          // [5] += partial rows, [6] += spilled rows, [3] |= 4 when the spill buffer overflowed
  // heavy partitions (gbc_split_map_kernel): partition p is worked on in nsp[p] >= splits sub-ranges; sub-ranges splits .. nsp[p] - 1 belong to
  // EXTRA workgroups (blockIdx.x >= nparts * splits), extra_map[e] = p | sub-range << 16, *extra_n of them. With per-partition lists
  // (pcount) the partial rows of a partition in more than one sub-range go to a packed list at partial[packed_base ...] (cursor: ctrl[7]).
  const uint32_t* nsp;
  const uint32_t* extra_n;
  const uint32_t* extra_map;
  int nparts;
  uint64_t packed_base;
};

// Sub-ranges per partition: ceil(rows / max_rows), at least `splits`. out = nsp[P] | (unused) | number of extra workgroups | map[extra_max].
// (sum over the partitions of the sub-ranges beyond `splits` <= rows / max_rows: extra_max = that + 1 always suffices)
__global__ __launch_bounds__(1024) void gbc_split_map_kernel(const uint32_t* pcursor, uint32_t part_cap, const uint32_t* base, int P, int splits,
                                                             uint32_t max_rows, uint32_t extra_max, uint32_t* out) {
  __shared__ uint32_t sc[1024];
  const int tid = threadIdx.x;
  const int per = (P + 1023) / 1024;
  uint32_t mine = 0;
  for (int i = 0; i < per; ++i) {
    const int p = tid * per + i;
    if (p < P) {
      const uint32_t len = pcursor ? (pcursor[p] < part_cap ? pcursor[p] : part_cap) : base[p + 1] - base[p];
      uint32_t n = (len + max_rows - 1) / max_rows;
      if (n < (uint32_t)splits) n = (uint32_t)splits;
      if (n > 65535u) n = 65535u;
      out[p] = n;
      mine += n - (uint32_t)splits;
    }
  }
  sc[tid] = mine;
  __syncthreads();
  for (int d = 1; d < 1024; d <<= 1) {
    const uint32_t v = tid >= d ? sc[tid - d] : 0;
    __syncthreads();
    sc[tid] += v;
    __syncthreads();
  }
  uint32_t off = sc[tid] - mine;
  if (tid == 1023) out[P + 1] = sc[1023] < extra_max ? sc[1023] : extra_max;
  uint32_t* map = out + P + 2;
  for (int i = 0; i < per; ++i) {
    const int p = tid * per + i;
    if (p < P) {
      const uint32_t n = out[p];
      for (uint32_t j = (uint32_t)splits; j < n; ++j) {
        if (off < extra_max) map[off] = (uint32_t)p | (j << 16);
        ++off;
      }
    }
  }
}

// The workgroup's table. States are PRE-SET to the identity of a fresh group when the kernel starts, so a claim only has to publish
// the key:
//   KW == 1  lrow[slot][0] is the key itself, GBC_EMPTY_KEY = free; find = one LDS read (hit) or one CAS on the key word (claim) —
//            no tag, no lock. A real key equal to the sentinel is handed to the row path like a row that met a full table.
//   KW == 2  a 32-bit tag per slot (0 free, 1 being published, else hash bits | 2): the winner of the tag CAS publishes both key
//            words and then the tag; a reader of 1 looks again (the publisher never waits for anybody).
constexpr uint64_t GBC_EMPTY_KEY = 0x9E3779B97F4A7C15ULL;

// Settles the rows of `pend` (bit x = row x of this lane still has no slot): every round issues the LDS reads of ALL pending rows of
// the lane together and then looks at them — the probe walks of a lane's R rows overlap instead of running one after the other
// (r04c counters: waves parked 63 % of their cycles, most of it in four back-to-back walks of dependent LDS round trips per tile).
template <int KW, int R>
__device__ __forceinline__ void gbc_settle(uint32_t* ltag, uint64_t* lrow, int LS, uint32_t lmask, uint32_t llimit, uint32_t* lcount,
                                           const uint64_t (&h)[R], const uint64_t (&k)[R][KW], uint32_t (&pos)[R], uint32_t (&slot)[R],
                                           uint32_t pend) {
  uint32_t steps[R];
#pragma unroll
  for (int x = 0; x < R; ++x) steps[x] = 0;
  while (pend) {
    uint64_t c0[R];
    uint32_t ct[R];
#pragma unroll
    for (int x = 0; x < R; ++x) {
      c0[x] = 0; ct[x] = 0;
      if ((pend >> x) & 1u) {
        if (KW == 1) c0[x] = __hip_atomic_load(lrow + (size_t)pos[x] * LS, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else ct[x] = __hip_atomic_load(&ltag[pos[x]], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
#pragma unroll
    for (int x = 0; x < R; ++x) {
      if (!((pend >> x) & 1u)) continue;
      bool done = false, advance = false;
      if (KW == 1) {
        unsigned long long* kp = (unsigned long long*)(lrow + (size_t)pos[x] * LS);
        if (c0[x] == k[x][0]) { slot[x] = pos[x]; done = true; }
        else if (c0[x] == GBC_EMPTY_KEY) {
          if (__hip_atomic_load(lcount, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= llimit) { slot[x] = GBC_FULL; done = true; }
          else {
            unsigned long long expect = GBC_EMPTY_KEY;
            if (__hip_atomic_compare_exchange_strong(kp, &expect, (unsigned long long)k[x][0], __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                     __HIP_MEMORY_SCOPE_WORKGROUP)) {
              atomicAdd(lcount, 1u);
              slot[x] = pos[x]; done = true;
            } else if (expect == k[x][0]) { slot[x] = pos[x]; done = true; }   // the same group, claimed by another lane meanwhile
            else advance = true;
          }
        } else advance = true;
      } else {
        const uint32_t tag = (uint32_t)(h[x] >> 16) | 2u;    // never 0 (free) or 1 (being published)
        if (ct[x] == tag) {
          uint64_t* d = lrow + (size_t)pos[x] * LS;
          bool eq = true;
#pragma unroll
          for (int j = 0; j < KW; ++j) eq &= __hip_atomic_load(&d[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == k[x][j];
          if (eq) { slot[x] = pos[x]; done = true; } else advance = true;
        } else if (ct[x] == 0u) {
          if (__hip_atomic_load(lcount, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= llimit) { slot[x] = GBC_FULL; done = true; }
          else {
            uint32_t expect = 0u;
            if (__hip_atomic_compare_exchange_strong(&ltag[pos[x]], &expect, 1u, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) {
              atomicAdd(lcount, 1u);
              uint64_t* d = lrow + (size_t)pos[x] * LS;
#pragma unroll
              for (int j = 0; j < KW; ++j) __hip_atomic_store(&d[j], k[x][j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
              __hip_atomic_store(&ltag[pos[x]], tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
              slot[x] = pos[x]; done = true;
            }
            // else: someone else took the slot — look at it again next round (same position)
          }
        } else if (ct[x] != 1u) advance = true;   // (1: being published by another lane — look again)
      }
      if (advance) {
        pos[x] = (pos[x] + 1) & lmask;
        if (++steps[x] > lmask) { slot[x] = GBC_FULL; done = true; }
      }
      if (done) pend &= ~(1u << x);
    }
  }
}

// the compact rows of one tile (R per lane): from the input columns, or from a partition's compact rows
template <int KW, int NV, bool FROM_COLS, int R>
__device__ __forceinline__ void gbc_load_tile(const GbcDesc& D, const GbCols& C, const GbcAggArgs& A, int64_t t0, int64_t t_begin, int64_t t_end,
                                              int T, int tid, bool (&in)[R], uint64_t (&k)[R][KW], uint64_t (&v)[R][NV]) {
  constexpr int RW = KW + NV;
  if (FROM_COLS) {
    int64_t row[R];
#pragma unroll
    for (int x = 0; x < R; ++x) {
      const int64_t li = t0 + (int64_t)x * T + tid;
      in[x] = li < t_end;
      row[x] = A.row0 + (in[x] ? li : 0);
      in[x] = in[x] && gb_row_passes(C, row[x]);
    }
    gbc_load_keys<KW, R>(D, C, row, k);
    gbc_load_values<NV, R>(D, C, row, v);
  } else {
#pragma unroll
    for (int x = 0; x < R; ++x) {
      const int64_t ri = t0 + (int64_t)x * T + tid;
      in[x] = ri < t_end;
      const uint64_t* r = A.rows + (uint64_t)(in[x] ? ri : t_begin) * RW;
      uint64_t w[RW];
      if (RW % 2 == 0) {     // 16-byte rows and multiples: one 16-byte load per pair of words (the buffer is 256-byte aligned)
        typedef unsigned long long gbc_u64x2 __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int j = 0; j < RW; j += 2) { const gbc_u64x2 q = *(const gbc_u64x2*)(r + j); w[j] = q.x; w[j + 1] = q.y; }
      } else {
#pragma unroll
        for (int j = 0; j < RW; ++j) w[j] = r[j];
      }
#pragma unroll
      for (int j = 0; j < KW; ++j) k[x][j] = w[j];
#pragma unroll
      for (int j = 0; j < NV; ++j) v[x][j] = w[KW + j];
    }
  }
}

// merge the contributions of R rows of a lane into their slots (LDS atomics); the switch on the aggregate is outside the row loop
template <int KW, int NV, int R>
__device__ __forceinline__ void gbc_merge(const GbcDesc& D, uint64_t* lrow, int LS, const uint32_t (&slot)[R], const uint64_t (&v)[R][NV]) {
  uint64_t vm[R];
  if (D.vmw >= 0) gbc_pick<NV, R>(v, D.vmw, vm);
  for (int a = 0; a < D.naggs; ++a) {
    const int so = KW + D.off[a];
    uint64_t val[R];
    bool ok[R];
    gbc_pick<NV, R>(v, D.val[a], val);
#pragma unroll
    for (int x = 0; x < R; ++x) ok[x] = slot[x] < GBC_NONE;
    if (D.abit[a] >= 0) {
#pragma unroll
      for (int x = 0; x < R; ++x) ok[x] = ok[x] && ((vm[x] >> D.abit[a]) & 1ULL);
    }
    switch (D.op[a]) {
      case GBC_COUNT:
#pragma unroll
        for (int x = 0; x < R; ++x) if (ok[x]) atomicAdd((unsigned long long*)(lrow + (size_t)slot[x] * LS + so), 1ULL);
        break;
      case GBC_SUM_INT:
#pragma unroll
        for (int x = 0; x < R; ++x) if (ok[x]) atomicAdd((unsigned long long*)(lrow + (size_t)slot[x] * LS + so), (unsigned long long)val[x]);
        break;
      case GBC_SUM_F32:
#pragma unroll
        for (int x = 0; x < R; ++x) if (ok[x]) atomicAdd((double*)(lrow + (size_t)slot[x] * LS + so), (double)__uint_as_float((uint32_t)val[x]));
        break;
      case GBC_SUM_F64:
#pragma unroll
        for (int x = 0; x < R; ++x) if (ok[x]) atomicAdd((double*)(lrow + (size_t)slot[x] * LS + so), __longlong_as_double((long long)val[x]));
        break;
      case GBC_SUM_I128: {
        uint64_t hi[R];
        gbc_pick<NV, R>(v, D.val[a] + 1, hi);
#pragma unroll
        for (int x = 0; x < R; ++x) if (ok[x]) atomic_add_u192(lrow + (size_t)slot[x] * LS + so, val[x], hi[x], (hi[x] >> 63) ? ~0ULL : 0ULL);
      } break;
      case GBC_MIN: {
        const int t = D.type[a];
#pragma unroll
        for (int x = 0; x < R; ++x) if (ok[x]) atomicMin((unsigned long long*)(lrow + (size_t)slot[x] * LS + so), (unsigned long long)ord_encode(val[x], t));
      } break;
      default: {
        const int t = D.type[a];
#pragma unroll
        for (int x = 0; x < R; ++x) if (ok[x]) atomicMax((unsigned long long*)(lrow + (size_t)slot[x] * LS + so), (unsigned long long)ord_encode(val[x], t));
      } break;
    }
    if (D.fset[a]) {   // "merged a non-NULL row" (every writer stores the same 1)
      const int fo = so + D.fword[a];
#pragma unroll
      for (int x = 0; x < R; ++x) if (ok[x]) __hip_atomic_store(lrow + (size_t)slot[x] * LS + fo, (uint64_t)1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  }
}

// rows that found the table full leave in table layout for the row path
template <int KW, int NV>
__device__ __forceinline__ void gbc_spill_row(const GbcDesc& D, const GbcAggArgs& A, bool spill, const uint64_t (&k)[KW], const uint64_t (&v)[NV], uint64_t h) {
  const uint64_t m = __ballot(spill);
  if (!m) return;
  unsigned long long sb = 0;
  const int leader = __ffsll((long long)m) - 1;
  if (lane_id() == leader) sb = atomicAdd((unsigned long long*)&A.ctrl[6], (unsigned long long)__popcll(m));
  sb = __shfl(sb, leader, 64);
  if (spill) {
    const unsigned long long at = sb + __popcll(m & ((1ULL << lane_id()) - 1));
    if (at < A.spill_cap) gbc_write_full_row<KW, NV>(D, k, v, h, A.spill + at * D.W);
    else atomicOr((unsigned long long*)&A.ctrl[3], 4ULL);
  }
}

// One lane, one row: 64 deferred rows of the wave's queue (or what is left of it) are settled with every lane busy. The queue holds
// the compact rows themselves, or (rows of more than 6 words) their positions — the row is then read again from where it came from.
template <int KW, int NV, bool FROM_COLS>
__device__ __forceinline__ void gbc_drain(const GbcDesc& D, const GbCols& C, const GbcAggArgs& A, uint32_t* ltag, uint64_t* lrow, int LS, uint32_t lmask,
                                          uint32_t* lcount, const uint64_t* q, int first, int count) {
  constexpr int RW = KW + NV;
  const int lane = lane_id();
  const bool in = lane < count;
  uint64_t k[1][KW], v[1][NV], h[1];
  if (gbc_queue_of_positions(RW)) {
    const int64_t at = (int64_t)q[first + (in ? lane : 0)];
    if (FROM_COLS) {
      const int64_t row[1] = {A.row0 + at};
      gbc_load_keys<KW, 1>(D, C, row, k);
      gbc_load_values<NV, 1>(D, C, row, v);
    } else {
      const uint64_t* r = A.rows + (uint64_t)at * RW;
#pragma unroll
      for (int j = 0; j < KW; ++j) k[0][j] = r[j];
#pragma unroll
      for (int j = 0; j < NV; ++j) v[0][j] = r[KW + j];
    }
  } else {
    const uint64_t* r = q + (size_t)(first + (in ? lane : 0)) * RW;
#pragma unroll
    for (int j = 0; j < KW; ++j) k[0][j] = r[j];
#pragma unroll
    for (int j = 0; j < NV; ++j) v[0][j] = r[KW + j];
  }
  h[0] = gbc_hash<KW>(D, k[0]);
  uint32_t pos[1] = {(uint32_t)h[0] & lmask};
  uint32_t slot[1] = {in ? GBC_FULL : GBC_NONE};
  const bool sentinel = KW == 1 && k[0][0] == GBC_EMPTY_KEY;   // (a real key equal to the free-slot mark: row path, slot stays GBC_FULL)
  gbc_settle<KW, 1>(ltag, lrow, LS, lmask, A.llimit, lcount, h, k, pos, slot, (in && !sentinel) ? 1u : 0u);
  gbc_merge<KW, NV, 1>(D, lrow, LS, slot, v);
  gbc_spill_row<KW, NV>(D, A, slot[0] == GBC_FULL, k[0], v[0], h[0]);
}

// rows of the deferred-row queue of one wave
constexpr int GBC_QCAP = 128;

template <int KW, int NV, bool FROM_COLS>
__global__ __launch_bounds__(GBC_T) void gbc_agg_kernel(GbcDesc D, GbCols C, GbcAggArgs A) {
  extern __shared__ uint64_t gbc_agg_lds[];
  __shared__ uint32_t lcount;
  constexpr int RW = KW + NV;
  constexpr int R = gbc_rows_per_lane(RW);
  constexpr int QW = gbc_queue_of_positions(RW) ? 1 : RW;   // words per queue entry
  const int LS = KW + D.sw;
  const int tid = threadIdx.x;
  const int T = blockDim.x;
  uint64_t* lrow = gbc_agg_lds;                                                      // [lcap][LS]
  uint64_t* q = gbc_agg_lds + (size_t)A.lcap * LS + (size_t)(tid >> 6) * GBC_QCAP * QW;   // this wave's queue: [GBC_QCAP][QW]
  uint32_t* ltag = (uint32_t*)(gbc_agg_lds + (size_t)A.lcap * LS + (size_t)(T >> 6) * GBC_QCAP * QW);   // [lcap] (KW >= 2 only)
  const uint32_t lmask = (uint32_t)A.lcap - 1;
  int64_t t_begin, t_end, t_step;
  uint32_t nsp = 1;   // sub-ranges of this workgroup's partition
  if (FROM_COLS) {
    t_begin = (int64_t)blockIdx.x * T * R; t_end = A.n; t_step = (int64_t)gridDim.x * T * R;
  } else {
    int p, sp;
    const int regular = A.nsp ? A.nparts * A.splits : (int)gridDim.x;
    if ((int)blockIdx.x < regular) { p = blockIdx.x / A.splits; sp = blockIdx.x % A.splits; }
    else {
      const uint32_t e = blockIdx.x - (uint32_t)regular;
      if (e >= *A.extra_n) return;
      const uint32_t m = A.extra_map[e];
      p = (int)(m & 0xFFFFu); sp = (int)(m >> 16);
    }
    nsp = A.nsp ? A.nsp[p] : (uint32_t)A.splits;
    uint32_t pb, len;
    if (A.pcursor) { pb = (uint32_t)p * A.part_cap; len = A.pcursor[p] < A.part_cap ? A.pcursor[p] : A.part_cap; }
    else { pb = A.base[p]; len = A.base[p + 1] - pb; }
    t_begin = pb + (int64_t)(((uint64_t)len * (uint32_t)sp) / nsp);
    t_end = pb + (int64_t)(((uint64_t)len * ((uint32_t)sp + 1)) / nsp);
    t_step = (int64_t)T * R;
    if (t_begin >= t_end) {
      if (A.pcount && nsp == 1 && tid == 0) A.pcount[blockIdx.x] = 0;
      return;
    }
  }
  // the first tile's loads are in flight while the table is set up
  bool in[R];
  uint64_t k[R][KW];
  uint64_t v[R][NV];
  gbc_load_tile<KW, NV, FROM_COLS, R>(D, C, A, t_begin, t_begin, t_end, T, tid, in, k, v);
  for (int s = tid; s < A.lcap; s += T) {
    uint64_t* d = lrow + (size_t)s * LS;
    if (KW == 1) d[0] = GBC_EMPTY_KEY; else ltag[s] = 0;
    for (int w = 0; w < D.sw; ++w) d[KW + w] = D.ident[w];
  }
  if (tid == 0) lcount = 0;
  __syncthreads();
  int qn = 0;   // rows in this wave's queue (wave-uniform)

  for (int64_t t0 = t_begin; t0 < t_end; t0 += t_step) {
    // the NEXT tile's loads go out before this tile is worked on (a lane has 2 x R rows in flight)
    bool in_n[R];
    uint64_t k_n[R][KW];
    uint64_t v_n[R][NV];
    const bool more = t0 + t_step < t_end;   // (uniform)
    if (more) gbc_load_tile<KW, NV, FROM_COLS, R>(D, C, A, t0 + t_step, t_begin, t_end, T, tid, in_n, k_n, v_n);
    uint32_t slot[R], pos[R];
    uint64_t h[R];
    // First probe of the R rows together (R independent LDS reads in flight). Rows it settles are merged at once. A row it does not
    // settle — its group sits further down the probe path, or is new — is DEFERRED into the wave's queue: walking on here would
    // take the whole wave through a loop of dependent LDS round trips for the one or two lanes in sixty-four that need it (r04f:
    // the probe walks were 0.18 of the 0.36 ms of this kernel at 200 groups, the state merges nothing). The queue is drained 64 rows
    // at a time, one row per lane, every lane busy.
    uint64_t cur[R];
    uint32_t ct[R];
#pragma unroll
    for (int x = 0; x < R; ++x) {
      h[x] = gbc_hash<KW>(D, k[x]);
      pos[x] = (uint32_t)h[x] & lmask;
      if (KW == 1) cur[x] = __hip_atomic_load(lrow + (size_t)pos[x] * LS, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      else ct[x] = __hip_atomic_load(&ltag[pos[x]], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
#pragma unroll
    for (int x = 0; x < R; ++x) {
      bool hit;
      if (KW == 1) hit = cur[x] == k[x][0] && k[x][0] != GBC_EMPTY_KEY;
      else {
        hit = ct[x] == ((uint32_t)(h[x] >> 16) | 2u);
        if (hit) {
          const uint64_t* d = lrow + (size_t)pos[x] * LS;
#pragma unroll
          for (int j = 0; j < KW; ++j) hit &= __hip_atomic_load(&d[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == k[x][j];
        }
      }
      slot[x] = (in[x] && hit) ? pos[x] : GBC_NONE;
      const bool defer = in[x] && !hit;
      const uint64_t m = __ballot(defer);
      if (m) {   // (uniform)
        if (defer) {
          uint64_t* d = q + (size_t)(qn + __popcll(m & ((1ULL << lane_id()) - 1))) * QW;
          if (gbc_queue_of_positions(RW)) d[0] = (uint64_t)(t0 + (int64_t)x * T + tid);
          else {
#pragma unroll
            for (int j = 0; j < KW; ++j) d[j] = k[x][j];
#pragma unroll
            for (int j = 0; j < NV; ++j) d[KW + j] = v[x][j];
          }
        }
        qn += (int)__popcll(m);
        if (qn >= 64) {
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
          gbc_drain<KW, NV, FROM_COLS>(D, C, A, ltag, lrow, LS, lmask, &lcount, q, qn - 64, 64);
          qn -= 64;
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
        }
      }
    }
    gbc_merge<KW, NV, R>(D, lrow, LS, slot, v);
    if (more) {
#pragma unroll
      for (int x = 0; x < R; ++x) {
        in[x] = in_n[x];
#pragma unroll
        for (int j = 0; j < KW; ++j) k[x][j] = k_n[x][j];
#pragma unroll
        for (int j = 0; j < NV; ++j) v[x][j] = v_n[x][j];
      }
    }
  }
  while (qn > 0) {   // what is left in the queue
    const int c = qn < 64 ? qn : 64;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    gbc_drain<KW, NV, FROM_COLS>(D, C, A, ltag, lrow, LS, lmask, &lcount, q, qn - c, c);
    qn -= c;
  }
  __syncthreads();
  const uint32_t occupied = lcount;
  __syncthreads();
  // ONE global atomic per workgroup reserves the place of its partial rows (ctrl[5] is a single address: an atomic per wave and 64
  // slots — 16 K of them at 256 workgroups x 4096 slots — serialise at ~9 ns each, 0.14 ms of a 0.7 ms call, r04j); the waves take
  // their places inside the reservation from the LDS counter
  __shared__ unsigned long long wg_base;
  if (tid == 0) {
    lcount = 0;
    unsigned long long b = 0;
    if (occupied) b = atomicAdd((unsigned long long*)&A.ctrl[5], (unsigned long long)occupied);
    if (A.pcount) {
      if (nsp == 1) { A.pcount[blockIdx.x] = occupied; b = (unsigned long long)blockIdx.x * A.lcap; }   // the partition's own list
      else b = A.packed_base + (occupied ? atomicAdd((unsigned long long*)&A.ctrl[7], (unsigned long long)occupied) : 0ULL);   // a heavy partition's sub-range
    }
    wg_base = b;
  }
  __syncthreads();
  for (int s = tid; s < A.lcap; s += T) {   // (lcap and T are multiples of 64: wave-uniform)
    const uint64_t* src = lrow + (size_t)s * LS;
    const bool occ = KW == 1 ? src[0] != GBC_EMPTY_KEY : ltag[s] != 0;
    const uint64_t m = __ballot(occ);
    unsigned long long base = 0;
    if (m && lane_id() == 0) base = wg_base + atomicAdd(&lcount, (uint32_t)__popcll(m));
    base = __shfl(base, 0, 64);
    if (occ) {
      const unsigned long long idx = base + __popcll(m & ((1ULL << lane_id()) - 1));
      uint64_t* o = A.partial + idx * D.W;
      uint64_t kk[KW];
#pragma unroll
      for (int j = 0; j < KW; ++j) { kk[j] = src[j]; o[j] = kk[j]; }
      o[D.hash_word] = gbc_hash<KW>(D, kk);
      for (int w = 0; w < D.sw; ++w) o[D.state_off + w] = src[KW + w];
    }
  }
}

// LDS of one workgroup: the table (key + state words and a tag per slot) + one deferred-row queue per wave
inline int gbc_nv_class(const GbcDesc& D) { return D.nv <= 1 ? 1 : (D.nv == 2 ? 2 : (D.nv <= 4 ? 4 : 8)); }
inline int gbc_row_words(const GbcDesc& D) { return D.kw + gbc_nv_class(D); }
inline size_t gbc_agg_lds_bytes(const GbcDesc& D, int lcap, int threads) {
  const int rw = gbc_row_words(D);
  return (size_t)lcap * ((size_t)(D.kw + D.sw) * 8 + 4) + (size_t)(threads / 64) * GBC_QCAP * (gbc_queue_of_positions(rw) ? 1 : rw) * 8;
}
// largest table of one 1024-thread workgroup alone on its CU (160 KB of LDS minus the static words and some slack)
inline int gbc_max_lcap(const GbcDesc& D) {
  int c = 256;
  while (gbc_agg_lds_bytes(D, c * 2, GBC_T) <= 150 * 1024) c *= 2;
  return c;
}

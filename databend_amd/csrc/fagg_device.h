// fagg_device.h — device side of the fused filter -> map -> partial-aggregate kernel (k_fagg.hip). Compiled twice:
//   * ahead of time by hipcc (k_fagg.hip): the program and the layout are kernel arguments, interpreted per chunk;
//   * at run time by hiprtc (DBHIP_JIT defined, k_fagg.hip's jit_*): the SAME source with the program, the layout and the
//     per-word metadata as a `static constexpr FaArgs kM` and every loop over them fully unrolled, so that the compiler
//     folds the interpreter's switches into the straight-line code of THIS query and keeps the register file in VGPRs
//     instead of LDS — what rustc's monomorphisation gives the reference's kernels, and what the hand-written k_q1.hip is
//     for one query. Pointers, row counts and offsets stay kernel arguments.
#pragma once
#include "dev_expr.h"
#include "gb_device.h"

namespace {

constexpr int FA_MAX_SLOTS = 8;
constexpr int FA_KW = 4;     // key words (incl. the validity word of nullable keys)
constexpr int FA_MAXW = 12;  // state words per group
constexpr int FA_MAXA = 8;   // aggregates
#ifndef DBHIP_JIT
constexpr int FA_ROWS = 2;   // row slots per lane (interpreter: the register file is LDS, [slot][row slot][lane])
#endif

// what one state word accumulates (wave-uniform metadata, decoded on the host)
enum { W_NONE = 0, W_ADD1, W_ADD3, W_FADD, W_OR, W_MIN, W_MAX, W_CONT };
// which part of the argument value feeds the word
enum { C_LO = 0, C_HI, C_EXT, C_FLAG, C_ENC };

struct FaKeyTable {
  uint32_t count;
  uint32_t lock;
  uint64_t key[FA_MAX_SLOTS][FA_KW];
};
// the table lives in LDS: say so (a volatile access through a generic pointer is a flat_load sc0 sc1 with a full wait behind it)
typedef volatile __attribute__((address_space(3))) FaKeyTable* FaKeyTableV;
#define FA_LDS_V(T) ((FaKeyTableV)(T))

// One block of a MULTI-BLOCK launch (pipelined mode, round 6): what differs between the blocks of one query shape. The reference hands
// TransformPartialAggregate <= 65,536-row DataBlocks (settings_default.rs:142-148) — a few microseconds of kernel each, less than a
// launch costs. The pipelined table therefore launches the blocks it has queued TOGETHER: every workgroup of the grid belongs to one
// block (wgs_per_block of them per block), reads that block's pointers from this table and runs the unchanged row loop over it.
struct FaBlock {
  const void* in_data[EX_MAX_INPUTS];
  const uint8_t* in_valid[EX_MAX_INPUTS];
  int64_t in_voff[EX_MAX_INPUTS];
  const void* key_data[FA_KW];
  const uint8_t* key_valid[FA_KW];
  int64_t key_voff[FA_KW];
  const uint8_t* filter_bits;
  int64_t filter_off;
  int64_t n;
  int64_t _pad;
};

struct FaArgs {
  ExProg P;
  GbCol key[FA_KW];
  int32_t key_type[FA_KW], key_off[FA_KW], key_words[FA_KW];
  int32_t key_has_valid[FA_KW];       // the key column carries a validity Bitmap (part of the query shape: compiled in)
  int32_t has_filter;                 // a pushed-down predicate Bitmap comes with the block (filter_bits != NULL), likewise
  int32_t nkeys, nkey_words, validity_word, hash_word, W;
  int32_t naggs, nwords, state_off;   // state words start at word `state_off` of a row
  uint32_t wm[FA_MAXW];               // packed per-word metadata (see WM_*)
  const uint8_t* filter_bits;         // pushed-down predicate Bitmap (may be NULL)
  int64_t filter_off;
  int64_t n;
  uint64_t* partial_rows;             // [gridDim.x * 4 waves * SLOTS][W]
  uint64_t* ctrl;                     // [0] = #partial rows, [1] = flags (1: > SLOTS groups, 2: long string key)
  const uint64_t* blocks;             // FA_MULTI kernels: the launch's packed block table (device memory, fa_blk_words(shape) words per block), gridDim.x = nblocks * wgs_per_block
  int32_t wgs_per_block, _pad;
};
static_assert(sizeof(FaArgs) <= 4000, "kernel arguments must stay below the 4 KB kernarg segment");

// The block table of a FA_MULTI launch is PACKED by the query shape (host: fa_pack_block, k_fagg.hip; kernel: constant offsets): per block
//   [n] [input ci: data (, validity, validity offset when the shape says the input has a Bitmap)]... [key q: likewise]... [(filter bits, offset)]
// u64 words. TPC-H Q1 (7 inputs, nothing nullable): 8 words = 64 B per block against the 512-byte FaBlock — 128 blocks are 8 KB. That
// matters beyond the bytes: a pinned-host -> device hipMemcpyAsync of 16 KB < size <= 64 KB BLOCKS the calling thread on this stack (158 us
// per 32 KB copy on a busy stream against 4 us for 16 KB, tools/probes/h2d_small_copy.hip) — 64 FaBlocks per launch took the 8-thread
// sweep from 43 to 3.8 G rows/s (r06).
__host__ __device__ constexpr int fa_blk_in_off(const FaArgs& M, int ci) {
  int o = 1;
  for (int c = 0; c < ci; ++c) o += M.P.in_has_valid[c] ? 3 : 1;
  return o;
}
__host__ __device__ constexpr int fa_blk_key_off(const FaArgs& M, int q) {
  int o = fa_blk_in_off(M, M.P.n_inputs);
  for (int k = 0; k < q; ++k) o += M.key_has_valid[k] ? 3 : 1;
  return o;
}
__host__ __device__ constexpr int fa_blk_filter_off(const FaArgs& M) { return fa_blk_key_off(M, M.nkeys); }
__host__ __device__ constexpr int fa_blk_words(const FaArgs& M) { return fa_blk_filter_off(M) + (M.has_filter ? 2 : 0); }

#ifdef DBHIP_JIT
// `static constexpr FaArgs kM = {...};` — this query's program / layout / per-word metadata (pointers and row counts null:
// they are read from the kernel argument), generated by k_fagg.hip's jit_source()
#include "fagg_meta.inc"
// Row slots per lane of the specialised kernel (its register file is per-lane VGPRs): chosen per shape by the host
// (FA_META_ROWS, k_fagg.hip jit_rows: enough rows that a lane keeps >= 256 bytes in flight once every load of a chunk is issued
// up front — 4 for Q1's 68-byte rows, 16 for a 16-byte key + argument row); -DFA_JIT_ROWS=n overrides (experiments).
#if defined(FA_JIT_ROWS)
constexpr int FA_ROWS = FA_JIT_ROWS;
#else
constexpr int FA_ROWS = FA_META_ROWS;
#endif
#define FA_LOOP_META _Pragma("unroll")
#else
#define FA_LOOP_META _Pragma("nounroll")
#endif

// ---- loads of the row loop -------------------------------------------------------------------------------------------------
// Columns live in global memory and every byte of them is read once per pass: the run-time specialised kernel says both
// (global address space: no LDS-aperture check and only the vector-memory counter to wait on; non-temporal: streamed lines do
// not displace the table's). A pointer that reaches the kernel inside a by-value struct is generic to the compiler otherwise.
typedef uint32_t fa_u32x4 __attribute__((ext_vector_type(4)));
#ifdef DBHIP_JIT
#define FA_G(T, p) ((const __attribute__((address_space(1))) T*)(p))
#define FA_LD(T, p, i) __builtin_nontemporal_load(FA_G(T, p) + (i))
#else
#define FA_G(T, p) ((const T*)(p))
#define FA_LD(T, p, i) (FA_G(T, p)[i])
#endif
// element `cb + j` of an input column as its widened 64-bit register image (dev_expr.h ex_load): cb = the chunk's first row
// (wave-uniform: the pointer arithmetic is scalar), j = the lane's 32-bit offset inside the chunk
__device__ __forceinline__ uint64_t fa_in_load(const void* p, int kind, int64_t cb, uint32_t j) {
  if (kind == LK_8) return FA_LD(uint64_t, FA_G(uint64_t, p) + cb, j);
  if (kind == LK_S4) return (uint64_t)(int64_t)FA_LD(int32_t, FA_G(int32_t, p) + cb, j);
  if (kind == LK_U4) return FA_LD(uint32_t, FA_G(uint32_t, p) + cb, j);
  if (kind == LK_F4) return (uint64_t)__double_as_longlong((double)FA_LD(float, FA_G(float, p) + cb, j));
  if (kind == LK_S2) return (uint64_t)(int64_t)FA_LD(int16_t, FA_G(int16_t, p) + cb, j);
  if (kind == LK_U2) return FA_LD(uint16_t, FA_G(uint16_t, p) + cb, j);
  if (kind == LK_S1) return (uint64_t)(int64_t)FA_LD(int8_t, FA_G(int8_t, p) + cb, j);
  if (kind == LK_U1) return FA_LD(uint8_t, FA_G(uint8_t, p) + cb, j);
  return (FA_LD(uint8_t, FA_G(uint8_t, p) + ((cb) >> 3), (uint32_t)(((uint32_t)cb & 7u) + j) >> 3) >> ((((uint32_t)cb & 7u) + j) & 7u)) & 1u;  // LK_BOOL
}
// bit (bit0 + j) of an LSB-first bitmap, in two steps so that the byte load can be issued with the other loads of the chunk
__device__ __forceinline__ uint32_t fa_bit_byte(const uint8_t* bm, int64_t bit0, uint32_t j) {
  return FA_LD(uint8_t, FA_G(uint8_t, bm) + (bit0 >> 3), (((uint32_t)bit0 & 7u) + j) >> 3);
}
__device__ __forceinline__ bool fa_bit_of(uint32_t byte, int64_t bit0, uint32_t j) { return (byte >> ((((uint32_t)bit0 & 7u) + j) & 7u)) & 1u; }

// A key column's value as it lies in memory (<= 16 bytes, ONE load whatever the type), widened like gb_load_words does
__device__ __forceinline__ fa_u32x4 fa_key_raw(const void* p, int type, int64_t cb, uint32_t j) {
  fa_u32x4 r = {0u, 0u, 0u, 0u};
  uint64_t v = 0;
  switch (type) {
    case DBHIP_T_STRING: case DBHIP_T_DEC128: return FA_LD(fa_u32x4, FA_G(fa_u32x4, p) + cb, j);
    case DBHIP_T_BOOL: v = fa_in_load(p, LK_BOOL, cb, j); break;
    case DBHIP_T_I8: v = (uint64_t)(int64_t)FA_LD(int8_t, FA_G(int8_t, p) + cb, j); break;
    case DBHIP_T_I16: v = (uint64_t)(int64_t)FA_LD(int16_t, FA_G(int16_t, p) + cb, j); break;
    case DBHIP_T_I32: case DBHIP_T_DATE: v = (uint64_t)(int64_t)FA_LD(int32_t, FA_G(int32_t, p) + cb, j); break;
    case DBHIP_T_U8: v = FA_LD(uint8_t, FA_G(uint8_t, p) + cb, j); break;
    case DBHIP_T_U16: v = FA_LD(uint16_t, FA_G(uint16_t, p) + cb, j); break;
    case DBHIP_T_U32: case DBHIP_T_F32: v = FA_LD(uint32_t, FA_G(uint32_t, p) + cb, j); break;
    default: v = FA_LD(uint64_t, FA_G(uint64_t, p) + cb, j); break;   // I64 / U64 / F64 / TIMESTAMP / DEC64
  }
  r.x = (uint32_t)v; r.y = (uint32_t)(v >> 32);
  return r;
}
// raw value -> canonical key words (gb_layout.h). Strings: the 16-byte view with the bytes past len zeroed, branch-free;
// false = longer than 12 bytes (the kernel gives the block back to the row path).
__device__ __forceinline__ bool fa_key_canon(int type, fa_u32x4 r, bool valid, uint64_t (&w)[2]) {
  bool ok = true;
  if (type == DBHIP_T_STRING) {
    const uint32_t len = r.x;
    const uint32_t m1 = len >= 4 ? 0xffffffffu : (len == 0 ? 0u : (0xffffffffu >> (8 * (4 - len))));
    const uint32_t m2 = len >= 8 ? 0xffffffffu : (len <= 4 ? 0u : (0xffffffffu >> (8 * (8 - len))));
    const uint32_t m3 = len >= 12 ? 0xffffffffu : (len <= 8 ? 0u : (0xffffffffu >> (8 * (12 - len))));
    w[0] = ((uint64_t)(r.y & m1) << 32) | len;
    w[1] = ((uint64_t)(r.w & m3) << 32) | (r.z & m2);
    ok = len <= 12;
  } else {
    w[0] = (uint64_t)r.x | ((uint64_t)r.y << 32);
    w[1] = type == DBHIP_T_DEC128 ? ((uint64_t)r.z | ((uint64_t)r.w << 32)) : 0;
  }
  if (!valid) { w[0] = 0; w[1] = 0; }
  return ok;
}

// slow path, ONE lane of a wave at a time: find or append under the lock. -1 when full.
template <int SLOTS>
__device__ __forceinline__ int fa_insert(FaKeyTable* T, const uint64_t (&k)[FA_KW]) {
  while (atomicCAS(&T->lock, 0u, 1u) != 0u) __builtin_amdgcn_s_sleep(1);
  FaKeyTableV V = FA_LDS_V(T);
  const uint32_t nk = V->count;
  int slot = -1;
  for (uint32_t s = 0; s < nk; ++s)
    if (V->key[s][0] == k[0] && V->key[s][1] == k[1] && V->key[s][2] == k[2] && V->key[s][3] == k[3]) slot = (int)s;
  if (slot < 0 && nk < (uint32_t)SLOTS) {
    V->key[nk][0] = k[0]; V->key[nk][1] = k[1]; V->key[nk][2] = k[2]; V->key[nk][3] = k[3];
    __threadfence_block();
    V->count = nk + 1;
    slot = (int)nk;
  }
  __threadfence_block();
  atomicExch(&T->lock, 0u);
  return slot;
}

__device__ __forceinline__ uint64_t fa_uniform_u64(uint64_t v) {
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
  const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  return ((uint64_t)hi << 32) | lo;
}

// wave-private, scalar-register copy of the published part of the workgroup's key table
template <int SLOTS>
struct FaCache {
  uint32_t nk;
  uint64_t k[SLOTS][FA_KW];
  __device__ __forceinline__ void refresh(FaKeyTable* T) {
    FaKeyTableV V = FA_LDS_V(T);
    nk = __builtin_amdgcn_readfirstlane(V->count);
#pragma unroll
    for (int s = 0; s < SLOTS; ++s)
#pragma unroll
      for (int j = 0; j < FA_KW; ++j) k[s][j] = fa_uniform_u64(V->key[s][j]);
  }
  __device__ __forceinline__ int lookup(const uint64_t (&q)[FA_KW]) const {
    int slot = -1;
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
      const bool eq = ((uint32_t)s < nk) & (k[s][0] == q[0]) & (k[s][1] == q[1]) & (k[s][2] == q[2]) & (k[s][3] == q[3]);
      slot = eq ? s : slot;
    }
    return slot;
  }
};

// Slots of the ROWS keys of a lane (wave-convergent call): 0xF = the row does not take part, 0xE = dropped (table full).
// ONE copy of the lookup inside a loop that normally runs once; the rare part (a key this wave has not seen published: insert
// under the lock, one leader lane at a time, then re-read the table) sits behind a wave-uniform branch and is shared by all rows
// — with the miss loop inlined per row the body of the row loop grew by ~100 instructions per row slot.
template <int SLOTS, int ROWS>
__device__ __forceinline__ void fa_resolve_all(FaKeyTable* T, FaCache<SLOTS>& C, const bool (&want)[ROWS], const uint64_t (&q)[ROWS][FA_KW],
                                               int (&slot)[ROWS], uint32_t& flags) {
  for (;;) {
    bool missing = false;
#pragma unroll
    for (int k = 0; k < ROWS; ++k) {
      const int s = C.lookup(q[k]);
      slot[k] = want[k] ? s : 0xF;
      missing |= slot[k] < 0;
    }
    if (__ballot(missing) == 0) return;
    if (C.nk >= (uint32_t)SLOTS) {   // the table is full and the key is not in it: the block is given back (flags bit 0)
#pragma unroll
      for (int k = 0; k < ROWS; ++k)
        if (slot[k] < 0) { slot[k] = 0xE; flags |= 1u; }
      return;
    }
    for (;;) {   // one missing key per round, broadcast from its first lane: ONE copy of the insert for all row slots
      bool have = false;
      uint64_t lk[FA_KW];
#pragma unroll
      for (int j = 0; j < FA_KW; ++j) lk[j] = 0;
#pragma unroll
      for (int k = 0; k < ROWS; ++k) {
        const uint64_t miss = __ballot(slot[k] == -1);
        if (!have && miss) {   // wave-uniform
          const int leader = __ffsll((long long)miss) - 1;
#pragma unroll
          for (int j = 0; j < FA_KW; ++j) {
            const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)q[k][j], leader);
            const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(q[k][j] >> 32), leader);
            lk[j] = ((uint64_t)hi << 32) | lo;
          }
          have = true;
        }
      }
      if (!have) break;
      int r = 0;
      if (lane_id() == __ffsll((long long)__ballot(true)) - 1) r = fa_insert<SLOTS>(T, lk);
      r = __builtin_amdgcn_readfirstlane(r);
      if (r < 0) {   // (wave-uniform) the table is full: the block is given back whatever else this chunk holds — stop resolving at
                     // once (r03: going on cost one round per distinct missing key, 0.43 ms at 200 groups and 1.15 ms at 1000 for the
                     // optimistic attempt of a fresh table)
#pragma unroll
        for (int k = 0; k < ROWS; ++k)
          if (slot[k] < 0) { slot[k] = 0xE; flags |= 1u; }
        return;
      }
#pragma unroll
      for (int k = 0; k < ROWS; ++k) {   // every lane that holds this key is served by the re-lookup (or dropped with it)
        bool same = slot[k] == -1;
#pragma unroll
        for (int j = 0; j < FA_KW; ++j) same &= q[k][j] == lk[j];
        if (same) { slot[k] = -2; if (r < 0) flags |= 1u; }
      }
    }
    C.refresh(T);
  }
}

__device__ __forceinline__ uint64_t fa_wave_or(uint64_t v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v |= __shfl_xor(v, off, 64);
  return v;
}
__device__ __forceinline__ uint64_t fa_wave_min(uint64_t v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) { const uint64_t o = __shfl_xor(v, off, 64); v = o < v ? o : v; }
  return v;
}
__device__ __forceinline__ uint64_t fa_wave_max(uint64_t v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) { const uint64_t o = __shfl_xor(v, off, 64); v = o > v ? o : v; }
  return v;
}
__device__ __forceinline__ double fa_wave_fsum(double v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// Packed per-word metadata (one u32 per state word, read into scalar registers ONCE before the row loop: a scalar load
// per word per row — the first version — stalls the wave for its latency every time, and a `switch` per word per slot
// unrolled over 12 words x 4 slots x 2 rows made the loop body larger than the instruction cache):
//   bits 0-2 op | 3-5 comp | 6-11 LDS slot of the argument's lo word | 12 wide | 13 signed | 14 carry-in from the word
//   before | 15 enabled | 16-23 nullable inputs the argument depends on | 24-28 dbhip_type for ord_encode
#define WM_OP(m) ((m) & 7u)
#define WM_COMP(m) (((m) >> 3) & 7u)
#define WM_SLOT(m) (((m) >> 6) & 63u)
#define WM_WIDE(m) (((m) >> 12) & 1u)
#define WM_SIGNED(m) (((m) >> 13) & 1u)
#define WM_CARRY(m) (((m) >> 14) & 1u)
#define WM_EN(m) (((m) >> 15) & 1u)
#define WM_DEP(m) (((m) >> 16) & 255u)
#define WM_ENC(m) (((m) >> 24) & 31u)

// End of the kernel, round 6: ONE partial row per (workgroup, touched slot).
//   1. every wave folds the accumulators of each slot it touched over its 64 lanes — a butterfly of 6 ds_bpermute steps in which
//      ALL state words of the slot travel together (the permutes of a step are issued back to back and waited for once; rounds 2-5
//      walked the words one by one through a rolled loop with a `switch` per word: 6 dependent LDS round trips per word, ~4 us per
//      touched slot and ~24 us per launch, which a 65,536-row block paid like a 600 M-row one);
//   2. lane 0 of every wave parks its folded words in LDS, the workgroup's rows get their indices from ONE atomic on the row
//      counter (rounds 2-5: one per (wave, slot) — 8192 atomics on one address per launch of the full grid, ~80 us);
//   3. thread g folds slot g over the four waves and writes the row: key words, group hash, state words.
// Integer words are added with the carry chain their metadata describes (WM_CARRY: the 192-bit sums), flags are counts until the
// row is written; GENERAL layouts add f64 sums and min / max (value word + seen word).
template <bool GENERAL, int NW>
__device__ __forceinline__ void fa_fold_words(uint64_t (&v)[NW], const uint64_t (&o)[NW], const uint32_t (&wm)[NW]) {
  uint64_t carry = 0;
#pragma unroll
  for (int w = 0; w < NW; ++w) {
    const uint32_t m = wm[w];
    const uint32_t op = WM_OP(m);
    if (GENERAL && op == W_FADD) {
      v[w] = (uint64_t)__double_as_longlong(__longlong_as_double((long long)v[w]) + __longlong_as_double((long long)o[w]));
    } else if (GENERAL && (op == W_MIN || op == W_MAX)) {   // value word + seen word
      if (w + 1 < NW && o[w + 1]) {
        const bool take = op == W_MIN ? o[w] < v[w] : o[w] > v[w];
        v[w] = (take || !v[w + 1]) ? o[w] : v[w];
        v[w + 1] = 1;
      }
    } else if (GENERAL && op == W_CONT && WM_COMP(m) == C_FLAG) {
      // the seen-word of a min / max: folded together with its value word above
    } else {
      const unsigned long long cin = WM_CARRY(m) ? (unsigned long long)carry : 0ULL;
      unsigned long long cout;
      v[w] = (uint64_t)__builtin_addcll((unsigned long long)v[w], (unsigned long long)o[w], cin, &cout);
      carry = (uint64_t)cout;
    }
  }
}
template <bool GENERAL, int NW>
__device__ __forceinline__ void fa_wave_fold(uint64_t (&v)[NW], const uint32_t (&wm)[NW]) {
#pragma nounroll
  for (int off = 32; off >= 1; off >>= 1) {
    uint64_t o[NW];
#pragma unroll
    for (int w = 0; w < NW; ++w) o[w] = __shfl_xor(v[w], off, 64);
    fa_fold_words<GENERAL, NW>(v, o, wm);
  }
}

// GENERAL: the layout has f64 sums or min / max words (uniform branches per word); otherwise every word is an integer add
// with an optional carry-in and the accumulate step is straight-line code.
// NW: compile-time bound of the state words per group (4 or 12): unused words would still cost registers and adds.
// Occupancy: 2 workgroups per CU while the per-lane accumulators (SLOTS x NW x 2 VGPRs) leave room, else 1 (512 registers:
// the first version ran at 2 with 16 bytes of scratch per lane that the loop touched ~77 times per wave row — 16x the VMEM
// instructions of the query-specific kernel and 75 % of the wave cycles waiting).
template <int SLOTS, bool GENERAL, int NW, bool DIV = false>
__device__ __forceinline__ void fagg_body(const FaArgs& A) {
  extern __shared__ uint64_t ex_lds[];  // AOT: the register file [n_slots][FA_ROWS][256]; both: staging of one slot's accumulators at the end
  __shared__ FaKeyTable T;
  __shared__ uint32_t wm_lds[FA_MAXW];
  __shared__ uint32_t wg_touched;                 // slots any thread of the workgroup accumulated into
  __shared__ unsigned long long wg_row0;          // this workgroup's first partial row
  __shared__ uint64_t fa_xw[4][SLOTS][NW];        // the end of the kernel: every wave's folded state words of every slot
#ifdef DBHIP_JIT
  const FaArgs& M = kM;                 // metadata: compile-time constants
  uint64_t ex_rf[EX_MAX_SLOTS * FA_ROWS];   // the register file in VGPRs: every index below is a constant after unrolling
  uint64_t* const ex_regs = ex_rf;
#define EX_REG(r, k) ex_regs[(r) * FA_ROWS + (k)]
#else
  const FaArgs& M = A;
  uint64_t* const ex_regs = ex_lds;
#define EX_REG(r, k) ex_regs[((r) * FA_ROWS + (k)) * 256 + tid]
#endif
  const ExProg& P = M.P;     // program metadata
  const ExProg& PA = A.P;    // its pointers (input columns, error counters)
  const int tid = threadIdx.x, lane = tid & 63;
  if (tid == 0) { T.count = 0; T.lock = 0; wg_touched = 0; }
  if (tid < FA_MAX_SLOTS * FA_KW) T.key[tid / FA_KW][tid % FA_KW] = 0;
  if (tid < FA_MAXW) wm_lds[tid] = M.wm[tid];
  __syncthreads();

  uint32_t wm[NW];  // scalar registers (JIT: constants)
#pragma unroll
  for (int w = 0; w < NW; ++w) {
#ifdef DBHIP_JIT
    wm[w] = M.wm[w];
#else
    wm[w] = __builtin_amdgcn_readfirstlane(A.wm[w]);
#endif
  }
  uint64_t acc[SLOTS][NW];
  uint32_t touched = 0;  // bit g: this lane accumulated a row into slot g
#pragma unroll
  for (int g = 0; g < SLOTS; ++g)
#pragma unroll
    for (int w = 0; w < NW; ++w) acc[g][w] = (GENERAL && WM_OP(wm[w]) == W_MIN) ? ~0ULL : 0ULL;

  FaCache<SLOTS> C;
  C.refresh(&T);
  uint32_t flags = 0;
  const int64_t rows_per_wave = 64 * FA_ROWS;
#ifdef FA_MULTI
  // this workgroup's block: its pointers come from the block table (uniform address, constant address space: scalar loads)
  const int wpb = A.wgs_per_block;
  const int blk = (int)blockIdx.x / wpb;
  const __attribute__((address_space(4))) uint64_t* const B = (const __attribute__((address_space(4))) uint64_t*)(A.blocks + (int64_t)blk * fa_blk_words(M));
  const int64_t blk_n = (int64_t)B[0];
  const int64_t wave_global = __builtin_amdgcn_readfirstlane((int)((((int64_t)blockIdx.x - (int64_t)blk * wpb) * blockDim.x + tid) >> 6));
  const int64_t nwaves = ((int64_t)wpb * blockDim.x) >> 6;
#define FA_IN_DATA(ci) ((const void*)B[fa_blk_in_off(M, ci)])
#define FA_IN_VALID(ci) ((const uint8_t*)B[fa_blk_in_off(M, ci) + 1])
#define FA_IN_VOFF(ci) ((int64_t)B[fa_blk_in_off(M, ci) + 2])
#define FA_KEY_DATA(q) ((const void*)B[fa_blk_key_off(M, q)])
#define FA_KEY_VALID(q) ((const uint8_t*)B[fa_blk_key_off(M, q) + 1])
#define FA_KEY_VOFF(q) ((int64_t)B[fa_blk_key_off(M, q) + 2])
#define FA_FILTER_BITS ((const uint8_t*)B[fa_blk_filter_off(M)])
#define FA_FILTER_OFF ((int64_t)B[fa_blk_filter_off(M) + 1])
#else
  const int64_t blk_n = A.n;
  const int64_t wave_global = __builtin_amdgcn_readfirstlane((int)(((int64_t)blockIdx.x * blockDim.x + tid) >> 6));   // scalar
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
#define FA_IN_DATA(ci) (PA.in_data[ci])
#define FA_IN_VALID(ci) (PA.in_valid[ci])
#define FA_IN_VOFF(ci) (PA.in_voff[ci])
#define FA_KEY_DATA(q) (A.key[q].data)
#define FA_KEY_VALID(q) (A.key[q].validity)
#define FA_KEY_VOFF(q) (A.key[q].voff)
#define FA_FILTER_BITS (A.filter_bits)
#define FA_FILTER_OFF (A.filter_off)
#endif
  const int64_t nchunks = (blk_n + rows_per_wave - 1) / rows_per_wave;

  for (int64_t c = wave_global; c < nchunks; c += nwaves) {   // (wave_global is a scalar: everything derived from c is SALU work)
    const int64_t base = c * rows_per_wave;
    // Rows of this lane: base + idx[k], idx[k] = 64 k + lane clamped into the block (the tail chunk's surplus rows re-read the
    // last row and take part in nothing). Every address below is (scalar base of the chunk) + (32-bit per-lane offset): no
    // 64-bit vector arithmetic per load, and the offsets are the same registers for every column of the same width.
    const int64_t left = blk_n - 1 - base;                                 // >= 0
    const uint32_t lim = left > 0x7fffffff ? 0x7fffffffu : (uint32_t)left;
    uint32_t idx[FA_ROWS];
    int64_t row[FA_ROWS];
    bool live[FA_ROWS];
    uint32_t vmask[FA_ROWS];
#pragma unroll
    for (int k = 0; k < FA_ROWS; ++k) {
      const uint32_t i = (uint32_t)(64 * k + lane);
      live[k] = i <= lim;
      idx[k] = i < lim ? i : lim;
      row[k] = base + idx[k];
      vmask[k] = 0xFFu;
    }
    // ---- every load of the chunk first, back to back, nothing in between that waits: inputs of the program (values,
    //      validity bytes), key columns (raw values, validity bytes), the pushed-down predicate's byte ----
    uint64_t in[EX_MAX_INPUTS][FA_ROWS], hi0[FA_ROWS], hi1[FA_ROWS];  // hi words of the (at most two) 128-bit inputs
    uint32_t inv[EX_MAX_INPUTS][FA_ROWS];                             // validity BYTE of a nullable input's row
#pragma unroll
    for (int k = 0; k < FA_ROWS; ++k) { hi0[k] = 0; hi1[k] = 0; }
#pragma unroll
    for (int ci = 0; ci < EX_MAX_INPUTS; ++ci) {
      if (ci < P.n_inputs) {
        const int64_t cb = P.in_scalar[ci] ? 0 : base;
#pragma unroll
        for (int k = 0; k < FA_ROWS; ++k) {
          const uint32_t j = P.in_scalar[ci] ? 0u : idx[k];
          if (P.in_type[ci] == LK_16) {
            const fa_u32x4 v = FA_LD(fa_u32x4, FA_G(fa_u32x4, FA_IN_DATA(ci)) + cb, j);
            in[ci][k] = (uint64_t)v.x | ((uint64_t)v.y << 32);
            const uint64_t h = (uint64_t)v.z | ((uint64_t)v.w << 32);
            if (P.in_wide_ord[ci] == 0) hi0[k] = h;
            else if (P.in_wide_ord[ci] == 1) hi1[k] = h;
          } else {
            in[ci][k] = fa_in_load(FA_IN_DATA(ci), P.in_type[ci], cb, j);
          }
          inv[ci][k] = 0xFFu;
          if (P.in_has_valid[ci]) inv[ci][k] = fa_bit_byte(FA_IN_VALID(ci), FA_IN_VOFF(ci) + cb, j);
        }
      }
    }
    fa_u32x4 kraw[FA_KW][FA_ROWS];
    uint32_t kvb[FA_KW][FA_ROWS];
    FA_LOOP_META
    for (int q = 0; q < M.nkeys; ++q) {  // AOT: rolled, ONE copy of the type switch in the loop body
      const int64_t cb = M.key[q].is_scalar ? 0 : base;
#pragma unroll
      for (int k = 0; k < FA_ROWS; ++k) {
        const uint32_t j = M.key[q].is_scalar ? 0u : idx[k];
        kraw[q][k] = fa_key_raw(FA_KEY_DATA(q), M.key[q].type, cb, j);
        kvb[q][k] = 0xFFu;
        if (M.key_has_valid[q]) kvb[q][k] = fa_bit_byte(FA_KEY_VALID(q), FA_KEY_VOFF(q) + cb, j);
      }
    }
    uint32_t fbyte[FA_ROWS];
#pragma unroll
    for (int k = 0; k < FA_ROWS; ++k) {
      fbyte[k] = 0xFFu;
      if (M.has_filter) fbyte[k] = fa_bit_byte(FA_FILTER_BITS, FA_FILTER_OFF + base, idx[k]);
    }
    // ---- decode: validity bits, canonical key words ----
#pragma unroll
    for (int ci = 0; ci < EX_MAX_INPUTS; ++ci) {
      if (ci < P.n_inputs && P.in_has_valid[ci]) {
        const int64_t cb = P.in_scalar[ci] ? 0 : base;
#pragma unroll
        for (int k = 0; k < FA_ROWS; ++k)
          if (!fa_bit_of(inv[ci][k], FA_IN_VOFF(ci) + cb, P.in_scalar[ci] ? 0u : idx[k])) vmask[k] &= ~(1u << ci);
      }
    }
    uint64_t kw[FA_ROWS][FA_KW];
    uint64_t kvm[FA_ROWS];
#pragma unroll
    for (int k = 0; k < FA_ROWS; ++k) {
      kvm[k] = 0;
#pragma unroll
      for (int j = 0; j < FA_KW; ++j) kw[k][j] = 0;
    }
    FA_LOOP_META
    for (int q = 0; q < M.nkeys; ++q) {
      const int off = M.key_off[q], two = M.key_words[q] == 2;
      const int64_t cb = M.key[q].is_scalar ? 0 : base;
#pragma unroll
      for (int k = 0; k < FA_ROWS; ++k) {
        const bool valid = !M.key_has_valid[q] || fa_bit_of(kvb[q][k], FA_KEY_VOFF(q) + cb, M.key[q].is_scalar ? 0u : idx[k]);
        uint64_t w[2];
        if (!fa_key_canon(M.key[q].type, kraw[q][k], valid, w)) flags |= live[k] ? 2u : 0u;
#pragma unroll
        for (int j = 0; j < FA_KW; ++j) {
          if (j == off) kw[k][j] = w[0];
          if (two && j == off + 1) kw[k][j] = w[1];
        }
        if (valid) kvm[k] |= 1ULL << q;
      }
    }
#pragma unroll
    for (int k = 0; k < FA_ROWS; ++k) {
#pragma unroll
      for (int j = 0; j < FA_KW; ++j)
        if (j == M.validity_word) kw[k][j] = kvm[k];
      if (M.has_filter) live[k] = live[k] && fa_bit_of(fbyte[k], FA_FILTER_OFF + base, idx[k]);
    }
#pragma unroll
    for (int ci = 0; ci < EX_MAX_INPUTS; ++ci) {
      if (ci < P.n_inputs && P.in_slot[ci] >= 0) {
#pragma unroll
        for (int k = 0; k < FA_ROWS; ++k) {
          EX_REG(P.in_slot[ci], k) = in[ci][k];
          if (P.in_wide_ord[ci] >= 0) EX_REG(P.in_slot[ci] + 1, k) = P.in_wide_ord[ci] == 0 ? hi0[k] : hi1[k];
        }
      }
    }
    // ---- filter expression first, then the maps only raise for rows the filter kept (TransformFilter precedes the maps);
    //      one copy of the interpreter, two passes ----
    FA_LOOP_META
    for (int stage = 0; stage < 2; ++stage) {
      const int pc0 = stage ? P.n_filter_ins : 0, pc1 = stage ? P.n_ins : P.n_filter_ins;
      if (pc1 > pc0) ex_interpret<FA_ROWS, DIV>(P, PA, ex_regs, tid, pc0, pc1, row, live, vmask);
      if (stage == 0 && P.filter_slot >= 0) {
#pragma unroll
        for (int k = 0; k < FA_ROWS; ++k)
          live[k] = live[k] && (EX_REG(P.filter_slot, k) & 1) && ((vmask[k] & P.filter_dep) == P.filter_dep);  // a NULL predicate drops the row
      }
    }
    if (__ballot((flags & 1u) != 0)) break;   // this workgroup met more groups than it has slots: the block is given back, stop streaming
    // another wave of the workgroup may have published new keys: pick them up (uniform, rare)
    if (__builtin_amdgcn_readfirstlane(FA_LDS_V(&T)->count) != C.nk) C.refresh(&T);
    int slots_of[FA_ROWS];
    fa_resolve_all<SLOTS, FA_ROWS>(&T, C, live, kw, slots_of, flags);
#pragma unroll
    for (int k = 0; k < FA_ROWS; ++k) {
      const int slot = slots_of[k];
      // ---- contribution words of this row: branch-free selects on scalar metadata ----
      uint64_t val[NW];
#pragma unroll
      for (int w = 0; w < NW; ++w) {
        const uint32_t m = wm[w];
        const uint32_t comp = WM_COMP(m), wide = WM_WIDE(m), sgn = WM_SIGNED(m);
        const bool valid = WM_EN(m) && ((vmask[k] & WM_DEP(m)) == WM_DEP(m));
        // the word's source: the argument's lo slot, or its hi slot for the HI / EXT words of a 128-bit value
        const uint64_t raw = EX_REG(WM_SLOT(m) + ((wide && (comp == C_HI || comp == C_EXT)) ? 1u : 0u), k);
        const uint64_t sx = (uint64_t)((int64_t)raw >> 63);   // all ones iff negative
        uint64_t v = raw;                                                    // C_LO; C_HI of a wide value
        if (comp == C_HI && !wide) v = sgn ? sx : 0;                         // sign / zero extension of a 64-bit value
        if (comp == C_EXT) v = (wide || sgn) ? sx : 0;
        if (comp == C_FLAG) v = 1;
        if (GENERAL && comp == C_ENC) v = ord_encode(raw, (int)WM_ENC(m));
        val[w] = valid ? v : 0;
      }
      // ---- accumulate into the slot's registers (divergent branch per slot, skipped when no lane of the wave has it) ----
#pragma unroll
      for (int g = 0; g < SLOTS; ++g) {
        if (slot == g) {
          touched |= 1u << g;
          uint64_t carry = 0;
#pragma unroll
          for (int w = 0; w < NW; ++w) {
            const uint32_t m = wm[w];
            const uint32_t op = WM_OP(m);
            if (GENERAL && op == W_FADD) {
              acc[g][w] = (uint64_t)__double_as_longlong(__longlong_as_double((long long)acc[g][w]) + __longlong_as_double((long long)val[w]));
            } else if (GENERAL && (op == W_MIN || op == W_MAX)) {   // value word + has word
              if (w + 1 < NW && val[w + 1]) {
                const bool take = op == W_MIN ? val[w] < acc[g][w] : val[w] > acc[g][w];
                acc[g][w] = (take || !acc[g][w + 1]) ? val[w] : acc[g][w];
                acc[g][w + 1] = 1;
              }
            } else if (GENERAL && op == W_CONT && WM_COMP(m) == C_FLAG) {
              // the has-word of a min / max: written together with its value word above
            } else {
              // integer add with an optional carry-in: W_ADD1, W_ADD3 and its two continuation words, flags (kept as
              // counts of valid rows), unused words (their contribution is 0)
              // (add-with-carry through the builtin: two v_addc per word instead of two 64-bit adds, two 64-bit compares and an
              // or — the chain of the 192-bit sums is the largest block of the row loop, r02j5)
              const unsigned long long cin = WM_CARRY(m) ? (unsigned long long)carry : 0ULL;
              unsigned long long cout;
              acc[g][w] = (uint64_t)__builtin_addcll((unsigned long long)acc[g][w], (unsigned long long)val[w], cin, &cout);
              carry = (uint64_t)cout;
            }
          }
        }
      }
    }
  }
  // ---- give-up flags ----
  flags = (uint32_t)fa_wave_or((uint64_t)flags);
  if (lane == 0 && flags) atomicOr((unsigned long long*)&A.ctrl[1], (unsigned long long)flags);
  // ---- one partial row per (WORKGROUP, touched slot), see fa_wave_fold ----
  const uint32_t wt = (uint32_t)fa_wave_or((uint64_t)touched);
  if (lane == 0 && wt) atomicOr(&wg_touched, wt);
  const int wave = tid >> 6;
#pragma unroll
  for (int g = 0; g < SLOTS; ++g) {
    if ((wt >> g) & 1) fa_wave_fold<GENERAL, NW>(acc[g], wm);   // (wave-uniform; an untouched slot holds the identity in every lane)
    if (lane == 0) {
#pragma unroll
      for (int w = 0; w < NW; ++w) fa_xw[wave][g][w] = acc[g][w];
    }
  }
  __syncthreads();
  const uint32_t tw = wg_touched;   // (workgroup-uniform from here on)
  if (tw == 0) return;
  if (tid == 0) wg_row0 = atomicAdd((unsigned long long*)&A.ctrl[0], (unsigned long long)__popc(tw));
  __syncthreads();
  if (tid < SLOTS && ((tw >> tid) & 1)) {
    const int g = tid;
    uint64_t v[NW];
#pragma unroll
    for (int w = 0; w < NW; ++w) v[w] = fa_xw[0][g][w];
#pragma unroll
    for (int x = 1; x < 4; ++x) {
      uint64_t o[NW];
#pragma unroll
      for (int w = 0; w < NW; ++w) o[w] = fa_xw[x][g][w];
      fa_fold_words<GENERAL, NW>(v, o, wm);
    }
    uint64_t* r = A.partial_rows + (wg_row0 + (unsigned long long)__popc(tw & ((1u << g) - 1u))) * M.W;
    uint64_t kk[FA_KW];
#pragma unroll
    for (int j = 0; j < FA_KW; ++j) kk[j] = T.key[g][j];
    uint64_t vmk = ~0ULL;
#pragma unroll
    for (int j = 0; j < FA_KW; ++j)
      if (j == M.validity_word) vmk = kk[j];
    uint64_t h = 0;
    for (int q = 0; q < M.nkeys; ++q) {
      uint64_t w2[2] = {0, 0};
#pragma unroll
      for (int j = 0; j < FA_KW; ++j) {
        if (j == M.key_off[q]) w2[0] = kk[j];
        if (M.key_words[q] == 2 && j == M.key_off[q] + 1) w2[1] = kk[j];
      }
      const uint64_t hk = gb_hash_words(M.key_type[q], w2, (vmk >> q) & 1);
      h = q == 0 ? hk : merge_hash(h, hk);  // group_hash_entries (group_hash.rs:40-61)
    }
#pragma unroll
    for (int j = 0; j < FA_KW; ++j)
      if (j < M.nkey_words) r[j] = kk[j];
    r[M.hash_word] = h;
    uint64_t* st = r + M.state_off;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      if (w >= M.nwords) continue;
      const uint32_t op = WM_OP(wm[w]);
      if (op == W_NONE) continue;
      // flags were accumulated as counts of valid rows; a min / max seen-word likewise
      const bool flag = op == W_OR || (GENERAL && op == W_CONT && WM_COMP(wm[w]) == C_FLAG);
      st[w] = flag ? (v[w] ? 1 : 0) : v[w];
    }
  }
#undef EX_REG
#undef FA_IN_DATA
#undef FA_IN_VALID
#undef FA_IN_VOFF
#undef FA_KEY_DATA
#undef FA_KEY_VALID
#undef FA_KEY_VOFF
#undef FA_FILTER_BITS
#undef FA_FILTER_OFF
}

#ifndef DBHIP_JIT
template <int SLOTS, bool GENERAL, int NW, bool DIV = false>
__global__ __launch_bounds__(256, (SLOTS * NW <= 16) ? 2 : 1) void fagg_kernel(FaArgs A) {
  fagg_body<SLOTS, GENERAL, NW, DIV>(A);
}
#endif

}  // namespace

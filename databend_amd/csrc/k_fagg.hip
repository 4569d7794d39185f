// k_fagg.hip — generic fused TransformFilter -> BlockOperator::Map -> TransformPartialAggregate for tables with a
// handful of groups (SURVEY §8 a1/a6/a10/a11 + §8f-2): ONE pass over the unfiltered input columns.
//
// Reference pipeline (three processors, one materialised column per call node between them):
//   TransformFilter            filter/filter_executor.rs:81-118 -> selection -> DataBlock::take of every column
//   CompoundBlockOperator      sql/src/evaluator/block_operator.rs:42-85 (Evaluator::run per expression)
//   TransformPartialAggregate  aggregate_hashtable.rs:168-292 (hash -> probe -> accumulate_keys)
// Here the binding hands over the expression program (filter root + one root per aggregate argument, dev_expr.h) and
// the kernel, per chunk of 2 x 64 rows per wave:
//   loads every input column once (coalesced), interprets the program over the LDS register file,
//   resolves each passing row's key words to one of <= 8 group slots through a tiny per-workgroup LDS key table that
//     every wave caches in scalar registers (the device analogue of a cache-resident partial AggregateHashTable),
//   and accumulates the argument values into PER-LANE register accumulators (no atomics, no cross-lane traffic in
//     the loop); one wave reduction per touched slot at the very end -> <= 8 partial rows per wave, merged into the
//     HBM table by the row path exactly like partial payloads in TransformFinalAggregate.
// This is the query-specific k_q1.hip generalised: any <= 4 key words, <= 8 aggregates (count / sum incl. exact
// Decimal128 / min / max, nullable arguments), any expression program dev_expr.h interprets. A workgroup that meets a
// 9th distinct key gives up (DBHIP_ERR_CAPACITY, nothing merged): the caller keeps the operator-at-a-time kernels.
#include "dev_expr.h"
#include "gb_device.h"
#include "runtime.h"

#include <stdlib.h>
#include <string.h>

using namespace dbhip;

int32_t dbhip_groupby_merge_rows_dev_internal(dbhip_groupby* g, const uint64_t* rows, int64_t n_max, const uint64_t* n_dev,
                                              const uint64_t* abort_dev, hipStream_t s);
int32_t dbhip_groupby_merge_rows_internal(dbhip_groupby* g, const uint64_t* rows, int64_t n, hipStream_t s);
int64_t dbhip_groupby_capacity_internal(dbhip_groupby* g);
int32_t dbhip_groupby_reserve_merge_internal(dbhip_groupby* g, int64_t n);
int64_t dbhip_groupby_count_internal(dbhip_groupby* g);
const GbLayout* dbhip_groupby_layout_internal(dbhip_groupby* g);

namespace {

constexpr int FA_MAX_SLOTS = 8;
constexpr int FA_KW = 4;     // key words (incl. the validity word of nullable keys)
constexpr int FA_MAXW = 12;  // state words per group
constexpr int FA_MAXA = 8;   // aggregates
constexpr int FA_ROWS = 2;   // row slots per lane

// what one state word accumulates (wave-uniform metadata, decoded on the host)
enum { W_NONE = 0, W_ADD1, W_ADD3, W_FADD, W_OR, W_MIN, W_MAX, W_CONT };
// which part of the argument value feeds the word
enum { C_LO = 0, C_HI, C_EXT, C_FLAG, C_ENC };

struct FaKeyTable {
  uint32_t count;
  uint32_t lock;
  uint64_t key[FA_MAX_SLOTS][FA_KW];
};

struct FaArgs {
  ExProg P;
  GbCol key[FA_KW];
  int32_t key_type[FA_KW], key_off[FA_KW], key_words[FA_KW];
  int32_t nkeys, nkey_words, validity_word, hash_word, W;
  int32_t naggs, nwords, state_off;   // state words start at word `state_off` of a row
  uint32_t wm[FA_MAXW];               // packed per-word metadata (see WM_*)
  const uint8_t* filter_bits;         // pushed-down predicate Bitmap (may be NULL)
  int64_t filter_off;
  int64_t n;
  int32_t debug;                      // env DBHIP_FAGG_DEBUG bits: 1 no partial rows, 2 no key resolution / accumulation, 4 (host) no merge
  uint64_t* partial_rows;             // [gridDim.x * 4 waves * SLOTS][W]
  uint64_t* ctrl;                     // [0] = #partial rows, [1] = flags (1: > SLOTS groups, 2: long string key)
};
static_assert(sizeof(FaArgs) <= 4000, "kernel arguments must stay below the 4 KB kernarg segment");

// slow path, ONE lane of a wave at a time: find or append under the lock. -1 when full.
template <int SLOTS>
__device__ __forceinline__ int fa_insert(FaKeyTable* T, const uint64_t (&k)[FA_KW]) {
  while (atomicCAS(&T->lock, 0u, 1u) != 0u) __builtin_amdgcn_s_sleep(1);
  volatile FaKeyTable* V = T;
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
    volatile FaKeyTable* V = T;
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

// slot of one row's key (wave-convergent call). 0xF = the row does not take part, 0xE = dropped (table full).
template <int SLOTS>
__device__ __forceinline__ int fa_resolve(FaKeyTable* T, FaCache<SLOTS>& C, bool want, const uint64_t (&q)[FA_KW], uint32_t& flags) {
  int slot = C.lookup(q);
  slot = want ? slot : 0xF;
  uint64_t miss = __ballot(slot < 0);
  while (miss) {  // rare: a key this wave has not seen published yet
    const int leader = __ffsll((long long)miss) - 1;
    if (lane_id() == leader) {
      if (fa_insert<SLOTS>(T, q) < 0) flags |= 1u;
    }
    C.refresh(T);
    const int again = C.lookup(q);
    const bool full = C.nk >= (uint32_t)SLOTS;
    if (slot < 0) slot = again >= 0 ? again : (full ? 0xE : -1);
    miss = __ballot(slot < 0);
  }
  return slot;
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

// The wave's accumulators of ONE slot, staged in the lane's own cells of the LDS register file ([w][256] u64 at `stage`),
// reduced over the wave word by word and written as a partial row by lane 0. Rolled and out of line: it runs once per
// (wave, touched slot) and must not bloat the row loop's code.
__device__ __noinline__ void fa_emit_partial(const uint64_t* stage, int tid, const uint32_t* wm_lds, int nwords, uint64_t* row_states) {
  const int lane = tid & 63;
  for (int w = 0; w < nwords; ++w) {
    const uint32_t m = wm_lds[w];
    const uint64_t x = stage[w * 256 + tid];
    switch (WM_OP(m)) {
      case W_ADD1: { const uint64_t r = wave_sum_u64(x); if (lane == 0) row_states[w] = r; } break;
      case W_OR: { const uint64_t r = fa_wave_or(x); if (lane == 0) row_states[w] = r ? 1 : 0; } break;   // accumulated as a count of valid rows
      case W_FADD: { const double r = fa_wave_fsum(__longlong_as_double((long long)x)); if (lane == 0) row_states[w] = (uint64_t)__double_as_longlong(r); } break;
      case W_ADD3: {
        uint64_t e = stage[(w + 2) * 256 + tid];
        const u128 t = wave_sum_u192(((u128)stage[(w + 1) * 256 + tid] << 64) | x, &e);
        if (lane == 0) { row_states[w] = (uint64_t)t; row_states[w + 1] = (uint64_t)(t >> 64); row_states[w + 2] = e; }
        w += 2;
      } break;
      case W_MIN: {
        const uint64_t has = stage[(w + 1) * 256 + tid];
        const uint64_t r = fa_wave_min(has ? x : ~0ULL), h = fa_wave_or(has);
        if (lane == 0) { row_states[w] = r; row_states[w + 1] = h ? 1 : 0; }
        w += 1;
      } break;
      case W_MAX: {
        const uint64_t has = stage[(w + 1) * 256 + tid];
        const uint64_t r = fa_wave_max(has ? x : 0ULL), h = fa_wave_or(has);
        if (lane == 0) { row_states[w] = r; row_states[w + 1] = h ? 1 : 0; }
        w += 1;
      } break;
      default: break;
    }
  }
}

// GENERAL: the layout has f64 sums or min / max words (uniform branches per word); otherwise every word is an integer add
// with an optional carry-in and the accumulate step is straight-line code.
// NW: compile-time bound of the state words per group (4 or 12): unused words would still cost registers and adds.
// Occupancy: 2 workgroups per CU while the per-lane accumulators (SLOTS x NW x 2 VGPRs) leave room, else 1 (512 registers:
// the first version ran at 2 with 16 bytes of scratch per lane that the loop touched ~77 times per wave row — 16x the VMEM
// instructions of the query-specific kernel and 75 % of the wave cycles waiting).
template <int SLOTS, bool GENERAL, int NW>
__global__ __launch_bounds__(256, (SLOTS * NW <= 16) ? 2 : 1) void fagg_kernel(FaArgs A) {
  extern __shared__ uint64_t ex_regs[];  // [n_slots][FA_ROWS][256]; at the end: staging of one slot's accumulators
  __shared__ FaKeyTable T;
  __shared__ uint32_t wm_lds[FA_MAXW];
  const ExProg& P = A.P;
  const int tid = threadIdx.x, lane = tid & 63;
  if (tid == 0) { T.count = 0; T.lock = 0; }
  if (tid < FA_MAX_SLOTS * FA_KW) T.key[tid / FA_KW][tid % FA_KW] = 0;
  if (tid < FA_MAXW) wm_lds[tid] = A.wm[tid];
  __syncthreads();
#define EX_REG(r, k) ex_regs[((r) * FA_ROWS + (k)) * 256 + tid]

  uint32_t wm[NW];  // scalar registers
#pragma unroll
  for (int w = 0; w < NW; ++w) wm[w] = __builtin_amdgcn_readfirstlane(A.wm[w]);
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
  const int64_t nchunks = (A.n + rows_per_wave - 1) / rows_per_wave;
  const int64_t wave_global = ((int64_t)blockIdx.x * blockDim.x + tid) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;

  for (int64_t c = wave_global; c < nchunks; c += nwaves) {
    const int64_t base = c * rows_per_wave;
    int64_t row[FA_ROWS];
    bool live[FA_ROWS];
    uint32_t vmask[FA_ROWS];
#pragma unroll
    for (int k = 0; k < FA_ROWS; ++k) {
      row[k] = base + 64 * k + lane;
      live[k] = row[k] < A.n;
      vmask[k] = 0xFFu;
      if (!live[k]) row[k] = A.n - 1;  // clamp: loads stay in bounds, the row takes part in nothing
    }
    // ---- every load of the chunk first: inputs of the program, key columns, the pushed-down predicate ----
    uint64_t in[EX_MAX_INPUTS][FA_ROWS], hi0[FA_ROWS], hi1[FA_ROWS];  // hi words of the (at most two) 128-bit inputs
#pragma unroll
    for (int k = 0; k < FA_ROWS; ++k) { hi0[k] = 0; hi1[k] = 0; }
#pragma unroll
    for (int ci = 0; ci < EX_MAX_INPUTS; ++ci) {
      if (ci < P.n_inputs) {
#pragma unroll
        for (int k = 0; k < FA_ROWS; ++k) {
          const int64_t j = P.in_scalar[ci] ? 0 : row[k];
          in[ci][k] = ex_load(P.in_data[ci], P.in_type[ci], j);
          if (P.in_wide_ord[ci] == 0) hi0[k] = ((const uint64_t*)P.in_data[ci])[2 * j + 1];
          else if (P.in_wide_ord[ci] == 1) hi1[k] = ((const uint64_t*)P.in_data[ci])[2 * j + 1];
          if (P.in_valid[ci] && !bit_get(P.in_valid[ci], P.in_voff[ci] + j)) vmask[k] &= ~(1u << ci);
        }
      }
    }
    uint64_t kw[FA_ROWS][FA_KW];
#pragma unroll
    for (int k = 0; k < FA_ROWS; ++k) {
#pragma unroll
      for (int j = 0; j < FA_KW; ++j) kw[k][j] = 0;
    }
    uint64_t kvm[FA_ROWS];
#pragma unroll
    for (int k = 0; k < FA_ROWS; ++k) kvm[k] = 0;
#pragma nounroll
    for (int q = 0; q < A.nkeys; ++q) {  // rolled: ONE copy of the type switch in the loop body
      const int off = A.key_off[q], two = A.key_words[q] == 2;
#pragma unroll
      for (int k = 0; k < FA_ROWS; ++k) {
        uint64_t w[2];
        bool valid;
        if (!gb_load_words(A.key[q], row[k], w, &valid)) flags |= live[k] ? 2u : 0u;
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
        if (j == A.validity_word) kw[k][j] = kvm[k];
      if (A.filter_bits) live[k] = live[k] && bit_get(A.filter_bits, A.filter_off + row[k]);
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
#pragma nounroll
    for (int stage = 0; stage < 2; ++stage) {
      const int pc0 = stage ? P.n_filter_ins : 0, pc1 = stage ? P.n_ins : P.n_filter_ins;
      if (pc1 > pc0) ex_interpret<FA_ROWS>(P, ex_regs, tid, pc0, pc1, row, live, vmask);
      if (stage == 0 && P.filter_slot >= 0) {
#pragma unroll
        for (int k = 0; k < FA_ROWS; ++k)
          live[k] = live[k] && (EX_REG(P.filter_slot, k) & 1) && ((vmask[k] & P.filter_dep) == P.filter_dep);  // a NULL predicate drops the row
      }
    }
    if (A.debug & 2) continue;
    // another wave of the workgroup may have published new keys: pick them up (uniform, rare)
    if (__builtin_amdgcn_readfirstlane(((volatile FaKeyTable*)&T)->count) != C.nk) C.refresh(&T);
#pragma unroll
    for (int k = 0; k < FA_ROWS; ++k) {
      const int slot = fa_resolve<SLOTS>(&T, C, live[k], kw[k], flags);
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
              const uint64_t cin = WM_CARRY(m) ? carry : 0;
              const uint64_t t = acc[g][w] + val[w];
              const uint64_t r = t + cin;
              carry = (uint64_t)((t < val[w]) | (r < cin));
              acc[g][w] = r;
            }
          }
        }
      }
    }
  }
  // ---- give-up flags ----
  flags = (uint32_t)fa_wave_or((uint64_t)flags);
  if (lane == 0 && flags) atomicOr((unsigned long long*)&A.ctrl[1], (unsigned long long)flags);
  // ---- one partial row per (wave, touched slot): stage the slot's accumulators in this lane's cells of the register file ----
#pragma unroll
  for (int g = 0; g < SLOTS; ++g) {
    const uint64_t any = __ballot((touched >> g) & 1);
    if (any == 0 || (A.debug & 1)) continue;  // wave-uniform
#pragma unroll
    for (int w = 0; w < NW; ++w) ex_regs[w * 256 + tid] = acc[g][w];
    uint64_t* r = nullptr;
    if (lane == 0) {
      const unsigned long long idx = atomicAdd((unsigned long long*)&A.ctrl[0], 1ULL);
      r = A.partial_rows + idx * A.W;
      uint64_t kk[FA_KW];
#pragma unroll
      for (int j = 0; j < FA_KW; ++j) kk[j] = T.key[g][j];
      uint64_t vmk = ~0ULL;
#pragma unroll
      for (int j = 0; j < FA_KW; ++j)
        if (j == A.validity_word) vmk = kk[j];
      uint64_t h = 0;
      for (int q = 0; q < A.nkeys; ++q) {
        uint64_t w2[2] = {0, 0};
#pragma unroll
        for (int j = 0; j < FA_KW; ++j) {
          if (j == A.key_off[q]) w2[0] = kk[j];
          if (A.key_words[q] == 2 && j == A.key_off[q] + 1) w2[1] = kk[j];
        }
        const uint64_t hk = gb_hash_words(A.key_type[q], w2, (vmk >> q) & 1);
        h = q == 0 ? hk : merge_hash(h, hk);  // group_hash_entries (group_hash.rs:40-61)
      }
#pragma unroll
      for (int j = 0; j < FA_KW; ++j)
        if (j < A.nkey_words) r[j] = kk[j];
      r[A.hash_word] = h;
    }
    r = (uint64_t*)fa_uniform_u64((uint64_t)r);   // lane 0's pointer for the whole wave (NOT `hi << 32 | readfirstlane(lo)`: the
                                                  // builtin returns a signed int, whose sign extension clobbered the high half
                                                  // whenever bit 31 of the address was set — an intermittent wild store)
    fa_emit_partial(ex_regs, tid, wm_lds, A.nwords, r + A.state_off);
  }
#undef EX_REG
}

bool fa_arg_type_ok(int t) { return type_class(t) >= 0 || t == DBHIP_T_BOOL || t == DBHIP_T_DEC128; }

}  // namespace

// Can this table's layout go through the fused kernel at all? (<= 4 key words incl. validity, <= 8 aggregates, <= 12
// state words laid out back to back)
bool dbhip_fagg_layout_ok_internal(const GbLayout& L) {
  if (L.nkeys > FA_KW || L.nkey_words > FA_KW || L.naggs > FA_MAXA || L.naggs < 1) return false;
  int words = 0;
  for (int a = 0; a < L.naggs; ++a) {
    if (L.agg_off[a] != L.agg_off[0] + words) return false;
    words += L.agg_words[a];
  }
  return words <= FA_MAXW && L.hash_word == L.nkey_words && L.agg_off[0] == L.hash_word + 1;
}

extern "C" {

int32_t dbhip_groupby_add_block_program(dbhip_groupby* g, const dbhip_col* keys, const dbhip_agg_program* prog, int64_t n,
                                        const uint8_t* filter_bitmap, int64_t filter_bit_offset, void* stream) {
  DBHIP_REQUIRE(g && keys && prog && prog->arg_regs, "dbhip_groupby_add_block_program: NULL argument");
  const GbLayout& L = *dbhip_groupby_layout_internal(g);
  if (!dbhip_fagg_layout_ok_internal(L)) {
    set_error("dbhip_groupby_add_block_program: layout outside the fused kernel (<= %d key words, <= %d aggregates, <= %d state words)", FA_KW, FA_MAXA, FA_MAXW);
    return DBHIP_ERR_UNSUPPORTED;
  }
  if (n == 0) return DBHIP_OK;
  FaArgs A;
  memset(&A, 0, sizeof(A));
  // ---- roots: filter + one per aggregate argument ----
  ExRoot roots[EX_MAX_ROOTS + 1];
  int n_roots = 0, filter_root = -1;
  int root_of_agg[FA_MAXA];
  if (prog->filter_reg >= 0) { roots[n_roots].reg = prog->filter_reg; filter_root = n_roots++; }
  for (int a = 0; a < L.naggs; ++a) {
    root_of_agg[a] = -1;
    const int32_t r = prog->arg_regs[a];
    if (r == DBHIP_ARG_NONE) {
      if (L.agg_kind[a] != DBHIP_AGG_COUNT) { set_error("dbhip_groupby_add_block_program: aggregate %d needs an argument", a); return DBHIP_ERR_INVALID; }
      continue;
    }
    // the same register may feed several aggregates: one root each (the compiler maps equal registers to equal slots)
    if (n_roots >= EX_MAX_ROOTS) { set_error("dbhip_groupby_add_block_program: more than %d program results", EX_MAX_ROOTS); return DBHIP_ERR_UNSUPPORTED; }
    roots[n_roots].reg = r;
    root_of_agg[a] = n_roots++;
  }
  bool may_raise = false, any_nullable = false;
  if (n_roots == 0) {  // count(*) only: a harmless root keeps the compiler's contract (an input column, if any)
    set_error("dbhip_groupby_add_block_program: nothing to evaluate (count(*) only): use dbhip_groupby_add_block_filtered");
    return DBHIP_ERR_UNSUPPORTED;
  }
  int32_t rc = dbhip_expr_compile_internal(prog->prog, prog->n_ins, prog->inputs, prog->n_inputs, roots, n_roots, filter_root, &A.P,
                                           &may_raise, &any_nullable);
  if (rc) return rc;
  // ---- aggregates <-> roots -> packed per-word metadata ----
  A.naggs = L.naggs;
  int nwords = 0;
  bool general = false;
  for (int a = 0; a < L.naggs; ++a) {
    int slot = 0, wide = 0, sgn = 0, enc_type = L.agg_type[a];
    uint32_t dep = 0;
    const int fw = L.agg_flag[a];
    const int base_words = L.agg_words[a] - (fw ? 1 : 0);
    if (root_of_agg[a] >= 0) {
      const ExRoot& R = roots[root_of_agg[a]];
      if (L.agg_kind[a] != DBHIP_AGG_COUNT) {
        if (R.type != L.agg_type[a] || !fa_arg_type_ok(R.type)) {
          set_error("dbhip_groupby_add_block_program: aggregate %d expects an argument of type %d, the program yields type %d", a, L.agg_type[a], R.type);
          return DBHIP_ERR_INVALID;
        }
      }
      slot = R.slot; wide = R.wide; dep = R.dep;
      sgn = R.type == DBHIP_T_DEC128 || type_class(R.type) == CLS_SIGNED;
      if (R.type == DBHIP_T_F32) enc_type = DBHIP_T_F64;  // the register image of an f32 is its f64 value: same order-preserving key
    }
    auto put = [&](int op, int comp, int carry_in) {
      A.wm[nwords++] = (uint32_t)op | ((uint32_t)comp << 3) | ((uint32_t)slot << 6) | ((uint32_t)wide << 12) | ((uint32_t)sgn << 13) |
                       ((uint32_t)carry_in << 14) | (1u << 15) | ((dep & 255u) << 16) | (((uint32_t)enc_type & 31u) << 24);
    };
    switch (L.agg_kind[a]) {
      case DBHIP_AGG_COUNT: put(W_ADD1, C_FLAG, 0); break;
      case DBHIP_AGG_SUM:
        if (base_words == 3) { put(W_ADD3, C_LO, 0); put(W_CONT, C_HI, 1); put(W_CONT, C_EXT, 1); }
        else if (L.agg_type[a] == DBHIP_T_F32 || L.agg_type[a] == DBHIP_T_F64) { put(W_FADD, C_LO, 0); general = true; }
        else put(W_ADD1, C_LO, 0);
        if (fw) put(W_OR, C_FLAG, 0);
        break;
      case DBHIP_AGG_MIN: put(W_MIN, C_ENC, 0); put(W_CONT, C_FLAG, 0); general = true; break;
      default: put(W_MAX, C_ENC, 0); put(W_CONT, C_FLAG, 0); general = true; break;
    }
  }
  A.nwords = nwords;
  A.state_off = L.agg_off[0];
  // ---- keys ----
  A.nkeys = L.nkeys; A.nkey_words = L.nkey_words; A.validity_word = L.validity_word; A.hash_word = L.hash_word; A.W = L.W;
  for (int k = 0; k < L.nkeys; ++k) {
    if (keys[k].type != L.key_type[k]) { set_error("dbhip_groupby_add_block_program: key %d has type %d, table expects %d", k, keys[k].type, L.key_type[k]); return DBHIP_ERR_INVALID; }
    if (keys[k].validity && !L.key_nullable[k]) { set_error("dbhip_groupby_add_block_program: key %d carries validity but was declared NOT NULL", k); return DBHIP_ERR_INVALID; }
    GbCol& c = A.key[k];
    c.data = keys[k].data; c.validity = keys[k].validity; c.voff = keys[k].validity_offset; c.buffers = keys[k].buffers;
    c.type = keys[k].type; c.is_scalar = keys[k].is_scalar;
    A.key_type[k] = L.key_type[k]; A.key_off[k] = L.key_off[k]; A.key_words[k] = L.key_words[k];
  }
  A.filter_bits = filter_bitmap; A.filter_off = filter_bit_offset; A.n = n;
  A.debug = getenv("DBHIP_FAGG_DEBUG") ? atoi(getenv("DBHIP_FAGG_DEBUG")) : 0;
  hipStream_t s = resolve_stream(stream);
  // (at least 6 slots: the end of the kernel stages one group's 12 state words per lane in the register file)
  const size_t lds = (size_t)(A.P.n_slots > 6 ? A.P.n_slots : 6) * FA_ROWS * 256 * 8;
  if (lds > 60 * 1024) {
    set_error("dbhip_groupby_add_block_program: %d live LDS slots exceed the register file; split the expression", A.P.n_slots);
    return DBHIP_ERR_UNSUPPORTED;
  }
  // grid: whole multiples of the 256 CUs, 2 workgroups per CU (k_q1.hip's sweep: fewer, longer-running workgroups stream best)
  const int64_t nchunks = ceil_div(n, 64 * FA_ROWS);
  int grid = (int)(ceil_div(nchunks, 4) < 512 ? ceil_div(nchunks, 4) : 512);
  static const int env_grid = getenv("DBHIP_FAGG_GRID") ? atoi(getenv("DBHIP_FAGG_GRID")) : 0;
  if (env_grid > 0) grid = (int)(ceil_div(nchunks, 4) < env_grid ? ceil_div(nchunks, 4) : env_grid);
  const size_t rows_bytes = (size_t)grid * 4 * FA_MAX_SLOTS * L.W * 8;
  uint8_t* ws = (uint8_t*)scratch(rows_bytes + 64, 6);
  if (!ws) return DBHIP_ERR_HIP;
  uint64_t* ctrl = (uint64_t*)ws;
  A.ctrl = ctrl;
  A.partial_rows = (uint64_t*)(ws + 64);
  A.P.err_words = nullptr;   // row errors of fused maps: reported through ctrl[2] (count) below
  A.P.err_count = (unsigned long long*)&ctrl[2];
  uint64_t* host_ctrl = pinned_words(1);   // read back asynchronously while the merge is queued behind the kernel
  if (!host_ctrl) return DBHIP_ERR_HIP;
  host_ctrl[0] = host_ctrl[1] = host_ctrl[2] = 0;
  const int64_t n_max = (int64_t)grid * 4 * FA_MAX_SLOTS;
  const bool chained = (dbhip_groupby_count_internal(g) + n_max) * 135 <= dbhip_groupby_capacity_internal(g) * 100;
  if ((rc = dbhip_groupby_reserve_merge_internal(g, n_max))) return rc;
  static const bool no_chain = getenv("DBHIP_FAGG_NOCHAIN") != nullptr;   // debugging: drain the stream between kernel and merge
  for (int variant = 0; variant < 2; ++variant) {
    DBHIP_CHECK(hipMemsetAsync(ctrl, 0, 64, s));
    kernel_timer_start(s);
#define FA_LAUNCH(SL, GEN)                                                                                     \
  do {                                                                                                         \
    if (nwords <= 4) hipLaunchKernelGGL((fagg_kernel<SL, GEN, 4>), dim3(grid), dim3(256), lds, s, A);           \
    else hipLaunchKernelGGL((fagg_kernel<SL, GEN, FA_MAXW>), dim3(grid), dim3(256), lds, s, A);                 \
  } while (0)
    if (variant == 0 && !general) FA_LAUNCH(4, false);
    else if (variant == 0) FA_LAUNCH(4, true);
    else if (!general) FA_LAUNCH(8, false);
    else FA_LAUNCH(8, true);
#undef FA_LAUNCH
    kernel_timer_stop(s);
    DBHIP_LAUNCH_CHECK();
    DBHIP_CHECK(hipMemcpyAsync(host_ctrl, ctrl, 24, hipMemcpyDeviceToHost, s));
    if (A.debug & 4) { DBHIP_CHECK(hipStreamSynchronize(s)); return DBHIP_OK; }
    if (chained && !may_raise && !no_chain) {
      // the merge of the partial rows is queued right behind the kernel with the row count and the give-up flags still
      // on the device: ONE host round trip per pass
      rc = dbhip_groupby_merge_rows_dev_internal(g, A.partial_rows, n_max, &ctrl[0], &ctrl[1], s);
      if (rc) return rc;  // (synchronises the stream: host_ctrl is valid now)
      if (!(host_ctrl[1] & 3)) return DBHIP_OK;
    } else {
      DBHIP_CHECK(hipStreamSynchronize(s));
    }
    if (!(host_ctrl[1] & 1)) break;
  }
  if (host_ctrl[1] & 2) {
    set_error("dbhip_groupby_add_block_program: a group key string is longer than 12 bytes");
    return DBHIP_ERR_UNSUPPORTED;
  }
  if (host_ctrl[1] & 1) {
    set_error("dbhip_groupby_add_block_program: more than %d distinct groups inside one workgroup; use the operator-at-a-time path", FA_MAX_SLOTS);
    return DBHIP_ERR_CAPACITY;
  }
  if (host_ctrl[2]) {
    // a map raised for a live row: like the reference, the block fails as a whole and nothing is merged
    set_error("dbhip_groupby_add_block_program: %llu row error(s) in the fused maps (Decimal overflow / divided by zero); "
              "evaluate the maps with dbhip_expr_eval to get the rows", (unsigned long long)host_ctrl[2]);
    return DBHIP_ERR_ROW_ERRORS;
  }
  return dbhip_groupby_merge_rows_internal(g, A.partial_rows, (int64_t)host_ctrl[0], s);
}

}  // extern "C"

// add_block's own use of the fused kernel (k_groupby.hip, tables that showed <= 8 groups in their probing chunk): rows
// [row0, row0 + n) of the key / argument columns `C`, arguments as they are (an empty program whose results are input
// columns). DBHIP_ERR_UNSUPPORTED: shape outside the kernel (the caller keeps its LDS path); DBHIP_ERR_CAPACITY: a
// workgroup met a 9th group (nothing merged).
static bool fa_offset_col(const GbCol& c, int64_t row0, dbhip_col* out) {
  memset(out, 0, sizeof(*out));
  out->type = c.type; out->is_scalar = c.is_scalar; out->buffers = c.buffers; out->validity = c.validity;
  out->validity_offset = c.voff + (c.is_scalar ? 0 : row0);
  if (c.is_scalar || row0 == 0) { out->data = c.data; return true; }
  if (c.type == DBHIP_T_BOOL) {
    if (row0 & 7) return false;
    out->data = (const uint8_t*)c.data + (row0 >> 3);
    return true;
  }
  const int es = type_size(c.type);
  if (es == 0) return false;
  out->data = (const uint8_t*)c.data + (size_t)row0 * es;
  return true;
}

int32_t dbhip_fagg_add_columns_internal(dbhip_groupby* g, const GbCols& C, int64_t row0, int64_t n, hipStream_t s) {
  const GbLayout& L = *dbhip_groupby_layout_internal(g);
  if (!dbhip_fagg_layout_ok_internal(L)) return DBHIP_ERR_UNSUPPORTED;
  dbhip_col keys[FA_KW], inputs[EX_MAX_INPUTS];
  int32_t arg_regs[FA_MAXA];
  int n_inputs = 0;
  for (int k = 0; k < L.nkeys; ++k)
    if (!fa_offset_col(C.key[k], row0, &keys[k])) return DBHIP_ERR_UNSUPPORTED;
  bool any_arg = false;
  for (int a = 0; a < L.naggs; ++a) {
    arg_regs[a] = DBHIP_ARG_NONE;
    if (C.arg[a].data == nullptr) {
      if (L.agg_kind[a] != DBHIP_AGG_COUNT) return DBHIP_ERR_UNSUPPORTED;
      continue;
    }
    if (!fa_arg_type_ok(C.arg[a].type)) return DBHIP_ERR_UNSUPPORTED;
    int found = -1;   // the same column under several aggregates is loaded once
    dbhip_col col;
    if (!fa_offset_col(C.arg[a], row0, &col)) return DBHIP_ERR_UNSUPPORTED;
    for (int c = 0; c < n_inputs && found < 0; ++c)
      if (inputs[c].data == col.data && inputs[c].type == col.type && inputs[c].validity == col.validity &&
          inputs[c].validity_offset == col.validity_offset && inputs[c].is_scalar == col.is_scalar) found = c;
    if (found < 0) {
      if (n_inputs >= EX_MAX_INPUTS) return DBHIP_ERR_UNSUPPORTED;
      if (col.type == DBHIP_T_DEC128) { int wide = 0; for (int c = 0; c < n_inputs; ++c) wide += inputs[c].type == DBHIP_T_DEC128; if (wide >= 2) return DBHIP_ERR_UNSUPPORTED; }
      inputs[n_inputs] = col;
      found = n_inputs++;
    }
    arg_regs[a] = DBHIP_ARG_INPUT(found);
    any_arg = true;
  }
  if (!any_arg) return DBHIP_ERR_UNSUPPORTED;  // count(*) only: the LDS path is fine for that
  dbhip_agg_program prog;
  prog.prog = nullptr; prog.n_ins = 0; prog.inputs = inputs; prog.n_inputs = n_inputs; prog.filter_reg = -1; prog.arg_regs = arg_regs;
  return dbhip_groupby_add_block_program(g, keys, &prog, n, C.filter, C.filter_off + row0, (void*)s);
}

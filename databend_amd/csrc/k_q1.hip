// k_q1.hip — fused TPC-H Q1 pipeline (BASELINE.json configs[1]).
//
// One pass over lineitem replaces three reference processors:
//   TransformFilter          l_shipdate <= cutoff          (filters/filter_predicate.rs:71-96)
//   CompoundBlockOperator    100 - l_discount              minus<UInt8, Decimal(15,2)> -> Decimal(16,2), i64
//                            price * (..)                  multiply -> Decimal(31,4), i128, scale_mul = 0
//                            100 + l_tax ; (..) * (..)     -> Decimal(38,6), i128, scale_mul = 0
//                            (decimal/src/arithmetic.rs:80-139,190-316: plain wrapping a*b when
//                             scale_a+scale_b == result scale)
//   TransformPartialAggregate  group by (l_returnflag, l_linestatus):
//                            sum(Decimal64)->i64 wrapping, sum(Decimal128)->i128, count(*)
//                            (aggregate_sum.rs:183-300, aggregate_count.rs:99-150)
//
// HBM-bound: 68 B/row (4 x i64 + 2 x 16-B view + i32), every load a fully
// coalesced 16-B (8-B for the date) per-lane access. Per 128-row tile a wave loads
//   keys   : views of rows l and l+64          (lane l)
//   values : 16-B pairs of rows 2l, 2l+1       (lane l)
// resolves each key to one of SLOTS group slots through a tiny per-block LDS key
// table (insert is rare and wave-serialised), moves the 4-bit slot ids to the
// value lanes with two ds_bpermute, and accumulates into per-lane register
// accumulators (no atomics, no cross-lane traffic in the loop). One wave-reduce
// per accumulator at the very end; the block's <= SLOTS partial rows go to a
// scratch array that is merged into the HBM group-by table by the general
// merge path (k_groupby.hip), exactly like partial payloads are combined in
// TransformFinalAggregate (transform_aggregate_final.rs:160-175).
// More than SLOTS distinct keys in a block, or a key string longer than 12
// bytes, raises DBHIP_ERR_CAPACITY/UNSUPPORTED: the caller then runs the
// operator-at-a-time kernels (still on the GPU).
#include "gb_device.h"
#include "runtime.h"

#include <stdlib.h>
#include <string.h>

using namespace dbhip;

int32_t dbhip_groupby_merge_rows_internal(dbhip_groupby* g, const uint64_t* rows, int64_t n, hipStream_t s);
int32_t dbhip_groupby_merge_rows_dev_internal(dbhip_groupby* g, const uint64_t* rows, int64_t n_max, const uint64_t* n_dev,
                                              const uint64_t* abort_dev, hipStream_t s);
int64_t dbhip_groupby_capacity_internal(dbhip_groupby* g);
int64_t dbhip_groupby_count_internal(dbhip_groupby* g);
const GbLayout* dbhip_groupby_layout_internal(dbhip_groupby* g);

namespace {

constexpr int MAX_SLOTS = 8;
constexpr int Q1_W = 15;  // words per row of the Q1 table layout (see dbhip_q1_create_groupby)

struct alignas(16) U4 {
  uint32_t x, y, z, w;
};
struct alignas(16) L2 {
  int64_t a, b;
};
struct alignas(8) I2 {
  int32_t a, b;
};

// streaming loads: every byte of lineitem is read exactly once per pass, so the loads may carry the non-temporal hint
typedef uint32_t q1_u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t q1_u32x2 __attribute__((ext_vector_type(2)));
template <bool NT>
__device__ __forceinline__ U4 ld_u4(const U4* p) {
  if (NT) return __builtin_bit_cast(U4, __builtin_nontemporal_load((const q1_u32x4*)p));
  return *p;
}
template <bool NT>
__device__ __forceinline__ L2 ld_l2(const int64_t* p) {
  if (NT) return __builtin_bit_cast(L2, __builtin_nontemporal_load((const q1_u32x4*)p));
  return *(const L2*)p;
}
template <bool NT>
__device__ __forceinline__ I2 ld_i2(const int32_t* p) {
  if (NT) return __builtin_bit_cast(I2, __builtin_nontemporal_load((const q1_u32x2*)p));
  return *(const I2*)p;
}

// Per-block key table in LDS: append-only array of the distinct group keys seen by the
// block. Readers scan entries [0, count); a writer appends under `lock` and publishes
// by bumping `count` after a workgroup fence, so a reader never sees a half-written key.
struct KeyTable {
  uint32_t count;
  uint32_t lock;
  uint64_t key[MAX_SLOTS][4];
};

// canonical words of an inline view (bytes past len zeroed), false if len > 12
__device__ __forceinline__ bool view_words(U4 v, uint64_t w[2]) {
  uint32_t len = v.x;
  uint32_t d1 = v.y, d2 = v.z, d3 = v.w;
  uint32_t m1 = len >= 4 ? 0xffffffffu : (len == 0 ? 0u : (0xffffffffu >> (8 * (4 - len))));
  uint32_t m2 = len >= 8 ? 0xffffffffu : (len <= 4 ? 0u : (0xffffffffu >> (8 * (8 - len))));
  uint32_t m3 = len >= 12 ? 0xffffffffu : (len <= 8 ? 0u : (0xffffffffu >> (8 * (12 - len))));
  w[0] = ((uint64_t)(d1 & m1) << 32) | len;
  w[1] = ((uint64_t)(d3 & m3) << 32) | (d2 & m2);
  return len <= 12;
}

__device__ __forceinline__ uint64_t hash_view_words(const uint64_t w[2]) {
  return agg_hash_inline_view((uint32_t)w[0], (uint32_t)(w[0] >> 32), (uint32_t)w[1], (uint32_t)(w[1] >> 32));
}

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

// resolve the slot of one key row (wave-convergent call). 0xF = row not valid, 0xE = dropped.
template <int SLOTS>
__device__ __forceinline__ int resolve_slot(KeyTable* T, TabCache<SLOTS>& C, bool row_valid, U4 v0, U4 v1,
                                            uint32_t& flags) {
  uint64_t ka[2], kb[2];
  bool ok = view_words(v0, ka) & view_words(v1, kb);
  flags |= (row_valid & !ok) ? 2u : 0u;
  const bool want = row_valid & ok;
  int slot = C.lookup(ka[0], ka[1], kb[0], kb[1]);
  slot = want ? slot : 0xF;
  uint64_t miss = __ballot(slot < 0);
  while (miss) {  // rare: a key this wave has not seen published yet
    const int leader = __ffsll((long long)miss) - 1;
    if (lane_id() == leader) {
      int ls = tab_insert<SLOTS>(T, ka[0], ka[1], kb[0], kb[1]);
      if (ls < 0) flags |= 1u;
    }
    C.refresh(T);
    int again = C.lookup(ka[0], ka[1], kb[0], kb[1]);
    // after the leader's insert its key is published (or the table is full)
    const bool full = C.nk >= (uint32_t)SLOTS;
    if (slot < 0) slot = again >= 0 ? again : (full ? 0xE : -1);
    miss = __ballot(slot < 0);
  }
  return slot;
}

struct Q1Args {
  const int64_t* qty;
  const int64_t* price;
  const int64_t* disc;
  const int64_t* tax;
  const U4* rf;
  const U4* ls;
  const int32_t* shipdate;
  int32_t cutoff;
  int64_t n;
  uint64_t* partial_rows;  // [gridDim.x * SLOTS][Q1_W]
  uint64_t* ctrl;          // [0] = #partial rows, [1] = flags (1: table full, 2: long string)
};

// per-lane register accumulators, one set per slot (static indexing only)
template <int SLOTS>
struct Q1Regs {
  uint64_t qty[SLOTS], price[SLOTS], disc[SLOTS];
  uint64_t dpl[SLOTS], dph[SLOTS], chl[SLOTS], chh[SLOTS];
  int32_t dpe[SLOTS], che[SLOTS];  // bits 128.. of the exact sums (per lane an i32 is ample)
  uint32_t cnt[SLOTS];
};

template <int SLOTS>
__device__ __forceinline__ void acc_row(Q1Regs<SLOTS>& R, int slot, int64_t qty, int64_t price, int64_t disc,
                                        int64_t tax) {
  // decimal maps (see header): all plain wrapping integer ops
  const int64_t one_minus_disc = (int64_t)(100ULL - (uint64_t)disc);
  const int64_t one_plus_tax = (int64_t)(100ULL + (uint64_t)tax);
  const i128 dp = (i128)price * (i128)one_minus_disc;
  const i128 ch = (i128)((u128)dp * (u128)(i128)one_plus_tax);
  const uint64_t dpl = (uint64_t)(u128)dp, dph = (uint64_t)((u128)dp >> 64);
  const uint64_t chl = (uint64_t)(u128)ch, chh = (uint64_t)((u128)ch >> 64);
  const int32_t dpn = (int32_t)(dph >> 63), chn = (int32_t)(chh >> 63);
#pragma unroll
  for (int g = 0; g < SLOTS; ++g) {
    if (slot == g) {  // divergent branch: adds only, skipped when no lane of the wave has the slot
      R.qty[g] += (uint64_t)qty;
      R.price[g] += (uint64_t)price;
      R.disc[g] += (uint64_t)disc;
      u128 a = (((u128)R.dph[g] << 64) | R.dpl[g]) + (((u128)dph << 64) | dpl);
      u128 b = (((u128)R.chh[g] << 64) | R.chl[g]) + (((u128)chh << 64) | chl);
      R.dpe[g] += (int32_t)((uint64_t)(a >> 64) < dph || ((uint64_t)(a >> 64) == dph && (uint64_t)a < dpl)) - dpn;
      R.che[g] += (int32_t)((uint64_t)(b >> 64) < chh || ((uint64_t)(b >> 64) == chh && (uint64_t)b < chl)) - chn;
      R.dpl[g] = (uint64_t)a; R.dph[g] = (uint64_t)(a >> 64);
      R.chl[g] = (uint64_t)b; R.chh[g] = (uint64_t)(b >> 64);
      R.cnt[g] += 1u;
    }
  }
}

template <int SLOTS, bool NT>
__device__ __forceinline__ void q1_body(const Q1Args& A) {
  __shared__ KeyTable T;
  __shared__ uint64_t red[4][SLOTS][10];
  if (threadIdx.x == 0) { T.count = 0; T.lock = 0; }
  if (threadIdx.x < MAX_SLOTS * 4) T.key[threadIdx.x >> 2][threadIdx.x & 3] = 0;
  __syncthreads();

  Q1Regs<SLOTS> R;
#pragma unroll
  for (int g = 0; g < SLOTS; ++g) {
    R.qty[g] = 0; R.price[g] = 0; R.disc[g] = 0; R.dpl[g] = 0; R.dph[g] = 0; R.chl[g] = 0; R.chh[g] = 0;
    R.dpe[g] = 0; R.che[g] = 0; R.cnt[g] = 0;
  }
  TabCache<SLOTS> C;
  C.refresh(&T);
  uint32_t flags = 0;
  const int lane = lane_id();
  const int64_t ntiles = (A.n + 127) >> 7;
  const int64_t wave_global = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;

  for (int64_t t = wave_global; t < ntiles; t += nwaves) {
    const int64_t t0 = t << 7;
    const bool full = t0 + 128 <= A.n;
    // ---- loads (all issued before any use) ----
    U4 rfA, rfB, lsA, lsB;
    L2 q, p, d, x;
    I2 sd;
    const int64_t ra = t0 + lane, rb = t0 + 64 + lane;
    const int64_t r0 = t0 + 2 * lane, r1 = r0 + 1;
    if (full) {
      rfA = ld_u4<NT>(A.rf + ra); rfB = ld_u4<NT>(A.rf + rb);
      lsA = ld_u4<NT>(A.ls + ra); lsB = ld_u4<NT>(A.ls + rb);
      q = ld_l2<NT>(A.qty + r0);
      p = ld_l2<NT>(A.price + r0);
      d = ld_l2<NT>(A.disc + r0);
      x = ld_l2<NT>(A.tax + r0);
      sd = ld_i2<NT>(A.shipdate + r0);
    } else {
      const int64_t last = A.n - 1;
      const int64_t ca = ra < A.n ? ra : last, cb = rb < A.n ? rb : last;
      const int64_t c0 = r0 < A.n ? r0 : last, c1 = r1 < A.n ? r1 : last;
      rfA = A.rf[ca]; rfB = A.rf[cb];
      lsA = A.ls[ca]; lsB = A.ls[cb];
      q.a = A.qty[c0]; q.b = A.qty[c1];
      p.a = A.price[c0]; p.b = A.price[c1];
      d.a = A.disc[c0]; d.b = A.disc[c1];
      x.a = A.tax[c0]; x.b = A.tax[c1];
      sd.a = A.shipdate[c0]; sd.b = A.shipdate[c1];
    }
    // another wave of the block may have published new keys: pick them up (uniform, rare)
    if (__builtin_amdgcn_readfirstlane(((volatile KeyTable*)&T)->count) != C.nk) C.refresh(&T);
    // ---- keys -> slots (key layout) ----
    const int slotA = resolve_slot<SLOTS>(&T, C, ra < A.n, rfA, lsA, flags);
    const int slotB = resolve_slot<SLOTS>(&T, C, rb < A.n, rfB, lsB, flags);
    // ---- move slot ids to the value layout ----
    const int packed = slotA | (slotB << 8);
    const int g0 = __shfl(packed, (2 * lane) & 63, 64);
    const int g1 = __shfl(packed, (2 * lane + 1) & 63, 64);
    const int s0 = (lane < 32 ? g0 : (g0 >> 8)) & 0xFF;
    const int s1 = (lane < 32 ? g1 : (g1 >> 8)) & 0xFF;
    // ---- filter + maps + accumulate ----
    const bool pass0 = r0 < A.n && sd.a <= A.cutoff;
    const bool pass1 = r1 < A.n && sd.b <= A.cutoff;
    acc_row<SLOTS>(R, pass0 ? s0 : 0xF, q.a, p.a, d.a, x.a);
    acc_row<SLOTS>(R, pass1 ? s1 : 0xF, q.b, p.b, d.b, x.b);
  }

  // ---- wave reduce, then block combine ----
  const int wave = threadIdx.x >> 6;
#pragma unroll
  for (int g = 0; g < SLOTS; ++g) {
    uint64_t a0 = wave_sum_u64(R.qty[g]);
    uint64_t a1 = wave_sum_u64(R.price[g]);
    uint64_t a2 = wave_sum_u64(R.disc[g]);
    uint64_t e3 = (uint64_t)(int64_t)R.dpe[g], e4 = (uint64_t)(int64_t)R.che[g];
    u128 a3 = wave_sum_u192(((u128)R.dph[g] << 64) | R.dpl[g], &e3);
    u128 a4 = wave_sum_u192(((u128)R.chh[g] << 64) | R.chl[g], &e4);
    uint64_t a5 = wave_sum_u64((uint64_t)R.cnt[g]);
    if (lane == 0) {
      red[wave][g][0] = a0; red[wave][g][1] = a1; red[wave][g][2] = a2;
      red[wave][g][3] = (uint64_t)a3; red[wave][g][4] = (uint64_t)(a3 >> 64);
      red[wave][g][5] = (uint64_t)a4; red[wave][g][6] = (uint64_t)(a4 >> 64);
      red[wave][g][7] = a5;
      red[wave][g][8] = e3; red[wave][g][9] = e4;
    }
  }
  flags = (uint32_t)wave_sum_u64((uint64_t)((flags & 1) | ((flags & 2) << 15)));  // counts per flag
  if (lane == 0 && flags) atomicOr((unsigned long long*)&A.ctrl[1],
                                   (unsigned long long)(((flags & 0xFFFF) ? 1 : 0) | ((flags >> 16) ? 2 : 0)));
  __syncthreads();
  if (threadIdx.x < SLOTS && threadIdx.x < T.count) {
    const int g = threadIdx.x;
    uint64_t qty = 0, price = 0, disc = 0, cnt = 0, dpe = 0, che = 0;
    u128 dp = 0, ch = 0;
    for (int w = 0; w < 4; ++w) {
      qty += red[w][g][0]; price += red[w][g][1]; disc += red[w][g][2];
      u128 vdp = ((u128)red[w][g][4] << 64) | red[w][g][3];
      u128 vch = ((u128)red[w][g][6] << 64) | red[w][g][5];
      dp += vdp; dpe += red[w][g][8] + (dp < vdp ? 1 : 0);
      ch += vch; che += red[w][g][9] + (ch < vch ? 1 : 0);
      cnt += red[w][g][7];
    }
    if (cnt != 0) {  // a key seen only in filtered-out rows creates no group
      unsigned long long idx = atomicAdd((unsigned long long*)&A.ctrl[0], 1ULL);
      uint64_t* r = A.partial_rows + idx * Q1_W;
      // table layout: [rf view 2w][ls view 2w][hash][sum_qty][sum_price][sum_dp 3w][sum_ch 3w][sum_disc][count]
      uint64_t k[4] = {T.key[g][0], T.key[g][1], T.key[g][2], T.key[g][3]};
      r[0] = k[0]; r[1] = k[1]; r[2] = k[2]; r[3] = k[3];
      r[4] = merge_hash(hash_view_words(k), hash_view_words(k + 2));  // group_hash_entries, 2 string columns
      r[5] = qty; r[6] = price;
      r[7] = (uint64_t)dp; r[8] = (uint64_t)(dp >> 64); r[9] = dpe;
      r[10] = (uint64_t)ch; r[11] = (uint64_t)(ch >> 64); r[12] = che;
      r[13] = disc; r[14] = cnt;
    }
  }
}

__global__ __launch_bounds__(256, 2) void q1_fused_kernel(Q1Args A) { q1_body<4, true>(A); }
__global__ __launch_bounds__(256, 2) void q1_fused_kernel_plain_loads(Q1Args A) { q1_body<4, false>(A); }
__global__ __launch_bounds__(256, 1) void q1_fused_kernel_8slots(Q1Args A) { q1_body<8, true>(A); }

}  // namespace

extern "C" {

int32_t dbhip_q1_create_groupby(dbhip_groupby** out_host) {
  int32_t key_types[2] = {DBHIP_T_STRING, DBHIP_T_STRING};
  uint8_t key_nullable[2] = {0, 0};
  dbhip_agg_desc aggs[6];
  memset(aggs, 0, sizeof(aggs));
  aggs[0] = {DBHIP_AGG_SUM, DBHIP_T_DEC64, 15, 2, 0, 0};   // sum(l_quantity)
  aggs[1] = {DBHIP_AGG_SUM, DBHIP_T_DEC64, 15, 2, 0, 0};   // sum(l_extendedprice)
  aggs[2] = {DBHIP_AGG_SUM, DBHIP_T_DEC128, 31, 4, 0, 0};  // sum(price*(1-disc))
  aggs[3] = {DBHIP_AGG_SUM, DBHIP_T_DEC128, 38, 6, 0, 0};  // sum(price*(1-disc)*(1+tax))
  aggs[4] = {DBHIP_AGG_SUM, DBHIP_T_DEC64, 15, 2, 0, 0};   // sum(l_discount)
  aggs[5] = {DBHIP_AGG_COUNT, 0, 0, 0, 0, 0};              // count(*)
  // 32768 slots: the <= 16384 partial rows of a fused pass can never push the table past its load factor, so
  // their merge needs no growth check (one host round trip per pass, see merge_rows)
  return dbhip_groupby_create(key_types, key_nullable, 2, aggs, 6, 32768, out_host);
}

int32_t dbhip_q1_fused(dbhip_groupby* g, const int64_t* l_quantity, const int64_t* l_extendedprice,
                       const int64_t* l_discount, const int64_t* l_tax, const void* l_returnflag_views,
                       const void* l_linestatus_views, const int32_t* l_shipdate, int32_t shipdate_cutoff,
                       int64_t n, void* stream) {
  DBHIP_REQUIRE(g, "dbhip_q1_fused: NULL table");
  const GbLayout* L = dbhip_groupby_layout_internal(g);
  bool layout_ok = L->nkeys == 2 && L->naggs == 6 && L->W == Q1_W && L->key_type[0] == DBHIP_T_STRING &&
                   L->key_type[1] == DBHIP_T_STRING && L->agg_off[0] == 5 && L->agg_off[2] == 7 &&
                   L->agg_off[5] == 14 && L->agg_words[2] == 3 && L->agg_words[3] == 3;
  DBHIP_REQUIRE(layout_ok, "dbhip_q1_fused: table was not created by dbhip_q1_create_groupby");
  if (n == 0) return DBHIP_OK;
  DBHIP_REQUIRE(l_quantity && l_extendedprice && l_discount && l_tax && l_returnflag_views &&
                    l_linestatus_views && l_shipdate,
                "dbhip_q1_fused: NULL column");
  uintptr_t al = (uintptr_t)l_quantity | (uintptr_t)l_extendedprice | (uintptr_t)l_discount |
                 (uintptr_t)l_tax | (uintptr_t)l_returnflag_views | (uintptr_t)l_linestatus_views;
  DBHIP_REQUIRE((al & 15) == 0 && ((uintptr_t)l_shipdate & 7) == 0,
                "dbhip_q1_fused: columns must be 16-byte aligned (shipdate 8-byte)");
  hipStream_t s = resolve_stream(stream);
  const int64_t ntiles = ceil_div(n, 128);
  // Grid: whole multiples of the 256 CUs (a ragged last round costs up to 40 %), and only 2 workgroups per CU:
  // measured on SF10 (r01x) 2048 WGs 0.722 ms, 1024 0.688 ms, 512 0.680 ms; with non-temporal loads 0.689 / 0.656 /
  // 0.646 ms; 640 or 768 WGs 0.77 - 0.90 ms. Fewer workgroups also mean fewer partial rows to merge.
  int grid = (int)(ceil_div(ntiles, 4) < 512 ? ceil_div(ntiles, 4) : 512);
  static const int env_grid = exp_env("DBHIP_Q1_GRID") ? atoi(exp_env("DBHIP_Q1_GRID")) : 0;   // tuning knobs (bench experiments)
  static const int env_nt = exp_env("DBHIP_Q1_NT") ? atoi(exp_env("DBHIP_Q1_NT")) : 1;
  if (env_grid > 0) grid = (int)(ceil_div(ntiles, 4) < env_grid ? ceil_div(ntiles, 4) : env_grid);
  size_t rows_bytes = (size_t)grid * MAX_SLOTS * Q1_W * 8;
  uint8_t* ws = (uint8_t*)scratch(rows_bytes + 64, 4, s);
  if (!ws) return DBHIP_ERR_HIP;
  uint64_t* ctrl = (uint64_t*)ws;
  uint64_t* partial = (uint64_t*)(ws + 64);
  Q1Args A;
  A.qty = l_quantity; A.price = l_extendedprice; A.disc = l_discount; A.tax = l_tax;
  A.rf = (const U4*)l_returnflag_views; A.ls = (const U4*)l_linestatus_views;
  A.shipdate = l_shipdate; A.cutoff = shipdate_cutoff; A.n = n;
  A.partial_rows = partial; A.ctrl = ctrl;
  uint64_t* host_ctrl = pinned_words(1);   // read back asynchronously while the merge is queued behind the kernel
  if (!host_ctrl) return DBHIP_ERR_HIP;
  host_ctrl[0] = host_ctrl[1] = 0;
  // Adaptive slot count (the device analogue of the reference's table growth): the 4-slot
  // variant keeps every accumulator in registers at 4 waves/SIMD; a block that meets a 5th
  // distinct key flags it and the pass is redone with 8 slots; beyond that the caller uses
  // the operator-at-a-time kernels.
  // When the table cannot outgrow its load factor the merge of the partial rows is queued right behind the fused
  // kernel with the row count and the give-up flags still on the device: ONE host round trip per pass.
  const int64_t n_max4 = (int64_t)grid * 4, n_max8 = (int64_t)grid * MAX_SLOTS;
  const bool chained = (dbhip_groupby_count_internal(g) + n_max8) * 135 <= dbhip_groupby_capacity_internal(g) * 100;
  for (int variant = 0; variant < 2; ++variant) {
    DBHIP_CHECK(hipMemsetAsync(ctrl, 0, 64, s));
    kernel_timer_start(s);
    if (variant == 0 && env_nt) hipLaunchKernelGGL(q1_fused_kernel, dim3(grid), dim3(256), 0, s, A);
    else if (variant == 0) hipLaunchKernelGGL(q1_fused_kernel_plain_loads, dim3(grid), dim3(256), 0, s, A);
    else hipLaunchKernelGGL(q1_fused_kernel_8slots, dim3(grid), dim3(256), 0, s, A);
    kernel_timer_stop(s);
    DBHIP_LAUNCH_CHECK();
    DBHIP_CHECK(hipMemcpyAsync(host_ctrl, ctrl, 16, hipMemcpyDeviceToHost, s));
    if (chained) {
      int32_t rc = dbhip_groupby_merge_rows_dev_internal(g, partial, variant == 0 ? n_max4 : n_max8, &ctrl[0], &ctrl[1], s);
      if (rc) return rc;  // (synchronises the stream: host_ctrl is valid now)
      if (!(host_ctrl[1] & 3)) return DBHIP_OK;
    } else {
      DBHIP_CHECK(hipStreamSynchronize(s));
    }
    if (!(host_ctrl[1] & 1)) break;
  }
  if (host_ctrl[1] & 2) {
    set_error("dbhip_q1_fused: a group key string is longer than 12 bytes; use the operator-at-a-time path");
    return DBHIP_ERR_UNSUPPORTED;
  }
  if (host_ctrl[1] & 1) {
    set_error("dbhip_q1_fused: more than %d distinct groups inside one workgroup; use the operator-at-a-time path", MAX_SLOTS);
    return DBHIP_ERR_CAPACITY;
  }
  return dbhip_groupby_merge_rows_internal(g, partial, (int64_t)host_ctrl[0], s);
}

}  // extern "C"

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

#include <string.h>

using namespace dbhip;

int32_t dbhip_groupby_merge_rows_internal(dbhip_groupby* g, const uint64_t* rows, int64_t n, hipStream_t s);
const GbLayout* dbhip_groupby_layout_internal(dbhip_groupby* g);

namespace {

constexpr int SLOTS = 8;
constexpr uint64_t TAB_EMPTY = 0;
constexpr uint64_t TAB_LOCK = 1;
constexpr int Q1_W = 15;  // words per row of the Q1 table layout (see dbhip_q1_create_groupby)

struct Q1Acc {
  uint64_t qty, price, disc;
  u128 dp, ch;
  int32_t dp_ext, ch_ext;  // bits 128.. of the exact sums (per lane an i32 is ample)
  uint32_t cnt;
};

struct alignas(16) U4 {
  uint32_t x, y, z, w;
};
struct alignas(16) L2 {
  int64_t a, b;
};
struct alignas(8) I2 {
  int32_t a, b;
};

struct KeyTable {
  uint64_t hash[SLOTS];     // TAB_EMPTY / TAB_LOCK / remapped hash
  uint64_t real_hash[SLOTS];
  uint64_t key[SLOTS][4];
};

__device__ __forceinline__ uint64_t tab_word(uint64_t h) { return h <= TAB_LOCK ? h + 2 : h; }

// canonical words of an inline view (bytes past len zeroed), false if len > 12
__device__ __forceinline__ bool view_words(U4 v, uint64_t w[2]) {
  uint32_t len = v.x;
  if (len > 12) return false;
  uint32_t d1 = v.y, d2 = v.z, d3 = v.w;
  if (len < 4) { d1 &= (len == 0) ? 0u : (0xffffffffu >> (8 * (4 - len))); d2 = 0; d3 = 0; }
  else if (len < 8) { d2 &= (len == 4) ? 0u : (0xffffffffu >> (8 * (8 - len))); d3 = 0; }
  else if (len < 12) { d3 &= (len == 8) ? 0u : (0xffffffffu >> (8 * (12 - len))); }
  w[0] = ((uint64_t)d1 << 32) | len;
  w[1] = ((uint64_t)d3 << 32) | d2;
  return true;
}

__device__ __forceinline__ uint64_t hash_view_words(const uint64_t w[2]) {
  return agg_hash_inline_view((uint32_t)w[0], (uint32_t)(w[0] >> 32), (uint32_t)w[1], (uint32_t)(w[1] >> 32));
}

// fast path: read-only probe of the block's key table. Returns slot or -1.
__device__ __forceinline__ int tab_lookup(volatile KeyTable* T, uint64_t hq, const uint64_t k[4]) {
#pragma unroll
  for (int p = 0; p < SLOTS; ++p) {
    int s = (int)((hq + p) & (SLOTS - 1));
    uint64_t th = T->hash[s];
    if (th == hq && T->key[s][0] == k[0] && T->key[s][1] == k[1] && T->key[s][2] == k[2] &&
        T->key[s][3] == k[3])
      return s;
    if (th == TAB_EMPTY || th == TAB_LOCK) return -1;
  }
  return -1;
}

// slow path, executed by ONE lane of a wave at a time: find or insert.
// Returns slot, or -1 when the table is full.
__device__ int tab_insert(KeyTable* T, uint64_t hq, uint64_t h, const uint64_t k[4]) {
  for (int p = 0; p < SLOTS; ++p) {
    int s = (int)((hq + p) & (SLOTS - 1));
    while (true) {
      unsigned long long old = atomicCAS((unsigned long long*)&T->hash[s], (unsigned long long)TAB_EMPTY,
                                         (unsigned long long)TAB_LOCK);
      if (old == TAB_EMPTY) {
        T->key[s][0] = k[0]; T->key[s][1] = k[1]; T->key[s][2] = k[2]; T->key[s][3] = k[3];
        T->real_hash[s] = h;
        __threadfence_block();
        atomicExch((unsigned long long*)&T->hash[s], (unsigned long long)hq);
        return s;
      }
      if (old == TAB_LOCK) {  // another wave is publishing this slot
        __builtin_amdgcn_s_sleep(1);
        continue;
      }
      if (old == hq) {
        volatile KeyTable* V = T;
        if (V->key[s][0] == k[0] && V->key[s][1] == k[1] && V->key[s][2] == k[2] && V->key[s][3] == k[3])
          return s;
      }
      break;  // occupied by another key: next slot
    }
  }
  return -1;
}

// resolve the slot of one key row (wave-convergent call)
__device__ __forceinline__ int resolve_slot(KeyTable* T, bool row_valid, U4 v0, U4 v1, uint32_t* flags) {
  uint64_t k[4];
  bool ok0 = view_words(v0, k);
  bool ok1 = view_words(v1, k + 2);
  bool ok = ok0 && ok1;
  if (row_valid && !ok) *flags |= 2;
  bool want = row_valid && ok;
  uint64_t h = 0, hq = 0;
  int slot = -1;
  if (want) {
    h = merge_hash(hash_view_words(k), hash_view_words(k + 2));
    hq = tab_word(h);
    slot = tab_lookup((volatile KeyTable*)T, hq, k);
  }
  uint64_t miss = __ballot(want && slot < 0);
  while (miss) {
    int leader = __ffsll((long long)miss) - 1;
    uint64_t lk[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) lk[j] = __shfl(k[j], leader, 64);
    int ls = -1;
    if (lane_id() == leader) {
      ls = tab_insert(T, hq, h, k);
      if (ls < 0) *flags |= 1;
    }
    ls = __shfl(ls, leader, 64);
    bool same = want && slot < 0 && k[0] == lk[0] && k[1] == lk[1] && k[2] == lk[2] && k[3] == lk[3];
    if (same) slot = ls >= 0 ? ls : 0xE;  // 0xE: dropped (table full), flagged above
    miss &= ~__ballot(same);
  }
  return (slot < 0) ? 0xF : slot;  // 0xF: row not valid
}

__device__ __forceinline__ void acc_row(Q1Acc acc[SLOTS], int slot, bool pass, int64_t qty, int64_t price,
                                        int64_t disc, int64_t tax) {
  // decimal maps (see header): all plain wrapping integer ops
  int64_t one_minus_disc = (int64_t)(100ULL - (uint64_t)disc);
  int64_t one_plus_tax = (int64_t)(100ULL + (uint64_t)tax);
  i128 dp = (i128)price * (i128)one_minus_disc;
  i128 ch = (i128)((u128)dp * (u128)(i128)one_plus_tax);
#pragma unroll
  for (int g = 0; g < SLOTS; ++g) {
    bool m = pass && slot == g;
    acc[g].qty += m ? (uint64_t)qty : 0;
    acc[g].price += m ? (uint64_t)price : 0;
    acc[g].disc += m ? (uint64_t)disc : 0;
    u128 vdp = m ? (u128)dp : (u128)0, vch = m ? (u128)ch : (u128)0;
    u128 ndp = acc[g].dp + vdp, nch = acc[g].ch + vch;
    acc[g].dp_ext += (int32_t)(ndp < vdp) - (int32_t)(m && dp < 0);
    acc[g].ch_ext += (int32_t)(nch < vch) - (int32_t)(m && ch < 0);
    acc[g].dp = ndp;
    acc[g].ch = nch;
    acc[g].cnt += m ? 1u : 0u;
  }
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

__global__ __launch_bounds__(256, 2) void q1_fused_kernel(Q1Args A) {
  __shared__ KeyTable T;
  __shared__ uint64_t red[4][SLOTS][10];
  if (threadIdx.x < SLOTS) T.hash[threadIdx.x] = TAB_EMPTY;
  __syncthreads();

  Q1Acc acc[SLOTS];
#pragma unroll
  for (int g = 0; g < SLOTS; ++g) {
    acc[g].qty = 0; acc[g].price = 0; acc[g].disc = 0; acc[g].dp = 0; acc[g].ch = 0; acc[g].cnt = 0;
    acc[g].dp_ext = 0; acc[g].ch_ext = 0;
  }
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
      rfA = A.rf[ra]; rfB = A.rf[rb];
      lsA = A.ls[ra]; lsB = A.ls[rb];
      q = *(const L2*)(A.qty + r0);
      p = *(const L2*)(A.price + r0);
      d = *(const L2*)(A.disc + r0);
      x = *(const L2*)(A.tax + r0);
      sd = *(const I2*)(A.shipdate + r0);
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
    // ---- keys -> slots (key layout) ----
    int slotA = resolve_slot(&T, ra < A.n, rfA, lsA, &flags);
    int slotB = resolve_slot(&T, rb < A.n, rfB, lsB, &flags);
    // ---- move slot ids to the value layout ----
    int packed = slotA | (slotB << 8);
    int g0 = __shfl(packed, (2 * lane) & 63, 64);
    int g1 = __shfl(packed, (2 * lane + 1) & 63, 64);
    int s0 = (lane < 32 ? g0 : (g0 >> 8)) & 0xFF;
    int s1 = (lane < 32 ? g1 : (g1 >> 8)) & 0xFF;
    // ---- filter + maps + accumulate ----
    bool pass0 = r0 < A.n && sd.a <= A.cutoff;
    bool pass1 = r1 < A.n && sd.b <= A.cutoff;
    acc_row(acc, s0, pass0, q.a, p.a, d.a, x.a);
    acc_row(acc, s1, pass1, q.b, p.b, d.b, x.b);
  }

  // ---- wave reduce, then block combine ----
  const int wave = threadIdx.x >> 6;
#pragma unroll
  for (int g = 0; g < SLOTS; ++g) {
    uint64_t a0 = wave_sum_u64(acc[g].qty);
    uint64_t a1 = wave_sum_u64(acc[g].price);
    uint64_t a2 = wave_sum_u64(acc[g].disc);
    uint64_t e3 = (uint64_t)(int64_t)acc[g].dp_ext, e4 = (uint64_t)(int64_t)acc[g].ch_ext;
    u128 a3 = wave_sum_u192(acc[g].dp, &e3);
    u128 a4 = wave_sum_u192(acc[g].ch, &e4);
    uint64_t a5 = wave_sum_u64((uint64_t)acc[g].cnt);
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
  if (threadIdx.x < SLOTS) {
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
    if (T.hash[g] > TAB_LOCK && cnt != 0) {
      unsigned long long idx = atomicAdd((unsigned long long*)&A.ctrl[0], 1ULL);
      uint64_t* r = A.partial_rows + idx * Q1_W;
      // table layout: [rf view 2w][ls view 2w][hash][sum_qty][sum_price][sum_dp 3w][sum_ch 3w][sum_disc][count]
      r[0] = T.key[g][0]; r[1] = T.key[g][1]; r[2] = T.key[g][2]; r[3] = T.key[g][3];
      r[4] = T.real_hash[g];
      r[5] = qty; r[6] = price;
      r[7] = (uint64_t)dp; r[8] = (uint64_t)(dp >> 64); r[9] = dpe;
      r[10] = (uint64_t)ch; r[11] = (uint64_t)(ch >> 64); r[12] = che;
      r[13] = disc; r[14] = cnt;
    }
  }
}

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
  return dbhip_groupby_create(key_types, key_nullable, 2, aggs, 6, 1024, out_host);
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
  int grid = (int)(ceil_div(ntiles, 4) < 2048 ? ceil_div(ntiles, 4) : 2048);
  size_t rows_bytes = (size_t)grid * SLOTS * Q1_W * 8;
  uint8_t* ws = (uint8_t*)scratch(rows_bytes + 64, 4);
  if (!ws) return DBHIP_ERR_HIP;
  uint64_t* ctrl = (uint64_t*)ws;
  uint64_t* partial = (uint64_t*)(ws + 64);
  DBHIP_CHECK(hipMemsetAsync(ctrl, 0, 64, s));
  Q1Args A;
  A.qty = l_quantity; A.price = l_extendedprice; A.disc = l_discount; A.tax = l_tax;
  A.rf = (const U4*)l_returnflag_views; A.ls = (const U4*)l_linestatus_views;
  A.shipdate = l_shipdate; A.cutoff = shipdate_cutoff; A.n = n;
  A.partial_rows = partial; A.ctrl = ctrl;
  kernel_timer_start(s);
  hipLaunchKernelGGL(q1_fused_kernel, dim3(grid), dim3(256), 0, s, A);
  kernel_timer_stop(s);
  DBHIP_LAUNCH_CHECK();
  uint64_t host_ctrl[2];
  DBHIP_CHECK(hipMemcpyAsync(host_ctrl, ctrl, 16, hipMemcpyDeviceToHost, s));
  DBHIP_CHECK(hipStreamSynchronize(s));
  if (host_ctrl[1] & 2) {
    set_error("dbhip_q1_fused: a group key string is longer than 12 bytes; use the operator-at-a-time path");
    return DBHIP_ERR_UNSUPPORTED;
  }
  if (host_ctrl[1] & 1) {
    set_error("dbhip_q1_fused: more than %d distinct groups inside one workgroup; use the operator-at-a-time path", SLOTS);
    return DBHIP_ERR_CAPACITY;
  }
  return dbhip_groupby_merge_rows_internal(g, partial, (int64_t)host_ctrl[0], s);
}

}  // extern "C"

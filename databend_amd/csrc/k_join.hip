// k_join.hip — inner hash join on KeysU64.
//
// Reference: HashJoinHashTable<u64> (hash_join_table/hashjoin_hashtable.rs:26-137): bucket array of
// entry pointers, capacity max(2*rows -> pow2, 1024) (:95-108), idx = hash >> (64 - log2 cap),
// lock-free CAS prepend (:110-137); probe walks the chain comparing keys
// (new_hash_join/hashtable/fixed_keys.rs:209-269) and emits (probe_idx, RowPtr).
// Device geometry: head[cap] (u32 build row + 1, 0 = empty) + next[rows] chains; insertion is one
// atomicExch per build row (prepend), chains are only walked by later launches. NULL keys never
// match (validity bit 0 rows are neither inserted nor probed). The join hash itself is not part
// of the parity contract (FastHash is CRC32-C or a murmur mix depending on the host CPU,
// common/hashtable/src/traits.rs:199-211): a 64-bit multiply-xorshift is used.
// Pair order across threads is unspecified in the reference; here pairs come out sorted by
// (probe_idx, build_row): count per probe row -> exclusive scan -> ordered emit.
#include "dev_common.h"
#include "dev_scan.h"
#include "runtime.h"

#include <string.h>

#include <new>

using namespace dbhip;

struct dbhip_join {
  uint64_t* keys;      // all build keys in arrival order
  uint8_t* valid;      // one byte per build row
  int64_t nrows, cap_rows;
  uint32_t* head;      // [buckets]
  uint32_t* next;      // [nrows]
  int64_t buckets;
  int shift;
  bool finalized;
  // probe scratch
  uint32_t* cnt; uint64_t* off; uint64_t* blk; size_t scratch_rows;
  uint64_t* total_dev;
};

namespace {

__device__ __forceinline__ uint64_t join_hash(uint64_t x) { return agg_hash_u64(x); }

__global__ __launch_bounds__(256) void join_copy_kernel(const uint64_t* keys, const uint8_t* validity, int64_t n,
                                                        uint64_t* dst_keys, uint8_t* dst_valid) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    dst_keys[i] = keys[i];
    dst_valid[i] = validity ? (uint8_t)bit_get(validity, i) : 1;
  }
}

__global__ __launch_bounds__(256) void join_build_kernel(const uint64_t* keys, const uint8_t* valid, int64_t n,
                                                         uint32_t* head, uint32_t* next, int shift) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    if (!valid[i]) { next[i] = 0; continue; }
    uint64_t idx = join_hash(keys[i]) >> shift;
    next[i] = atomicExch(&head[idx], (uint32_t)(i + 1));  // prepend
  }
}

__global__ __launch_bounds__(256) void join_count_kernel(const uint64_t* bkeys, const uint32_t* head,
                                                         const uint32_t* next, int shift, const uint64_t* pkeys,
                                                         const uint8_t* pvalid, int64_t n, uint32_t* cnt,
                                                         unsigned long long* total) {
  uint64_t local = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    uint32_t c = 0;
    if (!pvalid || bit_get(pvalid, i)) {
      uint64_t k = pkeys[i];
      for (uint32_t e = head[join_hash(k) >> shift]; e; e = next[e - 1]) c += (bkeys[e - 1] == k);
    }
    if (cnt) cnt[i] = c;
    local += c;
  }
  local = wave_sum_u64(local);
  if (lane_id() == 0 && local) atomicAdd(total, (unsigned long long)local);
}

__global__ __launch_bounds__(256) void join_emit_kernel(const uint64_t* bkeys, const uint32_t* head,
                                                        const uint32_t* next, int shift, const uint64_t* pkeys,
                                                        const uint8_t* pvalid, int64_t n, const uint32_t* cnt,
                                                        const uint64_t* off, uint32_t* out_p, uint32_t* out_b,
                                                        int64_t max_pairs) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    if (cnt[i] == 0) continue;
    uint64_t k = pkeys[i], o = off[i];
    if ((int64_t)(o + cnt[i]) > max_pairs) continue;
    uint32_t m = 0;
    for (uint32_t e = head[join_hash(k) >> shift]; e; e = next[e - 1]) {
      if (bkeys[e - 1] != k) continue;
      // insert build row e-1 keeping this probe row's segment ascending (segments are tiny)
      uint32_t b = e - 1, j = m;
      while (j > 0 && out_b[o + j - 1] > b) { out_b[o + j] = out_b[o + j - 1]; --j; }
      out_b[o + j] = b;
      out_p[o + m] = (uint32_t)i;
      ++m;
    }
  }
}

int32_t ensure_probe_scratch(dbhip_join* j, int64_t n) {
  if (j->scratch_rows >= (size_t)n) return DBHIP_OK;
  if (j->cnt) { DBHIP_CHECK(hipDeviceSynchronize()); (void)hipFree(j->cnt); (void)hipFree(j->off); (void)hipFree(j->blk); }
  size_t cap = (size_t)n + (n >> 3) + 1024;
  DBHIP_CHECK(hipMalloc((void**)&j->cnt, cap * 4));
  DBHIP_CHECK(hipMalloc((void**)&j->off, cap * 8));
  DBHIP_CHECK(hipMalloc((void**)&j->blk, (cap / SCAN_TILE + 2) * 8));
  j->scratch_rows = cap;
  return DBHIP_OK;
}

}  // namespace

extern "C" {

int32_t dbhip_join_create(int64_t expected_build_rows, dbhip_join** out_host) {
  DBHIP_REQUIRE(out_host, "dbhip_join_create: NULL out");
  dbhip_join* j = new (std::nothrow) dbhip_join();
  DBHIP_REQUIRE(j, "dbhip_join_create: out of host memory");
  memset(j, 0, sizeof(*j));
  j->cap_rows = expected_build_rows > 1024 ? expected_build_rows : 1024;
  DBHIP_CHECK(hipMalloc((void**)&j->keys, (size_t)j->cap_rows * 8));
  DBHIP_CHECK(hipMalloc((void**)&j->valid, (size_t)j->cap_rows));
  DBHIP_CHECK(hipMalloc((void**)&j->total_dev, 8));
  *out_host = j;
  return DBHIP_OK;
}

int32_t dbhip_join_add_build(dbhip_join* j, const uint64_t* keys, const uint8_t* validity, int64_t n,
                             void* stream) {
  DBHIP_REQUIRE(j && !j->finalized, "dbhip_join_add_build: table missing or already finalized");
  if (n == 0) return DBHIP_OK;
  DBHIP_REQUIRE(keys, "dbhip_join_add_build: NULL keys");
  DBHIP_REQUIRE(j->nrows + n < 0xFFFFFFFFLL, "dbhip_join_add_build: more than 2^32-1 build rows");
  hipStream_t s = resolve_stream(stream);
  if (j->nrows + n > j->cap_rows) {  // grow the chunk store (BasicHashJoin::add_block squashes chunks)
    int64_t nc = j->cap_rows * 2 > j->nrows + n ? j->cap_rows * 2 : j->nrows + n;
    uint64_t* nk; uint8_t* nv;
    DBHIP_CHECK(hipMalloc((void**)&nk, (size_t)nc * 8));
    DBHIP_CHECK(hipMalloc((void**)&nv, (size_t)nc));
    DBHIP_CHECK(hipMemcpyAsync(nk, j->keys, (size_t)j->nrows * 8, hipMemcpyDeviceToDevice, s));
    DBHIP_CHECK(hipMemcpyAsync(nv, j->valid, (size_t)j->nrows, hipMemcpyDeviceToDevice, s));
    DBHIP_CHECK(hipStreamSynchronize(s));
    (void)hipFree(j->keys); (void)hipFree(j->valid);
    j->keys = nk; j->valid = nv; j->cap_rows = nc;
  }
  hipLaunchKernelGGL(join_copy_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, keys, validity, n,
                     j->keys + j->nrows, j->valid + j->nrows);
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
  DBHIP_CHECK(hipMalloc((void**)&j->head, (size_t)cap * 4));
  DBHIP_CHECK(hipMalloc((void**)&j->next, (size_t)(j->nrows ? j->nrows : 1) * 4));
  DBHIP_CHECK(hipMemsetAsync(j->head, 0, (size_t)cap * 4, s));
  if (j->nrows) {
    hipLaunchKernelGGL(join_build_kernel, dim3(grid_for(j->nrows, 256)), dim3(256), 0, s, j->keys, j->valid,
                       j->nrows, j->head, j->next, j->shift);
    DBHIP_LAUNCH_CHECK();
  }
  j->finalized = true;
  return DBHIP_OK;
}

int32_t dbhip_join_probe_count(dbhip_join* j, const uint64_t* keys, const uint8_t* validity, int64_t n,
                               uint64_t* out_total_host, void* stream) {
  DBHIP_REQUIRE(j && j->finalized && out_total_host, "dbhip_join_probe_count: table not finalized / NULL out");
  hipStream_t s = resolve_stream(stream);
  DBHIP_CHECK(hipMemsetAsync(j->total_dev, 0, 8, s));
  if (n) {
    hipLaunchKernelGGL(join_count_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, j->keys, j->head, j->next,
                       j->shift, keys, validity, n, (uint32_t*)nullptr, (unsigned long long*)j->total_dev);
    DBHIP_LAUNCH_CHECK();
  }
  DBHIP_CHECK(hipMemcpyAsync(out_total_host, j->total_dev, 8, hipMemcpyDeviceToHost, s));
  DBHIP_CHECK(hipStreamSynchronize(s));
  return DBHIP_OK;
}

int32_t dbhip_join_probe(dbhip_join* j, const uint64_t* keys, const uint8_t* validity, int64_t n,
                         uint32_t* out_probe_idx, uint32_t* out_build_row, int64_t max_pairs,
                         uint64_t* out_n_pairs_host, void* stream) {
  DBHIP_REQUIRE(j && j->finalized && out_n_pairs_host, "dbhip_join_probe: table not finalized / NULL out");
  DBHIP_REQUIRE(n < 0xFFFFFFFFLL, "dbhip_join_probe: more than 2^32-1 probe rows in one block");
  hipStream_t s = resolve_stream(stream);
  *out_n_pairs_host = 0;
  if (n == 0) return DBHIP_OK;
  int32_t rc = ensure_probe_scratch(j, n);
  if (rc) return rc;
  DBHIP_CHECK(hipMemsetAsync(j->total_dev, 0, 8, s));
  const int grid = grid_for(n, 256);
  hipLaunchKernelGGL(join_count_kernel, dim3(grid), dim3(256), 0, s, j->keys, j->head, j->next, j->shift, keys,
                     validity, n, j->cnt, (unsigned long long*)j->total_dev);
  rc = dbscan::exclusive_scan_u32(j->cnt, n, j->blk, j->off, s);
  if (rc) return rc;
  uint64_t total = 0;
  DBHIP_CHECK(hipMemcpyAsync(&total, j->total_dev, 8, hipMemcpyDeviceToHost, s));
  DBHIP_CHECK(hipStreamSynchronize(s));
  *out_n_pairs_host = total;
  if ((int64_t)total > max_pairs) {
    set_error("dbhip_join_probe: %llu pairs do not fit max_pairs=%lld (call dbhip_join_probe_count first)",
              (unsigned long long)total, (long long)max_pairs);
    return DBHIP_ERR_CAPACITY;
  }
  if (total) {
    DBHIP_REQUIRE(out_probe_idx && out_build_row, "dbhip_join_probe: NULL output");
    hipLaunchKernelGGL(join_emit_kernel, dim3(grid), dim3(256), 0, s, j->keys, j->head, j->next, j->shift, keys,
                       validity, n, j->cnt, j->off, out_probe_idx, out_build_row, max_pairs);
    DBHIP_LAUNCH_CHECK();
  }
  return DBHIP_OK;
}

int32_t dbhip_join_destroy(dbhip_join* j) {
  if (!j) return DBHIP_OK;
  (void)hipDeviceSynchronize();
  void* ptrs[] = {j->keys, j->valid, j->head, j->next, j->cnt, j->off, j->blk, j->total_dev};
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
  delete j;
  return DBHIP_OK;
}

}  // extern "C"

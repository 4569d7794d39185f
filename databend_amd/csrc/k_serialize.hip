// k_serialize.hip — HashMethodSerializer (the "else" branch of DataBlock::choose_hash_method_with_types,
// src/query/expression/src/kernels/group_by.rs:40-80): key columns that are not all fixed-width numbers, or whose packed width
// exceeds 32 bytes, are serialized row by row into ONE BinaryColumn (group_by_hash/method_serializer.rs:33-52,
// group_by_hash/utils.rs:33-160 serialize_group_columns / serialize_column_binary) and the join hashes and compares those bytes.
//
//   per row, column after column:  number / decimal / date / timestamp : the value's little-endian bytes (1..32)
//                                  boolean                            : one byte 0 / 1
//                                  string                             : u64 length, then the bytes
//                                  nullable column                    : one byte `valid`, then the value only when valid
//
// Device plan: sizes (one thread per row) -> exclusive scan = the BinaryColumn's offsets -> write (one thread per row; rows are
// tens of bytes). The join on such keys (dbhip_join_*_binary) reduces to the fixed-key table: a 128-bit hash of the row's bytes
// is the KeysU128 key of the existing machinery (k_join.hip), and every emitted pair is then VERIFIED byte for byte against the
// build side's rows, which the table keeps — equal bytes are what the reference's table compares, the hash only routes.
#include "dev_common.h"
#include "dev_scan.h"
#include "runtime.h"

#include <new>
#include <vector>

using namespace dbhip;

namespace {

constexpr int SER_MAX_COLS = 16;

struct SerCol {
  const void* data;
  const uint8_t* validity;
  int64_t voff;
  const void* const* buffers;
  int32_t type, is_scalar, fixed;  // fixed = byte width of a fixed-width value, 0 = string
};
struct SerArgs {
  SerCol col[SER_MAX_COLS];
  int32_t ncols;
  int64_t n;
};

__device__ __forceinline__ const uint8_t* view_bytes(const SerCol& c, int64_t j, uint32_t* len) {
  const uint32_t* v = (const uint32_t*)c.data + 4 * j;
  *len = v[0];
  if (*len <= 12) return (const uint8_t*)(v + 1);
  return (const uint8_t*)c.buffers[v[2]] + v[3];
}

__global__ __launch_bounds__(256) void ser_size_kernel(SerArgs A, uint32_t* sizes, uint8_t* all_valid) {
  const int64_t n_pad = (A.n + 63) & ~63LL;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_pad; i += (int64_t)gridDim.x * blockDim.x) {
    bool allv = true;
    if (i < A.n) {
      uint32_t sz = 0;
      for (int c = 0; c < A.ncols; ++c) {
        const SerCol& C = A.col[c];
        const int64_t j = C.is_scalar ? 0 : i;
        bool valid = true;
        if (C.validity) { valid = bit_get(C.validity, C.voff + j); sz += 1; allv &= valid; }
        if (valid) sz += C.fixed ? (uint32_t)C.fixed : 8u + ((const uint32_t*)C.data)[4 * j];
      }
      sizes[i] = sz;
    }
    if (all_valid) {
      const uint64_t m = __ballot(allv && i < A.n);
      if ((lane_id() & 7) == 0 && (i >> 3) < ((A.n + 7) >> 3)) all_valid[i >> 3] = (uint8_t)(m >> lane_id());
    }
  }
}

__global__ __launch_bounds__(256) void ser_write_kernel(SerArgs A, const uint64_t* off, uint8_t* out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < A.n; i += (int64_t)gridDim.x * blockDim.x) {
    uint8_t* p = out + off[i];
    for (int c = 0; c < A.ncols; ++c) {
      const SerCol& C = A.col[c];
      const int64_t j = C.is_scalar ? 0 : i;
      if (C.validity) {
        const bool valid = bit_get(C.validity, C.voff + j);
        *p++ = valid ? 1 : 0;
        if (!valid) continue;
      }
      if (C.type == DBHIP_T_BOOL) {
        *p++ = bit_get((const uint8_t*)C.data, j) ? 1 : 0;
      } else if (C.fixed) {
        const uint8_t* s = (const uint8_t*)C.data + (size_t)j * C.fixed;
        for (int b = 0; b < C.fixed; ++b) p[b] = s[b];
        p += C.fixed;
      } else {
        uint32_t len;
        const uint8_t* s = view_bytes(C, j, &len);
        const uint64_t l64 = len;
        for (int b = 0; b < 8; ++b) p[b] = (uint8_t)(l64 >> (8 * b));
        p += 8;
        for (uint32_t b = 0; b < len; ++b) p[b] = s[b];
        p += len;
      }
    }
  }
}

// Two independent 64-bit hashes of a row's bytes (Murmur64A-style rounds over 8-byte words with different seeds and
// multipliers, tail bytes folded in) = the 128-bit routing key of the fixed-key join table. Equality of keys is decided on the
// bytes (verify kernel), never on this value.
__device__ __forceinline__ void hash128_bytes(const uint8_t* p, uint64_t len, uint64_t* h0, uint64_t* h1) {
  const uint64_t M0 = 0xc6a4a7935bd1e995ULL, M1 = 0x9e3779b97f4a7c15ULL;
  uint64_t a = 0xe17a1465ULL ^ (len * M0), b = 0x8445d61a4e774912ULL ^ (len * M1);
  uint64_t i = 0;
  for (; i + 8 <= len; i += 8) {
    uint64_t k = 0;
    for (int t = 0; t < 8; ++t) k |= (uint64_t)p[i + t] << (8 * t);
    uint64_t ka = k * M0; ka ^= ka >> 47; ka *= M0; a ^= ka; a *= M0;
    uint64_t kb = k * M1; kb ^= kb >> 29; kb *= M1; b = (b ^ kb) * M1 + 0x632be59bd9b4e019ULL;
  }
  uint64_t k = 0;
  for (uint64_t t = 0; i + t < len; ++t) k |= (uint64_t)p[i + t] << (8 * t);
  a ^= k; a *= M0; b ^= k * M1; b *= M1;
  a ^= a >> 47; a *= M0; a ^= a >> 47;
  b ^= b >> 32; b *= M1; b ^= b >> 29;
  *h0 = a; *h1 = b;
}
__global__ __launch_bounds__(256) void hash128_rows_kernel(const uint64_t* off, const uint8_t* data, int64_t n, uint64_t mask, uint64_t* out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    uint64_t h0, h1;
    hash128_bytes(data + off[i], off[i + 1] - off[i], &h0, &h1);
    out[2 * i] = h0 & mask; out[2 * i + 1] = h1 & mask;
  }
}

// keep[i] = the probe row's bytes equal the build row's bytes
__global__ __launch_bounds__(256) void verify_pairs_kernel(const uint32_t* pi, const uint32_t* bi, int64_t np, const uint64_t* poff,
                                                           const uint8_t* pdata, const uint64_t* boff, const uint8_t* bdata, uint32_t* keep) {
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < np; t += (int64_t)gridDim.x * blockDim.x) {
    const uint64_t po = poff[pi[t]], pl = poff[pi[t] + 1] - po, bo = boff[bi[t]], bl = boff[bi[t] + 1] - bo;
    bool eq = pl == bl;
    for (uint64_t b = 0; eq && b < pl; ++b) eq = pdata[po + b] == bdata[bo + b];
    keep[t] = eq ? 1u : 0u;
  }
}
__global__ __launch_bounds__(256) void compact_pairs_kernel(const uint32_t* pi, const uint32_t* bi, int64_t np, const uint32_t* keep,
                                                            const uint64_t* pos, uint32_t* out_pi, uint32_t* out_bi, uint8_t* matched) {
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < np; t += (int64_t)gridDim.x * blockDim.x) {
    if (!keep[t]) continue;
    out_pi[pos[t]] = pi[t];
    out_bi[pos[t]] = bi[t];
    if (matched) atomicOr((uint32_t*)(matched + ((pi[t] >> 5) << 2)), 1u << (pi[t] & 31));
  }
}
__global__ __launch_bounds__(256) void rebase_offsets_kernel(const uint64_t* src, int64_t n_plus_1, uint64_t delta, uint64_t* dst) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_plus_1; i += (int64_t)gridDim.x * blockDim.x) dst[i] = src[i] + delta;
}

int fixed_width(int type) {
  switch (type) {
    case DBHIP_T_BOOL: return 1;
    case DBHIP_T_STRING: return 0;
    default: return type_size(type);
  }
}

int32_t make_args(const dbhip_col* cols, int32_t ncols, int64_t n, SerArgs* A) {
  DBHIP_REQUIRE(cols && ncols >= 1 && ncols <= SER_MAX_COLS, "dbhip_serialize_keys: 1..16 key columns");
  A->ncols = ncols; A->n = n;
  for (int c = 0; c < ncols; ++c) {
    SerCol& S = A->col[c];
    S.data = cols[c].data; S.validity = cols[c].validity; S.voff = cols[c].validity_offset; S.buffers = cols[c].buffers;
    S.type = cols[c].type; S.is_scalar = cols[c].is_scalar; S.fixed = fixed_width(cols[c].type);
    if (S.type != DBHIP_T_STRING && S.fixed == 0) { set_error("dbhip_serialize_keys: column %d has unsupported type %d", c, S.type); return DBHIP_ERR_UNSUPPORTED; }
    DBHIP_REQUIRE(S.data || n == 0, "dbhip_serialize_keys: NULL column data");
  }
  return DBHIP_OK;
}

}  // namespace

struct dbhip_join_binary {
  dbhip_join* inner = nullptr;          // KeysU128 table over the rows' 128-bit hashes
  uint64_t* off = nullptr;              // build rows: offsets[n_rows + 1] into `data`
  uint8_t* data = nullptr;
  size_t off_cap = 0, data_cap = 0;     // in rows / bytes
  int64_t n_rows = 0;
  uint64_t n_bytes = 0;
};

namespace {

// test hook (like the group-by's hash mask, hash_index/index.rs:385-404): AND-ed into both hash words, so that different keys
// share a routing key and the byte-for-byte verification has something to reject
uint64_t g_hash_mask = ~0ULL;

int32_t grow(void** p, size_t old_bytes, size_t new_bytes, hipStream_t s) {
  void* q = nullptr;
  int32_t rc = dbhip_alloc(new_bytes, &q);
  if (rc) return rc;
  if (*p && old_bytes) DBHIP_CHECK(hipMemcpyAsync(q, *p, old_bytes, hipMemcpyDeviceToDevice, s));
  if (*p) { DBHIP_CHECK(hipStreamSynchronize(s)); (void)dbhip_free(*p); }
  *p = q;
  return DBHIP_OK;
}

// hashes of the n rows (offsets[n + 1], data) into scratch slot 12: u64[2 n]
int32_t hash_rows(const uint64_t* off, const uint8_t* data, int64_t n, uint64_t** out, hipStream_t s) {
  uint64_t* h = (uint64_t*)scratch((size_t)(n > 0 ? n : 1) * 16, 12, s);
  if (!h) return DBHIP_ERR_HIP;
  if (n) hipLaunchKernelGGL(hash128_rows_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, off, data, n, g_hash_mask, h);
  DBHIP_LAUNCH_CHECK();
  *out = h;
  return DBHIP_OK;
}

}  // namespace

extern "C" int32_t dbhip_join_binary_debug_set_hash_mask(uint64_t mask) { g_hash_mask = mask; return DBHIP_OK; }

extern "C" {

int32_t dbhip_serialize_keys_offsets(const dbhip_col* cols, int32_t ncols, int64_t n, uint64_t* out_offsets, uint8_t* out_all_valid,
                                     uint64_t* out_total_bytes_host, void* stream) {
  DBHIP_REQUIRE(out_offsets && out_total_bytes_host, "dbhip_serialize_keys_offsets: NULL argument");
  SerArgs A;
  int32_t rc = make_args(cols, ncols, n, &A);
  if (rc) return rc;
  hipStream_t s = resolve_stream(stream);
  // sizes of rows 0..n-1 and a trailing 0, so that the exclusive scan yields offsets[0..n]
  uint32_t* sizes = (uint32_t*)scratch((size_t)(n + 1) * 4 + (size_t)(ceil_div(n + 1, SCAN_TILE) + 2) * 8 + 64, 13, s);
  if (!sizes) return DBHIP_ERR_HIP;
  uint64_t* blk = (uint64_t*)((uint8_t*)sizes + (((size_t)(n + 1) * 4 + 15) & ~(size_t)15));
  DBHIP_CHECK(hipMemsetAsync(sizes + n, 0, 4, s));
  if (n) hipLaunchKernelGGL(ser_size_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, A, sizes, out_all_valid);
  DBHIP_LAUNCH_CHECK();
  if ((rc = dbscan::exclusive_scan_u32(sizes, n + 1, blk, out_offsets, s))) return rc;
  DBHIP_CHECK(hipMemcpyAsync(out_total_bytes_host, out_offsets + n, 8, hipMemcpyDeviceToHost, s));
  DBHIP_CHECK(hipStreamSynchronize(s));
  return DBHIP_OK;
}

int32_t dbhip_serialize_keys(const dbhip_col* cols, int32_t ncols, int64_t n, const uint64_t* offsets, uint8_t* out_data, void* stream) {
  DBHIP_REQUIRE(offsets && (out_data || n == 0), "dbhip_serialize_keys: NULL argument");
  SerArgs A;
  int32_t rc = make_args(cols, ncols, n, &A);
  if (rc) return rc;
  if (n == 0) return DBHIP_OK;
  hipLaunchKernelGGL(ser_write_kernel, dim3(grid_for(n, 256)), dim3(256), 0, resolve_stream(stream), A, offsets, out_data);
  DBHIP_LAUNCH_CHECK();
  return DBHIP_OK;
}

int32_t dbhip_join_create_binary(int64_t expected_build_rows, dbhip_join_binary** out_host) {
  DBHIP_REQUIRE(out_host, "dbhip_join_create_binary: NULL argument");
  dbhip_join_binary* j = new (std::nothrow) dbhip_join_binary();
  if (!j) return DBHIP_ERR_HIP;
  int32_t rc = dbhip_join_create_keys(expected_build_rows, 16, &j->inner);
  if (rc) { delete j; return rc; }
  *out_host = j;
  return DBHIP_OK;
}

int32_t dbhip_join_add_build_binary(dbhip_join_binary* j, const uint64_t* offsets, const uint8_t* data, const uint8_t* validity, int64_t n,
                                    void* stream) {
  DBHIP_REQUIRE(j && (offsets || n == 0), "dbhip_join_add_build_binary: NULL argument");
  if (n == 0) return DBHIP_OK;
  hipStream_t s = resolve_stream(stream);
  uint64_t first = 0, last = 0;
  DBHIP_CHECK(hipMemcpyAsync(&first, offsets, 8, hipMemcpyDeviceToHost, s));
  DBHIP_CHECK(hipMemcpyAsync(&last, offsets + n, 8, hipMemcpyDeviceToHost, s));
  DBHIP_CHECK(hipStreamSynchronize(s));
  const uint64_t nbytes = last - first;
  int32_t rc;
  // the table keeps the build rows: pairs are verified against them
  if ((size_t)(j->n_rows + n + 1) > j->off_cap) {
    const size_t want = (size_t)(j->n_rows + n + 1) * 2;
    if ((rc = grow((void**)&j->off, (size_t)(j->n_rows + (j->off ? 1 : 0)) * 8, want * 8, s))) return rc;
    j->off_cap = want;
  }
  if (j->n_bytes + nbytes > j->data_cap) {
    const size_t want = (size_t)(j->n_bytes + nbytes) * 2 + 64;
    if ((rc = grow((void**)&j->data, (size_t)j->n_bytes, want, s))) return rc;
    j->data_cap = want;
  }
  hipLaunchKernelGGL(rebase_offsets_kernel, dim3(grid_for(n + 1, 256)), dim3(256), 0, s, offsets, n + 1, j->n_bytes - first, j->off + j->n_rows);
  if (nbytes) DBHIP_CHECK(hipMemcpyAsync(j->data + j->n_bytes, data + first, (size_t)nbytes, hipMemcpyDeviceToDevice, s));
  uint64_t* h;
  if ((rc = hash_rows(j->off + j->n_rows, j->data, n, &h, s))) return rc;
  if ((rc = dbhip_join_add_build(j->inner, h, validity, n, stream))) return rc;
  j->n_rows += n; j->n_bytes += nbytes;
  return DBHIP_OK;
}

int32_t dbhip_join_finalize_binary(dbhip_join_binary* j, void* stream) {
  DBHIP_REQUIRE(j, "dbhip_join_finalize_binary: NULL argument");
  return dbhip_join_finalize(j->inner, stream);
}

int32_t dbhip_join_probe_count_binary(dbhip_join_binary* j, const uint64_t* offsets, const uint8_t* data, const uint8_t* validity, int64_t n,
                                      uint64_t* out_max_pairs_host, void* stream) {
  DBHIP_REQUIRE(j && out_max_pairs_host && (offsets || n == 0), "dbhip_join_probe_count_binary: NULL argument");
  *out_max_pairs_host = 0;
  if (n == 0) return DBHIP_OK;
  uint64_t* h;
  int32_t rc = hash_rows(offsets, data, n, &h, resolve_stream(stream));
  if (rc) return rc;
  return dbhip_join_probe_count(j->inner, h, validity, n, out_max_pairs_host, stream);
}

int32_t dbhip_join_probe_binary(dbhip_join_binary* j, const uint64_t* offsets, const uint8_t* data, const uint8_t* validity, int64_t n,
                                uint32_t* out_probe_idx, uint32_t* out_build_row, int64_t max_pairs, uint64_t* out_n_pairs_host,
                                uint8_t* out_matched_bitmap, void* stream) {
  DBHIP_REQUIRE(j && out_n_pairs_host && (offsets || n == 0), "dbhip_join_probe_binary: NULL argument");
  *out_n_pairs_host = 0;
  hipStream_t s = resolve_stream(stream);
  if (out_matched_bitmap) DBHIP_CHECK(hipMemsetAsync(out_matched_bitmap, 0, (size_t)ceil_div(n, 32) * 4, s));
  if (n == 0) return DBHIP_OK;
  uint64_t* h;
  int32_t rc = hash_rows(offsets, data, n, &h, s);
  if (rc) return rc;
  uint64_t cand = 0;
  if ((rc = dbhip_join_probe_count(j->inner, h, validity, n, &cand, stream))) return rc;
  if (cand == 0) return DBHIP_OK;
  if ((int64_t)cand > max_pairs) {
    set_error("dbhip_join_probe_binary: %llu candidate pairs do not fit max_pairs=%lld (size with dbhip_join_probe_count_binary)", (unsigned long long)cand, (long long)max_pairs);
    return DBHIP_ERR_CAPACITY;
  }
  // candidates by hash -> scratch, then verify the bytes and compact (order preserved: sorted by probe row, build row)
  const size_t np = (size_t)cand;
  uint8_t* ws = (uint8_t*)scratch(np * 4 * 3 + (np + 1) * 8 + (size_t)(ceil_div((int64_t)np, SCAN_TILE) + 2) * 8 + 256, 14, s);
  if (!ws) return DBHIP_ERR_HIP;
  uint32_t* cpi = (uint32_t*)ws;
  uint32_t* cbi = cpi + np;
  uint32_t* keep = cbi + np;
  uint64_t* pos = (uint64_t*)(((uintptr_t)(keep + np) + 15) & ~(uintptr_t)15);
  uint64_t* blk = pos + np + 1;
  uint64_t got = 0;
  if ((rc = dbhip_join_probe(j->inner, h, validity, n, cpi, cbi, (int64_t)np, &got, stream))) return rc;
  hipLaunchKernelGGL(verify_pairs_kernel, dim3(grid_for((int64_t)got, 256)), dim3(256), 0, s, cpi, cbi, (int64_t)got, offsets, data, j->off, j->data, keep);
  DBHIP_LAUNCH_CHECK();
  if ((rc = dbscan::exclusive_scan_u32(keep, (int64_t)got, blk, pos, s))) return rc;
  uint64_t last_pos = 0;
  uint32_t last_keep = 0;
  DBHIP_CHECK(hipMemcpyAsync(&last_pos, pos + got - 1, 8, hipMemcpyDeviceToHost, s));
  DBHIP_CHECK(hipMemcpyAsync(&last_keep, keep + got - 1, 4, hipMemcpyDeviceToHost, s));
  DBHIP_CHECK(hipStreamSynchronize(s));
  const uint64_t kept = last_pos + last_keep;
  if (kept) {
    DBHIP_REQUIRE(out_probe_idx && out_build_row, "dbhip_join_probe_binary: NULL pair buffers");
    hipLaunchKernelGGL(compact_pairs_kernel, dim3(grid_for((int64_t)got, 256)), dim3(256), 0, s, cpi, cbi, (int64_t)got, keep, pos, out_probe_idx,
                       out_build_row, out_matched_bitmap);
    DBHIP_LAUNCH_CHECK();
  }
  *out_n_pairs_host = kept;
  return DBHIP_OK;
}

int32_t dbhip_join_destroy_binary(dbhip_join_binary* j) {
  if (!j) return DBHIP_OK;
  if (j->inner) (void)dbhip_join_destroy(j->inner);
  if (j->off) (void)dbhip_free(j->off);
  if (j->data) (void)dbhip_free(j->data);
  delete j;
  return DBHIP_OK;
}

}  // extern "C"

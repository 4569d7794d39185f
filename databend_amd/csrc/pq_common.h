// pq_common.h — what the two modes of the scan-side decode share (k_parquet.hip: host-planned; k_parquet_dev.hip: the page
// payload never passes through the host): the thrift page-header reader, the handle, the PLAIN / view / bit-unpack device helpers
// and the null-spreading kernels. Included by exactly those two translation units.
#pragma once
#include "dev_common.h"
#include "dev_scan.h"
#include "runtime.h"

#include <string.h>

#include <algorithm>
#include <new>
#include <vector>

using namespace dbhip;

#define DBHIP_TRY(x) do { int32_t _rc = (x); if (_rc) return _rc; } while (0)

namespace {

// ---------------------------------------------------------------------------------------------
// host: thrift compact protocol (only what PageHeader needs)
// ---------------------------------------------------------------------------------------------
struct Rd {
  const uint8_t* p;
  const uint8_t* end;
  bool ok;
  uint8_t u8() {
    if (p >= end) { ok = false; return 0; }
    return *p++;
  }
  uint64_t varint() {
    uint64_t v = 0;
    for (int sh = 0; sh < 64; sh += 7) {
      const uint8_t b = u8();
      v |= (uint64_t)(b & 0x7F) << sh;
      if (!(b & 0x80)) return v;
      if (!ok) return 0;
    }
    ok = false;
    return 0;
  }
  int64_t zigzag() {
    const uint64_t v = varint();
    return (int64_t)(v >> 1) ^ -(int64_t)(v & 1);
  }
  void skip_bytes(uint64_t n) {
    if ((uint64_t)(end - p) < n) { ok = false; p = end; return; }
    p += n;
  }
  void skip(int type, int depth = 0);
  void skip_struct(int depth) {
    int16_t last = 0;
    while (ok) {
      const uint8_t h = u8();
      if (h == 0) return;
      const int type = h & 0x0F;
      if ((h >> 4) == 0) last = (int16_t)zigzag(); else last = (int16_t)(last + (h >> 4));
      skip(type, depth + 1);
    }
  }
};

void Rd::skip(int type, int depth) {
  if (depth > 16) { ok = false; return; }
  switch (type) {
    case 1: case 2: return;                 // bool carried in the field header
    case 3: u8(); return;                   // byte
    case 4: case 5: case 6: varint(); return;  // i16 / i32 / i64
    case 7: skip_bytes(8); return;          // double
    case 8: skip_bytes(varint()); return;   // binary
    case 9: case 10: {                      // list / set
      const uint8_t h = u8();
      uint64_t n = h >> 4;
      if (n == 15) n = varint();
      const int et = h & 0x0F;
      for (uint64_t i = 0; i < n && ok; ++i) {
        if (et == 1 || et == 2) u8(); else skip(et, depth + 1);
      }
      return;
    }
    case 11: {                              // map
      const uint64_t n = varint();
      if (n == 0) return;
      const uint8_t kv = u8();
      for (uint64_t i = 0; i < n && ok; ++i) { skip(kv >> 4, depth + 1); skip(kv & 0x0F, depth + 1); }
      return;
    }
    case 12: skip_struct(depth); return;
    default: ok = false; return;
  }
}

enum { PG_DATA = 0, PG_INDEX = 1, PG_DICT = 2, PG_DATA_V2 = 3 };
enum { ENC_PLAIN = 0, ENC_PLAIN_DICT = 2, ENC_RLE = 3, ENC_RLE_DICT = 8 };
enum { PT_BOOLEAN = 0, PT_INT32 = 1, PT_INT64 = 2, PT_INT96 = 3, PT_FLOAT = 4, PT_DOUBLE = 5, PT_BYTE_ARRAY = 6, PT_FLBA = 7 };

struct PageHdr {
  int32_t type = -1, uncompressed = -1, compressed = -1;
  int32_t num_values = -1, encoding = -1, def_enc = ENC_RLE, rep_enc = ENC_RLE;
  int32_t num_nulls = -1, def_len = 0, rep_len = 0;
  bool v2_compressed = true;
};

// fields of DataPageHeader (1 num_values, 2 encoding, 3 definition_level_encoding, 4 repetition_level_encoding), DictionaryPageHeader
// (1 num_values, 2 encoding) and DataPageHeaderV2 (1 num_values, 2 num_nulls, 3 num_rows, 4 encoding,
// 5 definition_levels_byte_length, 6 repetition_levels_byte_length, 7 is_compressed) — parquet.thrift
void read_sub(Rd& r, PageHdr& h, int which) {
  int16_t last = 0;
  while (r.ok) {
    const uint8_t b = r.u8();
    if (b == 0) return;
    const int type = b & 0x0F;
    if ((b >> 4) == 0) last = (int16_t)r.zigzag(); else last = (int16_t)(last + (b >> 4));
    const bool is_int = type == 4 || type == 5 || type == 6;
    if (which == PG_DATA_V2) {
      if (last == 7 && (type == 1 || type == 2)) { h.v2_compressed = type == 1; continue; }
      if (is_int && last >= 1 && last <= 6) {
        const int32_t v = (int32_t)r.zigzag();
        if (last == 1) h.num_values = v; else if (last == 2) h.num_nulls = v; else if (last == 4) h.encoding = v;
        else if (last == 5) h.def_len = v; else if (last == 6) h.rep_len = v;
        continue;
      }
    } else {
      if (is_int && last >= 1 && last <= 4) {
        const int32_t v = (int32_t)r.zigzag();
        if (last == 1) h.num_values = v; else if (last == 2) h.encoding = v; else if (which == PG_DATA && last == 3) h.def_enc = v;
        else if (which == PG_DATA && last == 4) h.rep_enc = v;
        continue;
      }
    }
    r.skip(type);
  }
}

bool read_page_header(Rd& r, PageHdr& h) {
  int16_t last = 0;
  while (r.ok) {
    const uint8_t b = r.u8();
    if (b == 0) break;
    const int type = b & 0x0F;
    if ((b >> 4) == 0) last = (int16_t)r.zigzag(); else last = (int16_t)(last + (b >> 4));
    if (type == 5 && last >= 1 && last <= 3) {
      const int32_t v = (int32_t)r.zigzag();
      if (last == 1) h.type = v; else if (last == 2) h.uncompressed = v; else h.compressed = v;
    } else if (type == 12 && last == 5) {
      read_sub(r, h, PG_DATA);
    } else if (type == 12 && last == 7) {
      read_sub(r, h, PG_DICT);
    } else if (type == 12 && last == 8) {
      read_sub(r, h, PG_DATA_V2);
    } else {
      r.skip(type);
    }
  }
  return r.ok && h.type >= 0 && h.compressed >= 0 && h.uncompressed >= 0;
}

// ---------------------------------------------------------------------------------------------
// work items
// ---------------------------------------------------------------------------------------------
enum { IT_LVL_RLE = 0, IT_LVL_BP = 1, IT_IDX_RLE = 2, IT_IDX_BP = 3, IT_PLAIN = 4, IT_STR = 5, IT_BOOL_RLE = 6, IT_BOOL_BP = 7 };
constexpr uint32_t ITEM_MAX = 2048;  // values per item (a multiple of 8: bit-packed splits stay byte aligned)
constexpr uint32_t ITEM_SMALL = 48;  // items up to this many values are expanded by ONE thread each, longer ones by a wave

struct PqItem {
  uint32_t kind;
  uint32_t count;
  uint64_t out_start;  // row index (levels) or ordinal among the non-null values (everything else)
  uint64_t src;        // byte offset into the chunk (bit-packed / plain), or the repeated value (RLE)
  uint32_t bitw;       // bit width (bit-packed indices), unused otherwise
  uint32_t dict_base;  // first entry of this item's dictionary in the dictionary arrays (always 0: one dictionary page per chunk)
};

}  // namespace

// one page of the chunk as its thrift header describes it (device mode, k_parquet_dev.hip): all the host ever learns about the page
struct DvPage {
  uint32_t type;        // PG_DICT / PG_DATA / PG_DATA_V2
  uint32_t enc;         // value encoding (parquet.thrift Encoding)
  uint64_t src_off;     // payload offset in the chunk as stored
  uint32_t comp_len;    // payload bytes as stored
  uint32_t uncomp_len;  // payload bytes once decompressed
  uint64_t img_off;     // where the decompressed payload sits in the image (16-byte aligned); UNCOMPRESSED chunks: = src_off
  uint32_t num_values;  // rows of a data page / entries of the dictionary page
  uint32_t lev_len;     // DATA_PAGE_V2: bytes of definition levels in front of the values (never compressed)
  uint32_t compressed;  // the bytes after the levels are compressed
  uint32_t def_enc;     // v1: encoding of the definition levels
  uint32_t rep_len;     // DATA_PAGE_V2 of a List column: bytes of repetition levels (the first rep_len of the lev_len level bytes)
  uint64_t row_start;   // first row of a data page (List columns: first level entry)
};

struct dbhip_pq_chunk {
  // device mode (dbhip_pq_chunk_open_device): page table from the headers; everything else is found on the device
  bool device_mode = false;
  int32_t codec = 0;
  std::vector<DvPage> pages;                  // in stream order (dictionary page first when there is one)
  std::vector<uint32_t> data_pages;           // index into `pages` of every data page
  std::vector<uint32_t> nn_init, voff_init;   // per data page: non-null count / values offset as far as the header tells (required columns)
  int64_t dict_page = -1, image_len = 0;
  bool known_no_nulls = false;                // required column, or every page is a v2 page that says num_nulls = 0
  DvPage* dv_pages = nullptr; uint32_t* dv_dp = nullptr; uint32_t* dv_nn = nullptr; uint32_t* dv_voff = nullptr;
  uint64_t* dv_vbase = nullptr; uint32_t* dv_ctl = nullptr;
  int32_t physical, type_length, max_def, out_type;
  int64_t chunk_len, rows, nulls, nonnull;
  int64_t n_pages;
  // dictionary
  int64_t dict_n, dict_off, dict_bytes;   // entries, byte offset of the page payload in the chunk
  std::vector<uint32_t> dict_str_off;      // BYTE_ARRAY dictionary: offset of every entry's bytes in the chunk
  std::vector<PqItem> val_items;              // as planned, in stream order; split by size at the first decode:
  int64_t n_val_small;                        // d_val holds the short items first, then the long ones
  std::vector<uint64_t> valid_bits;           // nullable columns: the validity bitmap (= the definition levels), built by open()
  std::vector<uint32_t> str_off;           // PLAIN BYTE_ARRAY data pages: offset of every value's bytes (by ordinal)
  std::vector<uint8_t> image;              // compressed chunks: the decompressed page payloads back to back — what decode() reads
                                           // (and what String views point into) instead of the chunk itself
  // device side (uploaded on first decode)
  uint64_t* d_valid; PqItem* d_val; uint32_t* d_str_off; uint32_t* d_dict_str_off;
  void* d_dict;                            // dictionary in the output type (values or 16-byte views)
  void* d_dense;                           // non-null values, output type (only with nulls)
  uint32_t* d_wcnt; uint64_t* d_woff; uint64_t* d_blk;
  bool uploaded;
  // List<primitive> columns (dbhip_pq_chunk_open_device_list, round 5): `rows` counts LEVEL ENTRIES; the flat pipeline decodes the leaf
  // as a nullable column over the entries (valid = the entry carries a value) into d_ent_values / d_ent_valid, the list pass
  // (pq_list_finish_kernel) then drops the entries that are not elements and builds offsets and validities
  bool list = false;
  int32_t list_nullable = 0, elem_nullable = 0, list_max_def = 0;
  uint32_t* d_isrep = nullptr; uint32_t* d_iselem = nullptr; uint32_t* d_lvalid = nullptr; uint32_t* d_ent_valid = nullptr;
  void* d_ent_values = nullptr;
  uint32_t* d_rcnt = nullptr; uint32_t* d_ecnt = nullptr; uint64_t* d_roff = nullptr; uint64_t* d_eoff = nullptr; uint64_t* d_lblk = nullptr;
  uint64_t* d_lcounts = nullptr;
};

namespace {

int out_elem_size(int32_t t) {
  if (t == DBHIP_T_BOOL) return 0;  // bitmap
  return type_size(t);
}

bool type_pair_ok(int physical, int type_length, int out_type) {
  switch (physical) {
    case PT_BOOLEAN: return out_type == DBHIP_T_BOOL;
    case PT_INT32:
      switch (out_type) {
        case DBHIP_T_I8: case DBHIP_T_I16: case DBHIP_T_I32: case DBHIP_T_U8: case DBHIP_T_U16: case DBHIP_T_U32:
        case DBHIP_T_DATE: case DBHIP_T_DEC64: case DBHIP_T_I64: return true;
        default: return false;
      }
    case PT_INT64:
      return out_type == DBHIP_T_I64 || out_type == DBHIP_T_U64 || out_type == DBHIP_T_TIMESTAMP || out_type == DBHIP_T_DEC64 ||
             out_type == DBHIP_T_DEC128;
    case PT_FLOAT: return out_type == DBHIP_T_F32;
    case PT_DOUBLE: return out_type == DBHIP_T_F64;
    case PT_BYTE_ARRAY: return out_type == DBHIP_T_STRING;
    case PT_FLBA: return (out_type == DBHIP_T_DEC128 && type_length >= 1 && type_length <= 16) ||
                         (out_type == DBHIP_T_DEC64 && type_length >= 1 && type_length <= 8);
    default: return false;
  }
}

__host__ __device__ inline int plain_width(int physical, int type_length) {
  switch (physical) {
    case PT_INT32: case PT_FLOAT: return 4;
    case PT_INT64: case PT_DOUBLE: return 8;
    case PT_FLBA: return type_length;
    default: return 0;
  }
}

struct PqConv {
  int physical, type_length, out_type, esize;
};

__device__ __forceinline__ uint64_t load_le(const uint8_t* p, int n) {  // n <= 8 bytes, unaligned
  uint64_t v = 0;
  for (int b = 0; b < n; ++b) v |= (uint64_t)p[b] << (8 * b);
  return v;
}

// one PLAIN-encoded value at `p` -> element `o` of `out` in the output type
__device__ __forceinline__ void store_plain(const PqConv& cv, const uint8_t* p, void* out, uint64_t o) {
  if (cv.physical == PT_FLBA) {
    // big-endian two's complement of type_length bytes -> sign-extended little-endian integer (decimal)
    const int L = cv.type_length;
    u128 v = (p[0] & 0x80) ? ~(u128)0 : (u128)0;
    for (int b = 0; b < L; ++b) v = (v << 8) | p[b];
    if (cv.esize == 16) ((u128*)out)[o] = v; else ((uint64_t*)out)[o] = (uint64_t)v;
    return;
  }
  if (cv.physical == PT_INT32 || cv.physical == PT_FLOAT) {
    const uint32_t v = (uint32_t)load_le(p, 4);
    switch (cv.esize) {
      case 1: ((uint8_t*)out)[o] = (uint8_t)v; break;
      case 2: ((uint16_t*)out)[o] = (uint16_t)v; break;
      case 4: ((uint32_t*)out)[o] = v; break;
      default: ((int64_t*)out)[o] = (int64_t)(int32_t)v; break;  // Decimal(p <= 9) / widening
    }
    return;
  }
  const uint64_t v = load_le(p, 8);
  if (cv.esize == 16) ((i128*)out)[o] = (i128)(int64_t)v; else ((uint64_t*)out)[o] = v;
}

// 16-byte view of the string whose bytes start at chunk offset `off` (its 4-byte length prefix sits right before)
__device__ __forceinline__ void store_view(const uint8_t* chunk, uint32_t off, void* out, uint64_t o) {
  const uint8_t* p = chunk + off;
  const uint32_t len = (uint32_t)load_le(p - 4, 4);
  uint32_t w[4] = {len, 0, 0, 0};
  if (len <= 12) {
    for (uint32_t b = 0; b < len; ++b) w[1 + (b >> 2)] |= (uint32_t)p[b] << (8 * (b & 3));
  } else {
    w[1] = (uint32_t)load_le(p, 4);
    w[2] = 0;     // buffer index: the chunk itself is buffer 0 of the column
    w[3] = off;
  }
  ((uint4*)out)[o] = make_uint4(w[0], w[1], w[2], w[3]);
}

__device__ __forceinline__ uint32_t extract_bits(const uint8_t* src, uint64_t i, int bitw) {
  const uint64_t bit = i * (uint64_t)bitw;
  const uint8_t* p = src + (bit >> 3);
  const int sh = (int)(bit & 7);
  const int nbytes = (sh + bitw + 7) >> 3;  // <= 5
  const uint64_t v = load_le(p, nbytes);
  return (uint32_t)((v >> sh) & ((bitw >= 32) ? 0xFFFFFFFFu : ((1u << bitw) - 1)));
}

__global__ __launch_bounds__(256) void pq_popc_kernel(const uint32_t* __restrict__ bitmap, int64_t nwords, uint32_t* __restrict__ cnt) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nwords; i += (int64_t)gridDim.x * blockDim.x)
    cnt[i] = (uint32_t)__popc(bitmap[i]);
}

// out[row] = valid ? dense[rank(row)] : 0
template <typename T>
__global__ __launch_bounds__(256) void pq_spread_kernel(const uint32_t* __restrict__ bitmap, const uint64_t* __restrict__ woff,
                                                        const T* __restrict__ dense, int64_t rows, T* __restrict__ out) {
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += (int64_t)gridDim.x * blockDim.x) {
    const uint32_t w = bitmap[r >> 5];
    const int b = (int)(r & 31);
    T v{};
    if ((w >> b) & 1) v = dense[woff[r >> 5] + (uint64_t)__popc(w & ((1u << b) - 1))];
    out[r] = v;
  }
}

// booleans: dense bitmap of the non-null values -> row bitmap
__global__ __launch_bounds__(256) void pq_spread_bool_kernel(const uint32_t* __restrict__ bitmap, const uint64_t* __restrict__ woff,
                                                             const uint32_t* __restrict__ dense, int64_t nwords, uint32_t* __restrict__ out) {
  for (int64_t wi = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; wi < nwords; wi += (int64_t)gridDim.x * blockDim.x) {
    uint32_t w = bitmap[wi], o = 0;
    uint64_t k = woff[wi];
    while (w) {
      const int b = __ffs(w) - 1;
      w &= w - 1;
      o |= ((dense[k >> 5] >> (k & 31)) & 1u) << b;
      ++k;
    }
    out[wi] = o;
  }
}

}  // namespace

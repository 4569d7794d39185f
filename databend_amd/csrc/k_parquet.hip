// k_parquet.hip — scan side: one Parquet column chunk -> one device-resident column.
//
// Stands where the reference hands a block's raw column chunks to arrow-rs:
//   column_chunks_to_record_batch (src/query/storages/fuse/src/io/read/block/parquet/deserialize.rs:33-81:
//   ParquetRecordBatchReader over an in-memory row group) followed by the arrow -> Column conversion
//   (block_reader_parquet_deserialize.rs), i.e. `DataItem::RawData(bytes)` in, one `Column` out.
// The decoder itself is the third-party `parquet` crate (Cargo.lock: parquet 58.1.0, datafuse-extras/arrow-rs
// rev bbbe79543) and is absent from /root/reference; what is restated here is the published Apache Parquet
// format (Encodings.md / thrift definitions) for exactly what the reference's WRITER emits
// (src/query/storages/common/blocks/src/parquet_rs.rs:91-160: one row group, Encoding::PLAIN fallback, statistics
// off, dictionary on => WriterVersion::PARQUET_2_0 (DATA_PAGE_V2 + RLE_DICTIONARY), dictionary off => DATA_PAGE
// v1 + PLAIN), for flat columns (max repetition level 0, max definition level <= 1), TableCompression::None
// (table_compression.rs:38). Compressed chunks, nested columns and the DELTA_* / BYTE_STREAM_SPLIT encodings
// return DBHIP_ERR_UNSUPPORTED (the binding keeps arrow-rs for those).
//
// Split of work. The chunk arrives in HOST memory (object storage read). Everything inherently serial is done
// there, touching only header bytes: thrift page headers, the varint run headers of the RLE / bit-packed hybrid
// streams (levels and dictionary indices), the popcount of bit-packed definition levels (null count per page)
// and, for PLAIN BYTE_ARRAY pages, the chain of 4-byte length prefixes (one u32 offset per value). The result is
// a list of work ITEMS, each <= 2048 values of one kind with a known output position. The device expands the
// items from the HBM-resident copy of the chunk, one wave per item:
//   levels   -> validity bitmap (LSB first), then word popcounts -> exclusive scan = rank of every row
//   values   -> dense array of the non-null values in the OUTPUT type (sign/zero extension, big-endian
//               FIXED_LEN_BYTE_ARRAY decimals -> i128, dictionary gather, 16-byte string views that point INTO the
//               resident chunk: no string byte is copied)
//   spread   -> out[row] = valid ? dense[rank(row)] : 0 (skipped when the chunk has no nulls)
// All three are streaming kernels: the roofline is HBM (chunk bytes in + column bytes out).
#include "pq_common.h"

#include <dlfcn.h>

namespace {

// bits [pos, pos + n) of the LSB-first bitmap `w` := 1
void bits_set(uint64_t* w, uint64_t pos, uint64_t n) {
  while (n) {
    const uint64_t o = pos & 63, take = (64 - o) < n ? (64 - o) : n;
    w[pos >> 6] |= (take == 64 ? ~0ULL : ((1ULL << take) - 1)) << o;
    pos += take;
    n -= take;
  }
}
// bits [pos, pos + n) of `w` := the first n bits of the LSB-first byte stream `src` (`avail` bytes readable); the target bits are zero
void bits_copy(uint64_t* w, uint64_t pos, const uint8_t* src, uint64_t n, uint64_t avail) {
  uint64_t done = 0;
  while (done < n) {
    const uint64_t byte = done >> 3;          // done is a multiple of 8 here except on the last round
    uint64_t v = 0;
    const uint64_t nb = (avail - byte) < 8 ? (avail - byte) : 8;
    memcpy(&v, src + byte, (size_t)nb);       // little endian host
    uint64_t take = (n - done) < 64 ? (n - done) : 64;
    if (take > nb * 8) take = nb * 8;
    if (take == 0) return;                    // (the caller checked that the stream covers n bits)
    if (take < 64) v &= (1ULL << take) - 1;
    const uint64_t p = pos + done, o = p & 63;
    w[p >> 6] |= v << o;
    if (o && take > 64 - o) w[(p >> 6) + 1] |= v >> (64 - o);
    done += take;
  }
}

// Walks one RLE / bit-packed hybrid stream of `nvals` values. `first_out` = output position of its first value.
// Indices / booleans: appends work items (rle_kind / bp_kind). Definition levels (`level_bits` != nullptr, bit width 1): the
// levels ARE the validity bitmap, so they are written straight into it on the host — one bit per row instead of one
// 32-byte item per 8..48-row run for the device to expand (a 3 %-NULL column of 60 M rows: 7.5 MB instead of 224 MB of
// items over PCIe) — and the values equal to 1 are counted (*ones). Returns false on a malformed stream.
bool scan_hybrid(const uint8_t* base, uint64_t off, uint64_t len, int bitw, uint64_t nvals, uint64_t first_out, uint32_t rle_kind,
                 uint32_t bp_kind, std::vector<PqItem>& items, uint64_t* ones, uint64_t* level_bits = nullptr) {
  Rd r{base + off, base + off + len, true};
  const int vbytes = (bitw + 7) / 8;
  uint64_t done = 0;
  while (done < nvals) {
    const uint64_t h = r.varint();
    if (!r.ok) return false;
    if (h & 1) {
      const uint64_t groups = h >> 1;
      const uint64_t avail = (uint64_t)(r.end - r.p);
      // `groups` is an untrusted 64-bit varint: groups * bitw and groups * 8 must not wrap (2^59 groups of 32 bits
      // would wrap to 0 bytes and pass a length check). A stream of `avail` bytes holds at most avail * 8 values.
      // (bit width 0 — a one-entry dictionary — carries no bytes; its run length is clamped to the values still missing.)
      if (groups == 0 || groups > (UINT64_MAX / 64) || (bitw > 0 && groups > avail + 1)) return false;
      const uint64_t bytes = groups * (uint64_t)bitw;
      const uint64_t src0 = (uint64_t)(r.p - base);
      uint64_t n = groups * 8;
      if (n > nvals - done) n = nvals - done;  // padding of the last group
      // a writer may truncate the padding of the last group, but the bytes present must ALWAYS cover the n values that
      // are emitted below (the device expands the items from these bytes without further checks)
      if (n * (uint64_t)bitw > avail * 8) return false;
      if (ones) {
        // popcount of the first n bits (bit width 1)
        const uint8_t* q = base + src0;
        uint64_t c = 0, full = n / 8;
        for (uint64_t i = 0; i < full; ++i) c += (uint64_t)__builtin_popcount(q[i]);
        if (n & 7) c += (uint64_t)__builtin_popcount(q[full] & ((1u << (n & 7)) - 1));
        *ones += c;
      }
      if (level_bits) bits_copy(level_bits, first_out + done, base + src0, n, (uint64_t)(r.end - (base + src0)));
      else for (uint64_t s = 0; s < n; s += ITEM_MAX) {
        const uint32_t c = (uint32_t)((n - s) < ITEM_MAX ? (n - s) : ITEM_MAX);
        items.push_back(PqItem{bp_kind, c, first_out + done + s, src0 + s / 8 * (uint64_t)bitw, (uint32_t)bitw, 0});
      }
      r.skip_bytes(bytes < (uint64_t)(r.end - r.p) ? bytes : (uint64_t)(r.end - r.p));
      done += n;
    } else {
      uint64_t n = h >> 1;
      uint64_t v = 0;
      for (int b = 0; b < vbytes; ++b) v |= (uint64_t)r.u8() << (8 * b);
      if (!r.ok || n == 0) return false;
      if (bitw < 64 && (v >> bitw) != 0) return false;  // a repeated value wider than the stream's bit width (levels: only 0 / 1)
      if (n > nvals - done) n = nvals - done;
      if (ones && v == 1) *ones += n;
      if (level_bits) { if (v == 1) bits_set(level_bits, first_out + done, n); }
      else for (uint64_t s = 0; s < n; s += 1u << 20) {  // an RLE item is a fill: long pieces are fine, but keep several waves busy
        const uint32_t c = (uint32_t)((n - s) < (1u << 20) ? (n - s) : (1u << 20));
        items.push_back(PqItem{rle_kind, c, first_out + done + s, v, (uint32_t)bitw, 0});
      }
      done += n;
    }
  }
  return true;
}

int bits_for(int max_level) {
  int b = 0;
  while ((1 << b) <= max_level) ++b;
  return b;
}

// levels / booleans -> bitmap. LANES = 64: one wave per item (long items); LANES = 1: one thread per item (the short runs a
// nullable column's level stream is made of — a wave per 8..48-value run would idle most of its lanes). Bits are OR-ed into a
// zeroed bitmap (items do not end on word boundaries).
template <int LANES>
__global__ __launch_bounds__(256) void pq_bits_kernel(const PqItem* __restrict__ items, int64_t n_items,
                                                      const uint8_t* __restrict__ chunk, uint32_t* __restrict__ bitmap) {
  const int lane = LANES == 64 ? (threadIdx.x & 63) : 0;
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t first = LANES == 64 ? (tid >> 6) : tid;
  const int64_t step = LANES == 64 ? (((int64_t)gridDim.x * blockDim.x) >> 6) : (int64_t)gridDim.x * blockDim.x;
  for (int64_t it = first; it < n_items; it += step) {
    const PqItem I = items[it];
    const bool rle = I.kind == IT_LVL_RLE || I.kind == IT_BOOL_RLE;
    if (rle && I.src == 0) continue;  // zeros: the bitmap is already clear
    // every lane owns one 32-bit output word per round
    const uint64_t first_word = I.out_start >> 5, last_word = (I.out_start + I.count - 1) >> 5;
    for (uint64_t w = first_word + lane; w <= last_word; w += LANES) {
      const uint64_t lo = w << 5;                       // first row of the word
      const uint64_t a = lo > I.out_start ? lo : I.out_start;
      const uint64_t e = (lo + 32 < I.out_start + I.count) ? lo + 32 : I.out_start + I.count;
      uint32_t bits;
      if (rle) {
        const int nb = (int)(e - a);
        bits = (nb == 32 ? 0xFFFFFFFFu : ((1u << nb) - 1)) << (a - lo);
      } else {
        // source bits [a - out_start, e - out_start) of the packed stream, bit width 1
        const uint64_t s0 = a - I.out_start;
        const uint8_t* p = chunk + I.src + (s0 >> 3);
        const int sh = (int)(s0 & 7);
        const int nb = (int)(e - a);
        const uint64_t v = load_le(p, (sh + nb + 7) >> 3);
        bits = (uint32_t)((v >> sh) & (nb == 32 ? 0xFFFFFFFFull : ((1ull << nb) - 1))) << (a - lo);
      }
      if (bits) {
        if (a == lo && e == lo + 32) bitmap[w] = bits;  // whole word owned by this item
        else atomicOr(&bitmap[w], bits);
      }
    }
  }
}

// values -> dense array in the output type. LANES = 64: one wave per item; LANES = 1: one thread per (short) item.
template <int LANES>
__global__ __launch_bounds__(256) void pq_values_kernel(const PqItem* __restrict__ items, int64_t n_items,
                                                        const uint8_t* __restrict__ chunk, PqConv cv,
                                                        const void* __restrict__ dict, uint32_t dict_n,
                                                        const uint32_t* __restrict__ str_off, void* __restrict__ out) {
  const int lane = LANES == 64 ? (threadIdx.x & 63) : 0;
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t first = LANES == 64 ? (tid >> 6) : tid;
  const int64_t step = LANES == 64 ? (((int64_t)gridDim.x * blockDim.x) >> 6) : (int64_t)gridDim.x * blockDim.x;
  for (int64_t it = first; it < n_items; it += step) {
    const PqItem I = items[it];
    for (uint32_t i = lane; i < I.count; i += LANES) {
      const uint64_t o = I.out_start + i;
      switch (I.kind) {
        case IT_PLAIN:
          store_plain(cv, chunk + I.src + (uint64_t)i * (uint64_t)I.bitw, out, o);
          break;
        case IT_STR:
          store_view(chunk, str_off[o], out, o);
          break;
        default: {
          uint32_t idx = I.kind == IT_IDX_RLE ? (uint32_t)I.src : extract_bits(chunk + I.src, i, (int)I.bitw);
          if (idx >= dict_n) idx = dict_n - 1;  // a corrupt index must not read outside the dictionary (open() guarantees dict_n >= 1)
          switch (cv.esize) {
            case 1: ((uint8_t*)out)[o] = ((const uint8_t*)dict)[idx]; break;
            case 2: ((uint16_t*)out)[o] = ((const uint16_t*)dict)[idx]; break;
            case 4: ((uint32_t*)out)[o] = ((const uint32_t*)dict)[idx]; break;
            case 8: ((uint64_t*)out)[o] = ((const uint64_t*)dict)[idx]; break;
            default: ((uint4*)out)[o] = ((const uint4*)dict)[idx]; break;
          }
        }
      }
    }
  }
}

// dictionary page (PLAIN) -> dictionary in the output type
__global__ __launch_bounds__(256) void pq_dict_kernel(const uint8_t* __restrict__ chunk, uint64_t dict_off, int64_t n, PqConv cv,
                                                      const uint32_t* __restrict__ dict_str_off, void* __restrict__ dict) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    if (cv.physical == PT_BYTE_ARRAY) store_view(chunk, dict_str_off[i], dict, (uint64_t)i);
    else store_plain(cv, chunk + dict_off + (uint64_t)i * (uint64_t)plain_width(cv.physical, cv.type_length), dict, (uint64_t)i);
  }
}

}  // namespace

namespace {

// ---------------------------------------------------------------------------------------------
// host: page decompression. The reference decompresses pages on the CPU too (the parquet crate calls the zstd / lz4 /
// snap crates before a page reaches its decoders); here: libzstd / liblz4 of the system through dlopen (no headers in this
// image), Snappy's raw format decoded in place (format_description.txt of google/snappy). What reaches the device is the
// DECOMPRESSED page stream.
// ---------------------------------------------------------------------------------------------
enum { CODEC_NONE = 0, CODEC_SNAPPY = 1, CODEC_ZSTD = 6, CODEC_LZ4_RAW = 7 };

bool snappy_decompress(const uint8_t* src, size_t n, uint8_t* dst, size_t cap) {
  Rd r{src, src + n, true};
  const uint64_t total = r.varint();
  if (!r.ok || total != cap) return false;
  size_t o = 0;
  while (r.p < r.end) {
    const uint8_t tag = *r.p++;
    size_t len, off;
    switch (tag & 3) {
      case 0: {
        len = (size_t)(tag >> 2) + 1;
        if (len > 60) {
          const int extra = (int)len - 60;  // 1..4 length bytes follow
          if ((size_t)(r.end - r.p) < (size_t)extra) return false;
          len = 0;
          for (int b = 0; b < extra; ++b) len |= (size_t)r.p[b] << (8 * b);
          len += 1;
          r.p += extra;
        }
        if ((size_t)(r.end - r.p) < len || cap - o < len) return false;
        memcpy(dst + o, r.p, len);
        r.p += len;
        o += len;
        continue;
      }
      case 1:
        if (r.end - r.p < 1) return false;
        len = (size_t)((tag >> 2) & 7) + 4;
        off = ((size_t)(tag >> 5) << 8) | *r.p++;
        break;
      case 2:
        if (r.end - r.p < 2) return false;
        len = (size_t)(tag >> 2) + 1;
        off = (size_t)r.p[0] | ((size_t)r.p[1] << 8);
        r.p += 2;
        break;
      default:
        if (r.end - r.p < 4) return false;
        len = (size_t)(tag >> 2) + 1;
        off = (size_t)r.p[0] | ((size_t)r.p[1] << 8) | ((size_t)r.p[2] << 16) | ((size_t)r.p[3] << 24);
        r.p += 4;
        break;
    }
    if (off == 0 || off > o || cap - o < len) return false;
    for (size_t i = 0; i < len; ++i) dst[o + i] = dst[o + i - off];  // may overlap: byte by byte, forward
    o += len;
  }
  return o == cap;
}

typedef size_t (*zstd_decompress_fn)(void*, size_t, const void*, size_t);
typedef unsigned (*zstd_iserror_fn)(size_t);
typedef int (*lz4_safe_fn)(const char*, char*, int, int);

// 1 = ok, 0 = corrupt input, -1 = codec (library) not available
int page_decompress(int codec, const uint8_t* src, size_t n, uint8_t* dst, size_t cap) {
  if (codec == CODEC_SNAPPY) return snappy_decompress(src, n, dst, cap) ? 1 : 0;
  if (codec == CODEC_ZSTD) {
    static zstd_decompress_fn dec = nullptr;
    static zstd_iserror_fn iserr = nullptr;
    static const bool have = [] {
      void* h = dlopen("libzstd.so.1", RTLD_NOW | RTLD_LOCAL);
      if (!h) return false;
      dec = (zstd_decompress_fn)dlsym(h, "ZSTD_decompress");
      iserr = (zstd_iserror_fn)dlsym(h, "ZSTD_isError");
      return dec && iserr;
    }();
    if (!have) return -1;
    const size_t got = dec(dst, cap, src, n);
    return (!iserr(got) && got == cap) ? 1 : 0;
  }
  if (codec == CODEC_LZ4_RAW) {
    static lz4_safe_fn dec = nullptr;
    static const bool have = [] {
      void* h = dlopen("liblz4.so.1", RTLD_NOW | RTLD_LOCAL);
      if (!h) return false;
      dec = (lz4_safe_fn)dlsym(h, "LZ4_decompress_safe");
      return dec != nullptr;
    }();
    if (!have) return -1;
    if (n > 0x7FFFFFFF || cap > 0x7FFFFFFF) return 0;
    return dec((const char*)src, (char*)dst, (int)n, (int)cap) == (int)cap ? 1 : 0;
  }
  return -1;
}

int32_t unsupported(const char* what) {
  set_error("dbhip_pq_chunk_open: %s (keep the CPU reader for this chunk)", what);
  return DBHIP_ERR_UNSUPPORTED;
}
int32_t malformed(const char* what) {
  set_error("dbhip_pq_chunk_open: malformed column chunk: %s", what);
  return DBHIP_ERR_INVALID;
}

// values section of one data page: bytes [pos, endp) of the chunk hold `nn` non-null values in `enc`
int32_t plan_values(dbhip_pq_chunk* c, const uint8_t* base, uint64_t pos, uint64_t endp, uint64_t nn, int enc) {
  const uint64_t o0 = (uint64_t)c->nonnull;
  if (enc == ENC_PLAIN) {
    if (c->physical == PT_BOOLEAN) {
      if ((endp - pos) * 8 < nn) return malformed("boolean page shorter than its values");
      for (uint64_t s = 0; s < nn; s += ITEM_MAX)
        c->val_items.push_back(PqItem{IT_BOOL_BP, (uint32_t)((nn - s) < ITEM_MAX ? (nn - s) : ITEM_MAX), o0 + s, pos + s / 8, 1, 0});
      return DBHIP_OK;
    }
    if (c->physical == PT_BYTE_ARRAY) {
      c->str_off.resize((size_t)(o0 + nn), 0);
      uint64_t off = pos;
      for (uint64_t i = 0; i < nn; ++i) {
        if (endp - off < 4) return malformed("byte array length runs past the page");
        uint32_t len;
        memcpy(&len, base + off, 4);
        if (endp - off - 4 < len) return malformed("byte array runs past the page");
        c->str_off[(size_t)(o0 + i)] = (uint32_t)(off + 4);
        off += 4 + (uint64_t)len;
      }
      for (uint64_t s = 0; s < nn; s += ITEM_MAX)
        c->val_items.push_back(PqItem{IT_STR, (uint32_t)((nn - s) < ITEM_MAX ? (nn - s) : ITEM_MAX), o0 + s, 0, 0, 0});
      return DBHIP_OK;
    }
    const uint64_t w = (uint64_t)plain_width(c->physical, c->type_length);
    if ((endp - pos) / w < nn) return malformed("plain page shorter than its values");
    for (uint64_t s = 0; s < nn; s += ITEM_MAX)
      c->val_items.push_back(PqItem{IT_PLAIN, (uint32_t)((nn - s) < ITEM_MAX ? (nn - s) : ITEM_MAX), o0 + s, pos + s * w, (uint32_t)w, 0});
    return DBHIP_OK;
  }
  if (enc == ENC_PLAIN_DICT || enc == ENC_RLE_DICT) {
    if (c->dict_n < 0) return malformed("dictionary-encoded page without a dictionary page");
    if (nn == 0) return DBHIP_OK;
    if (c->dict_n == 0) return malformed("values refer to an empty dictionary");
    if (endp - pos < 1) return malformed("missing index bit width");
    const int bitw = base[pos];
    if (bitw > 32) return malformed("index bit width > 32");
    if (!scan_hybrid(base, pos + 1, endp - pos - 1, bitw, nn, o0, IT_IDX_RLE, IT_IDX_BP, c->val_items, nullptr))
      return malformed("dictionary index stream");
    return DBHIP_OK;
  }
  if (enc == ENC_RLE && c->physical == PT_BOOLEAN) {
    if (endp - pos < 4) return malformed("missing RLE length");
    uint32_t len;
    memcpy(&len, base + pos, 4);
    if (endp - pos - 4 < len) return malformed("RLE boolean stream runs past the page");
    if (!scan_hybrid(base, pos + 4, len, 1, nn, o0, IT_BOOL_RLE, IT_BOOL_BP, c->val_items, nullptr)) return malformed("RLE boolean stream");
    return DBHIP_OK;
  }
  return unsupported("value encoding other than PLAIN / RLE_DICTIONARY");
}

}  // namespace

extern "C" {

int32_t dbhip_pq_chunk_open(const uint8_t* chunk_host, int64_t chunk_len, int32_t codec, int32_t physical_type, int32_t type_length,
                            int32_t max_def_level, int32_t max_rep_level, int32_t out_type, dbhip_pq_chunk** out_host,
                            dbhip_pq_info* info_host) {
  DBHIP_REQUIRE(chunk_host && out_host && chunk_len >= 0, "dbhip_pq_chunk_open: NULL argument");
  *out_host = nullptr;
  if (codec != CODEC_NONE && codec != CODEC_SNAPPY && codec != CODEC_ZSTD && codec != CODEC_LZ4_RAW)
    return unsupported("compression codec other than UNCOMPRESSED / SNAPPY / ZSTD / LZ4_RAW");
  if (max_rep_level != 0 || max_def_level < 0 || max_def_level > 1) return unsupported("nested column (repetition / definition level > 1)");
  if (chunk_len >= (1LL << 32)) return unsupported("column chunk of 4 GiB or more");
  if (!type_pair_ok(physical_type, type_length, out_type)) {
    set_error("dbhip_pq_chunk_open: physical type %d (length %d) cannot be decoded into dbhip type %d", physical_type, type_length, out_type);
    return DBHIP_ERR_UNSUPPORTED;
  }
  dbhip_pq_chunk* c = new (std::nothrow) dbhip_pq_chunk();
  if (!c) { set_error("dbhip_pq_chunk_open: out of host memory"); return DBHIP_ERR_HIP; }
  c->physical = physical_type; c->type_length = type_length; c->max_def = max_def_level; c->out_type = out_type;
  c->chunk_len = chunk_len; c->rows = 0; c->nulls = 0; c->nonnull = 0; c->n_pages = 0;
  c->dict_n = -1; c->dict_off = 0; c->dict_bytes = 0;
  c->d_valid = nullptr; c->d_val = nullptr; c->d_str_off = nullptr; c->d_dict_str_off = nullptr; c->d_dict = nullptr; c->d_dense = nullptr;
  c->d_wcnt = nullptr; c->d_woff = nullptr; c->d_blk = nullptr; c->uploaded = false; c->n_val_small = 0;
  Rd r{chunk_host, chunk_host + chunk_len, true};
  int32_t rc = DBHIP_OK;
  if (codec != CODEC_NONE) {
    // size of the decompressed image: one pass over the page headers (the image must not move while offsets into it are planned)
    Rd q = r;
    uint64_t total = 0;
    while (q.p < q.end) {
      PageHdr h;
      if (!read_page_header(q, h) || (uint64_t)h.compressed > (uint64_t)(q.end - q.p)) break;  // reported by the main pass
      total += (uint64_t)h.uncompressed;
      q.p += h.compressed;
    }
    if (total >= (1ULL << 32)) { delete c; return unsupported("column chunk that decompresses to 4 GiB or more"); }
    // the sizes come from untrusted headers: refuse absurd expansion before reserving host memory for it
    // (up to 1 GiB is taken at face value — constant PLAIN columns do compress 10^4-fold —, more only at <= 1024x the chunk)
    if (total > (1ULL << 30) && total > 1024ULL * (uint64_t)chunk_len) { delete c; return malformed("declared uncompressed size out of proportion to the chunk"); }
    try { c->image.reserve((size_t)total); } catch (...) { delete c; set_error("dbhip_pq_chunk_open: out of host memory"); return DBHIP_ERR_HIP; }
  }
  while (rc == DBHIP_OK && r.p < r.end) {
    PageHdr h;
    if (!read_page_header(r, h)) { rc = malformed("page header"); break; }
    if ((uint64_t)h.compressed > (uint64_t)(r.end - r.p)) { rc = malformed("page runs past the chunk"); break; }
    const uint8_t* craw = r.p;                 // the page's payload as stored
    const uint8_t* next = craw + h.compressed;
    const uint8_t* base;                       // what the plan's offsets are relative to: the chunk, or the decompressed image
    uint64_t pos0, endp;
    if (codec == CODEC_NONE) {
      if (h.compressed != h.uncompressed) { rc = unsupported("compressed page in a chunk declared UNCOMPRESSED"); break; }
      base = chunk_host;
      pos0 = (uint64_t)(craw - chunk_host);
      endp = pos0 + (uint64_t)h.compressed;
    } else {
      // DATA_PAGE_V2 keeps its levels uncompressed in front of the (optionally) compressed values
      const uint64_t lev = h.type == PG_DATA_V2 ? (uint64_t)(h.rep_len < 0 ? 0 : h.rep_len) + (uint64_t)(h.def_len < 0 ? 0 : h.def_len) : 0;
      if (lev > (uint64_t)h.compressed || lev > (uint64_t)h.uncompressed) { rc = malformed("level bytes exceed the page"); break; }
      const size_t off = c->image.size();
      if (c->image.capacity() - off < (size_t)h.uncompressed) { rc = malformed("page sizes changed between passes"); break; }
      c->image.resize(off + (size_t)h.uncompressed);
      uint8_t* dst = c->image.data() + off;
      memcpy(dst, craw, (size_t)lev);
      const bool comp = h.type == PG_DATA_V2 ? h.v2_compressed : true;
      if (!comp) {
        if (h.compressed != h.uncompressed) { rc = malformed("uncompressed page with differing sizes"); break; }
        memcpy(dst + lev, craw + lev, (size_t)h.uncompressed - (size_t)lev);
      } else if ((uint64_t)h.uncompressed == lev) {
        // nothing to decompress (an empty dictionary page still carries a few bytes of compressed "nothing")
      } else {
        const int ok = page_decompress(codec, craw + lev, (size_t)h.compressed - (size_t)lev, dst + lev, (size_t)h.uncompressed - (size_t)lev);
        if (ok < 0) { rc = unsupported("compression library not available on this host"); break; }
        if (ok == 0) { rc = malformed("page does not decompress to its declared size"); break; }
      }
      base = c->image.data();
      pos0 = (uint64_t)off;
      endp = pos0 + (uint64_t)h.uncompressed;
    }
    if (h.type == PG_DICT) {
      if (c->dict_n >= 0 || c->n_pages > 0) { rc = malformed("dictionary page not first / repeated"); break; }
      if (h.encoding != ENC_PLAIN && h.encoding != ENC_PLAIN_DICT) { rc = unsupported("dictionary page encoding"); break; }
      if (h.num_values < 0 || c->physical == PT_BOOLEAN) { rc = malformed("dictionary page"); break; }
      c->dict_n = h.num_values; c->dict_off = (int64_t)pos0; c->dict_bytes = (int64_t)(endp - pos0);
      if (c->physical == PT_BYTE_ARRAY) {
        uint64_t off = pos0;
        c->dict_str_off.resize((size_t)c->dict_n);
        for (int64_t i = 0; i < c->dict_n; ++i) {
          if (endp - off < 4) { rc = malformed("dictionary entry length"); break; }
          uint32_t len;
          memcpy(&len, base + off, 4);
          if (endp - off - 4 < len) { rc = malformed("dictionary entry runs past the page"); break; }
          c->dict_str_off[(size_t)i] = (uint32_t)(off + 4);
          off += 4 + (uint64_t)len;
        }
      } else if ((uint64_t)c->dict_n * (uint64_t)plain_width(c->physical, c->type_length) > endp - pos0) {
        rc = malformed("dictionary page shorter than its entries");
      }
    } else if (h.type == PG_DATA || h.type == PG_DATA_V2) {
      if (h.num_values < 0) { rc = malformed("data page without num_values"); break; }
      const uint64_t nv = (uint64_t)h.num_values;
      // (checked per page, before anything is sized by it: an all-NULL page claims 2^31 rows with a 6-byte RLE run)
      if ((uint64_t)c->rows + nv >= 0xFFFFFFF0ULL) { rc = unsupported("more than 2^32 rows in one chunk"); break; }
      uint64_t pos = pos0, nn = nv;
      if (h.type == PG_DATA) {
        if (c->max_def == 1) {
          if (h.def_enc != ENC_RLE) { rc = unsupported("definition levels not RLE encoded"); break; }
          if (endp - pos < 4) { rc = malformed("missing level length"); break; }
          uint32_t len;
          memcpy(&len, base + pos, 4);
          if (endp - pos - 4 < len) { rc = malformed("levels run past the page"); break; }
          uint64_t ones = 0;
          c->valid_bits.resize((size_t)(((uint64_t)c->rows + nv + 63) / 64 + 1), 0);
          if (!scan_hybrid(base, pos + 4, len, 1, nv, (uint64_t)c->rows, IT_LVL_RLE, IT_LVL_BP, c->val_items, &ones, c->valid_bits.data())) {
            rc = malformed("definition level stream");
            break;
          }
          nn = ones;
          pos += 4 + (uint64_t)len;
        }
      } else {
        if (h.rep_len != 0) { rc = unsupported("repetition levels"); break; }
        if (h.def_len < 0 || (uint64_t)h.def_len > endp - pos) { rc = malformed("level byte length"); break; }
        if (c->max_def == 1) {
          uint64_t ones = 0;
          c->valid_bits.resize((size_t)(((uint64_t)c->rows + nv + 63) / 64 + 1), 0);
          if (!scan_hybrid(base, pos, (uint64_t)h.def_len, 1, nv, (uint64_t)c->rows, IT_LVL_RLE, IT_LVL_BP, c->val_items, &ones, c->valid_bits.data())) {
            rc = malformed("definition level stream");
            break;
          }
          nn = ones;
          if (h.num_nulls >= 0 && (uint64_t)h.num_nulls != nv - nn) { rc = malformed("num_nulls disagrees with the definition levels"); break; }
        }
        pos += (uint64_t)h.def_len;
      }
      rc = plan_values(c, base, pos, endp, nn, h.encoding);
      if (rc) break;
      c->rows += (int64_t)nv;
      c->nonnull += (int64_t)nn;
      c->n_pages += 1;
    }  // index pages and unknown page types are skipped
    r.p = next;
  }
  if (rc == DBHIP_OK && c->rows >= 0xFFFFFFF0LL) rc = unsupported("more than 2^32 rows in one chunk");
  if (rc) { delete c; return rc; }
  c->nulls = c->rows - c->nonnull;
  if (info_host) {
    info_host->num_values = c->rows;
    info_host->num_nulls = c->nulls;
    info_host->out_type = out_type;
    info_host->has_validity = c->max_def;
    info_host->out_bytes = out_type == DBHIP_T_BOOL ? ceil_div(c->rows, 64) * 8 : c->rows * (int64_t)out_elem_size(out_type);
    info_host->validity_bytes = ceil_div(c->rows, 64) * 8;
    info_host->n_pages = c->n_pages;
    info_host->n_dict_values = c->dict_n < 0 ? 0 : c->dict_n;
    info_host->image_bytes = (int64_t)c->image.size();
  }
  *out_host = c;
  return DBHIP_OK;
}

int32_t dbhip_pq_chunk_decode(dbhip_pq_chunk* c, const uint8_t* chunk_dev, void* out_values_dev, uint8_t* out_validity_dev,
                              void* stream) {
  DBHIP_REQUIRE(c, "dbhip_pq_chunk_decode: NULL handle");
  DBHIP_REQUIRE(!c->device_mode, "dbhip_pq_chunk_decode: the handle was opened by dbhip_pq_chunk_open_device (use dbhip_pq_chunk_decode_device)");
  if (c->rows == 0) return DBHIP_OK;
  DBHIP_REQUIRE(chunk_dev && out_values_dev, "dbhip_pq_chunk_decode: NULL buffer");
  DBHIP_REQUIRE(c->max_def == 0 || out_validity_dev, "dbhip_pq_chunk_decode: a nullable column needs a validity buffer");
  hipStream_t s = resolve_stream(stream);
  const int esize = out_elem_size(c->out_type);
  const bool is_bool = c->out_type == DBHIP_T_BOOL;
  const int64_t nwords = ceil_div(c->rows, 32);
  if (!c->uploaded) {
    // short items first (stable: both halves stay in stream order, so neighbouring threads / waves write neighbouring output)
    auto split = [](std::vector<PqItem>& v) {
      return (int64_t)(std::stable_partition(v.begin(), v.end(), [](const PqItem& i) { return i.count <= ITEM_SMALL; }) - v.begin());
    };
    c->n_val_small = split(c->val_items);
    if (c->max_def == 1) {
      const size_t vb = (size_t)ceil_div(c->rows, 64) * 8;
      if (!c->d_valid) DBHIP_TRY(dbhip_alloc(vb, (void**)&c->d_valid));
      DBHIP_CHECK(hipMemcpyAsync(c->d_valid, c->valid_bits.data(), vb, hipMemcpyHostToDevice, s));
    }
    if (!c->val_items.empty()) {
      if (!c->d_val) DBHIP_TRY(dbhip_alloc(c->val_items.size() * sizeof(PqItem), (void**)&c->d_val));
      DBHIP_CHECK(hipMemcpyAsync(c->d_val, c->val_items.data(), c->val_items.size() * sizeof(PqItem), hipMemcpyHostToDevice, s));
    }
    if (!c->str_off.empty()) {
      if (!c->d_str_off) DBHIP_TRY(dbhip_alloc(c->str_off.size() * 4, (void**)&c->d_str_off));
      DBHIP_CHECK(hipMemcpyAsync(c->d_str_off, c->str_off.data(), c->str_off.size() * 4, hipMemcpyHostToDevice, s));
    }
    if (!c->dict_str_off.empty()) {
      if (!c->d_dict_str_off) DBHIP_TRY(dbhip_alloc(c->dict_str_off.size() * 4, (void**)&c->d_dict_str_off));
      DBHIP_CHECK(hipMemcpyAsync(c->d_dict_str_off, c->dict_str_off.data(), c->dict_str_off.size() * 4, hipMemcpyHostToDevice, s));
    }
    if (c->dict_n > 0 && !c->d_dict) DBHIP_TRY(dbhip_alloc((size_t)c->dict_n * (size_t)esize, &c->d_dict));
    if (c->nulls > 0) {
      if (!c->d_dense) DBHIP_TRY(dbhip_alloc(is_bool ? (size_t)ceil_div(c->nonnull + 1, 64) * 8 : (size_t)(c->nonnull + 1) * (size_t)esize, &c->d_dense));
      if (!c->d_wcnt) DBHIP_TRY(dbhip_alloc((size_t)nwords * 4, (void**)&c->d_wcnt));
      if (!c->d_woff) DBHIP_TRY(dbhip_alloc((size_t)nwords * 8, (void**)&c->d_woff));
      if (!c->d_blk) DBHIP_TRY(dbhip_alloc((size_t)(ceil_div(nwords, SCAN_TILE) + 2) * 8, (void**)&c->d_blk));
    }
    DBHIP_CHECK(hipStreamSynchronize(s));  // the host vectors may be released by close() right after this call returns
    c->uploaded = true;
  }
  const PqConv cv{c->physical, c->type_length, c->out_type, esize};
  kernel_timer_start(s);
  if (c->dict_n > 0) {
    hipLaunchKernelGGL(pq_dict_kernel, dim3(grid_for(c->dict_n, 256)), dim3(256), 0, s, chunk_dev, (uint64_t)c->dict_off, c->dict_n, cv,
                       c->d_dict_str_off, c->d_dict);
  }
  uint32_t* vbits = (uint32_t*)out_validity_dev;
  if (c->max_def == 1) {
    // the validity bitmap was decoded from the definition levels by open(); it sits on the device since the first decode
    DBHIP_CHECK(hipMemcpyAsync(vbits, c->d_valid, (size_t)ceil_div(c->rows, 64) * 8, hipMemcpyDeviceToDevice, s));
  } else if (vbits) {
    DBHIP_CHECK(hipMemsetAsync(vbits, 0xFF, (size_t)ceil_div(c->rows, 64) * 8, s));
  }
  void* target = c->nulls > 0 ? c->d_dense : out_values_dev;
  if (!c->val_items.empty()) {
    const int64_t ns = c->n_val_small, nl = (int64_t)c->val_items.size() - ns;
    const uint32_t dn = (uint32_t)(c->dict_n > 0 ? c->dict_n : 1);
    if (is_bool) {
      DBHIP_CHECK(hipMemsetAsync(target, 0, (size_t)ceil_div(c->nulls > 0 ? c->nonnull + 1 : c->rows, 64) * 8, s));
      if (ns) hipLaunchKernelGGL(pq_bits_kernel<1>, dim3(grid_for(ns, 256)), dim3(256), 0, s, c->d_val, ns, chunk_dev, (uint32_t*)target);
      if (nl) hipLaunchKernelGGL(pq_bits_kernel<64>, dim3(grid_for(nl * 64, 256)), dim3(256), 0, s, c->d_val + ns, nl, chunk_dev, (uint32_t*)target);
    } else {
      if (ns) hipLaunchKernelGGL(pq_values_kernel<1>, dim3(grid_for(ns, 256)), dim3(256), 0, s, c->d_val, ns, chunk_dev, cv,
                                 (const void*)c->d_dict, dn, c->d_str_off, target);
      if (nl) hipLaunchKernelGGL(pq_values_kernel<64>, dim3(grid_for(nl * 64, 256)), dim3(256), 0, s, c->d_val + ns, nl, chunk_dev, cv,
                                 (const void*)c->d_dict, dn, c->d_str_off, target);
    }
  }
  if (c->nulls > 0) {
    hipLaunchKernelGGL(pq_popc_kernel, dim3(grid_for(nwords, 256)), dim3(256), 0, s, vbits, nwords, c->d_wcnt);
    DBHIP_TRY(dbscan::exclusive_scan_u32(c->d_wcnt, nwords, c->d_blk, c->d_woff, s));
    const int grid = grid_for(c->rows, 256);
    if (is_bool) {
      hipLaunchKernelGGL(pq_spread_bool_kernel, dim3(grid_for(nwords, 256)), dim3(256), 0, s, vbits, c->d_woff, (const uint32_t*)c->d_dense,
                         nwords, (uint32_t*)out_values_dev);
      if (nwords & 1) DBHIP_CHECK(hipMemsetAsync((uint32_t*)out_values_dev + nwords, 0, 4, s));
    } else if (esize == 1) {
      hipLaunchKernelGGL(pq_spread_kernel<uint8_t>, dim3(grid), dim3(256), 0, s, vbits, c->d_woff, (const uint8_t*)c->d_dense, c->rows, (uint8_t*)out_values_dev);
    } else if (esize == 2) {
      hipLaunchKernelGGL(pq_spread_kernel<uint16_t>, dim3(grid), dim3(256), 0, s, vbits, c->d_woff, (const uint16_t*)c->d_dense, c->rows, (uint16_t*)out_values_dev);
    } else if (esize == 4) {
      hipLaunchKernelGGL(pq_spread_kernel<uint32_t>, dim3(grid), dim3(256), 0, s, vbits, c->d_woff, (const uint32_t*)c->d_dense, c->rows, (uint32_t*)out_values_dev);
    } else if (esize == 8) {
      hipLaunchKernelGGL(pq_spread_kernel<uint64_t>, dim3(grid), dim3(256), 0, s, vbits, c->d_woff, (const uint64_t*)c->d_dense, c->rows, (uint64_t*)out_values_dev);
    } else {
      hipLaunchKernelGGL(pq_spread_kernel<uint4>, dim3(grid), dim3(256), 0, s, vbits, c->d_woff, (const uint4*)c->d_dense, c->rows, (uint4*)out_values_dev);
    }
  }
  kernel_timer_stop(s);
  DBHIP_LAUNCH_CHECK();
  return DBHIP_OK;
}

int32_t dbhip_pq_chunk_validity(dbhip_pq_chunk* c, const uint8_t** out_ptr_host, int64_t* out_bytes_host) {
  DBHIP_REQUIRE(c && out_ptr_host && out_bytes_host, "dbhip_pq_chunk_validity: NULL argument");
  const bool have = c->max_def == 1 && c->rows > 0;
  *out_ptr_host = have ? (const uint8_t*)c->valid_bits.data() : nullptr;
  *out_bytes_host = have ? ceil_div(c->rows, 64) * 8 : 0;
  return DBHIP_OK;
}

int32_t dbhip_pq_chunk_image(dbhip_pq_chunk* c, const uint8_t** out_ptr_host, int64_t* out_len_host) {
  DBHIP_REQUIRE(c && out_ptr_host && out_len_host, "dbhip_pq_chunk_image: NULL argument");
  *out_ptr_host = c->image.empty() ? nullptr : c->image.data();
  *out_len_host = (int64_t)c->image.size();
  return DBHIP_OK;
}

int32_t dbhip_pq_chunk_close(dbhip_pq_chunk* c) {
  if (!c) return DBHIP_OK;
  (void)hipDeviceSynchronize();
  void* ptrs[] = {c->d_valid, c->d_val, c->d_str_off, c->d_dict_str_off, c->d_dict, c->d_dense, c->d_wcnt, c->d_woff, c->d_blk,
                  c->dv_pages, c->dv_dp, c->dv_nn, c->dv_voff, c->dv_vbase, c->dv_ctl,
                  c->d_isrep, c->d_iselem, c->d_lvalid, c->d_ent_valid, c->d_ent_values, c->d_rcnt, c->d_ecnt, c->d_roff, c->d_eoff, c->d_lblk, c->d_lcounts};
  for (void* p : ptrs)
    if (p) (void)dbhip_free(p);
  delete c;
  return DBHIP_OK;
}

}  // extern "C"

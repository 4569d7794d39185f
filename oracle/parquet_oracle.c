/*
 * parquet_oracle.c — CPU restatement of the Parquet column-chunk decode.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle.h): the checker of dbhip_pq_chunk_*. Never linked into libdbhip.so.
 *
 * What it restates: the reference reads a block's column chunks with arrow-rs
 * (src/query/storages/fuse/src/io/read/block/parquet/deserialize.rs:33-81, ParquetRecordBatchReader). That decoder
 * is the third-party `parquet` crate (Cargo.lock: parquet 58.1.0, datafuse-extras/arrow-rs rev bbbe79543), absent
 * from /root/reference, so this file follows the published Apache Parquet format instead (parquet.thrift PageHeader /
 * DataPageHeader / DataPageHeaderV2 / DictionaryPageHeader, Encodings.md: PLAIN, RLE / bit-packed hybrid,
 * RLE_DICTIONARY) for what the reference's writer emits (storages/common/blocks/src/parquet_rs.rs:91-160).
 * Pinned against pyarrow 25 (the Arrow C++ implementation of the same format) on the fixtures of
 * tests/golden/parquet/ and on files written on the fly by the tests; "parity pinned on the format's reference
 * implementation, not on arrow-rs itself" — stated in DESIGN.md.
 *
 * Deliberately the simplest possible shape — one value at a time through a streaming hybrid reader — so that it
 * shares nothing with the device plan (work items, ranks, dense + spread).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

enum { PT_BOOLEAN = 0, PT_INT32 = 1, PT_INT64 = 2, PT_FLOAT = 4, PT_DOUBLE = 5, PT_BYTE_ARRAY = 6, PT_FLBA = 7 };
enum { T_BOOL = 1, T_I8, T_I16, T_I32, T_I64, T_U8, T_U16, T_U32, T_U64, T_F32, T_F64, T_DATE, T_TIMESTAMP, T_DEC64, T_DEC128, T_STRING };

typedef struct { const uint8_t* p; const uint8_t* end; int bad; } rd_t;

static uint8_t rd_u8(rd_t* r) { if (r->p >= r->end) { r->bad = 1; return 0; } return *r->p++; }
static uint64_t rd_varint(rd_t* r) {
  uint64_t v = 0;
  for (int sh = 0; sh < 70; sh += 7) {
    uint8_t b = rd_u8(r);
    v |= (uint64_t)(b & 0x7F) << sh;
    if (!(b & 0x80) || r->bad) return v;
  }
  r->bad = 1;
  return 0;
}
static int64_t rd_zz(rd_t* r) { uint64_t v = rd_varint(r); return (int64_t)(v >> 1) ^ -(int64_t)(v & 1); }
static void rd_skip(rd_t* r, int type, int depth);
static void rd_skip_struct(rd_t* r, int depth) {
  for (;;) {
    uint8_t h = rd_u8(r);
    if (h == 0 || r->bad) return;
    if ((h >> 4) == 0) (void)rd_zz(r);
    rd_skip(r, h & 15, depth + 1);
  }
}
static void rd_skip(rd_t* r, int type, int depth) {
  if (depth > 16) { r->bad = 1; return; }
  switch (type) {
    case 1: case 2: return;
    case 3: (void)rd_u8(r); return;
    case 4: case 5: case 6: (void)rd_varint(r); return;
    case 7: if (r->end - r->p < 8) r->bad = 1; else r->p += 8; return;
    case 8: { uint64_t n = rd_varint(r); if ((uint64_t)(r->end - r->p) < n) r->bad = 1; else r->p += n; return; }
    case 9: case 10: {
      uint8_t h = rd_u8(r);
      uint64_t n = h >> 4;
      if (n == 15) n = rd_varint(r);
      for (uint64_t i = 0; i < n && !r->bad; ++i) { if ((h & 15) <= 2) (void)rd_u8(r); else rd_skip(r, h & 15, depth + 1); }
      return;
    }
    case 11: {
      uint64_t n = rd_varint(r);
      if (!n) return;
      uint8_t kv = rd_u8(r);
      for (uint64_t i = 0; i < n && !r->bad; ++i) { rd_skip(r, kv >> 4, depth + 1); rd_skip(r, kv & 15, depth + 1); }
      return;
    }
    case 12: rd_skip_struct(r, depth); return;
    default: r->bad = 1;
  }
}

typedef struct {
  int type, usize, csize;          /* PageHeader 1,2,3 */
  int nvals, enc, def_enc;         /* (Data|Dictionary)PageHeader 1,2,3 / V2 1,4 */
  int nnulls, def_len, rep_len;    /* V2 2,5,6 */
} page_t;

static void read_inner(rd_t* r, page_t* pg, int v2) {
  int id = 0;
  for (;;) {
    uint8_t h = rd_u8(r);
    if (h == 0 || r->bad) return;
    int type = h & 15;
    if ((h >> 4) == 0) id = (int)rd_zz(r); else id += h >> 4;
    if (type >= 4 && type <= 6) {
      int v = (int)rd_zz(r);
      if (v2) { if (id == 1) pg->nvals = v; else if (id == 2) pg->nnulls = v; else if (id == 4) pg->enc = v; else if (id == 5) pg->def_len = v; else if (id == 6) pg->rep_len = v; }
      else { if (id == 1) pg->nvals = v; else if (id == 2) pg->enc = v; else if (id == 3) pg->def_enc = v; }
    } else {
      rd_skip(r, type, 0);
    }
  }
}

static int read_page(rd_t* r, page_t* pg) {
  memset(pg, 0, sizeof(*pg));
  pg->type = pg->usize = pg->csize = pg->nvals = pg->enc = pg->nnulls = -1;
  pg->def_enc = 3;
  int id = 0;
  for (;;) {
    uint8_t h = rd_u8(r);
    if (r->bad) return -1;
    if (h == 0) break;
    int type = h & 15;
    if ((h >> 4) == 0) id = (int)rd_zz(r); else id += h >> 4;
    if (type == 5 && id <= 3) { int v = (int)rd_zz(r); if (id == 1) pg->type = v; else if (id == 2) pg->usize = v; else pg->csize = v; }
    else if (type == 12 && (id == 5 || id == 7)) read_inner(r, pg, 0);
    else if (type == 12 && id == 8) read_inner(r, pg, 1);
    else rd_skip(r, type, 0);
  }
  return (r->bad || pg->type < 0 || pg->csize < 0) ? -1 : 0;
}

/* streaming RLE / bit-packed hybrid reader (Encodings.md "Run Length Encoding / Bit-Packing Hybrid") */
typedef struct {
  rd_t r; int bitw;
  uint64_t rle_left, rle_val;
  uint64_t bp_left; uint64_t bitpos; const uint8_t* bp_base; uint64_t bp_bytes;
} hyb_t;

static void hyb_init(hyb_t* h, const uint8_t* p, uint64_t len, int bitw) {
  memset(h, 0, sizeof(*h));
  h->r.p = p; h->r.end = p + len; h->bitw = bitw;
}
static int hyb_next(hyb_t* h, uint32_t* out) {
  for (;;) {
    if (h->rle_left) { --h->rle_left; *out = (uint32_t)h->rle_val; return 0; }
    if (h->bp_left) {
      uint64_t v = 0;
      for (int b = 0; b < h->bitw; ++b) {
        uint64_t bit = h->bitpos + (uint64_t)b;
        if ((bit >> 3) >= h->bp_bytes) return -1;
        v |= (uint64_t)((h->bp_base[bit >> 3] >> (bit & 7)) & 1) << b;
      }
      h->bitpos += (uint64_t)h->bitw;
      --h->bp_left;
      *out = (uint32_t)v;
      return 0;
    }
    uint64_t hd = rd_varint(&h->r);
    if (h->r.bad) return -1;
    if (hd & 1) {
      uint64_t groups = hd >> 1, bytes = groups * (uint64_t)h->bitw;
      uint64_t avail = (uint64_t)(h->r.end - h->r.p);
      if (groups == 0) return -1;
      h->bp_left = groups * 8; h->bitpos = 0; h->bp_base = h->r.p;
      h->bp_bytes = bytes < avail ? bytes : avail;   /* a truncated final group is caught bit by bit */
      h->r.p += h->bp_bytes;
    } else {
      uint64_t v = 0;
      for (int b = 0; b < (h->bitw + 7) / 8; ++b) v |= (uint64_t)rd_u8(&h->r) << (8 * b);
      if (h->r.bad || (hd >> 1) == 0) return -1;
      h->rle_left = hd >> 1; h->rle_val = v;
    }
  }
}

static int esize_of(int t) {
  switch (t) {
    case T_I8: case T_U8: return 1;
    case T_I16: case T_U16: return 2;
    case T_I32: case T_U32: case T_F32: case T_DATE: return 4;
    case T_I64: case T_U64: case T_F64: case T_TIMESTAMP: case T_DEC64: return 8;
    case T_DEC128: case T_STRING: return 16;
    default: return 0;
  }
}

/* one PLAIN value at p -> out element `o` */
static void put_plain(int physical, int tlen, int out_type, const uint8_t* p, uint8_t* out, int64_t o) {
  int es = esize_of(out_type);
  if (physical == PT_FLBA) {
    unsigned __int128 v = (p[0] & 0x80) ? ~(unsigned __int128)0 : 0;
    for (int b = 0; b < tlen; ++b) v = (v << 8) | p[b];
    memcpy(out + o * es, &v, (size_t)es);  /* little endian host */
    return;
  }
  if (physical == PT_INT32 || physical == PT_FLOAT) {
    int32_t v;
    memcpy(&v, p, 4);
    if (es == 8) { int64_t w = v; memcpy(out + o * 8, &w, 8); }
    else memcpy(out + o * es, &v, (size_t)es);
    return;
  }
  int64_t v;
  memcpy(&v, p, 8);
  if (es == 16) { __int128 w = v; memcpy(out + o * 16, &w, 16); } else memcpy(out + o * 8, &v, 8);
}

static void put_view(const uint8_t* chunk, uint64_t off, uint8_t* out, int64_t o) {
  uint32_t len, w[4] = {0, 0, 0, 0};
  memcpy(&len, chunk + off - 4, 4);
  w[0] = len;
  if (len <= 12) memcpy(&w[1], chunk + off, len);
  else { memcpy(&w[1], chunk + off, 4); w[2] = 0; w[3] = (uint32_t)off; }
  memcpy(out + o * 16, w, 16);
}

/* Returns 0, -1 (malformed) or -2 (unsupported). out_valid: one byte per row. BOOL values: one byte per row too. */
/* Snappy raw format (google/snappy format_description.txt): varint uncompressed length, then literal / copy elements.
 * -> bytes written, -1 on malformed input. Used for the SNAPPY-compressed chunks of the reference's own test files. */
static int64_t snappy_raw(const uint8_t* in, int64_t n, uint8_t* out, int64_t cap) {
  int64_t ip = 0, op = 0;
  uint64_t ulen = 0; int sh = 0;
  for (;;) { if (ip >= n || sh > 35) return -1; uint8_t b = in[ip++]; ulen |= (uint64_t)(b & 0x7f) << sh; if (!(b & 0x80)) break; sh += 7; }
  if ((int64_t)ulen > cap) return -1;
  while (ip < n) {
    const uint8_t tag = in[ip++];
    int64_t l, off = 0;
    if ((tag & 3) == 0) {
      l = (tag >> 2) + 1;
      if (l > 60) { const int nb = (int)l - 60; if (ip + nb > n) return -1; l = 0; for (int k = 0; k < nb; ++k) l |= (int64_t)in[ip + k] << (8 * k); l += 1; ip += nb; }
      if (ip + l > n || op + l > (int64_t)ulen) return -1;
      memcpy(out + op, in + ip, (size_t)l); ip += l; op += l;
      continue;
    }
    if ((tag & 3) == 1) { if (ip + 1 > n) return -1; l = 4 + ((tag >> 2) & 7); off = ((int64_t)(tag >> 5) << 8) | in[ip]; ip += 1; }
    else if ((tag & 3) == 2) { if (ip + 2 > n) return -1; l = (tag >> 2) + 1; off = in[ip] | ((int64_t)in[ip + 1] << 8); ip += 2; }
    else { if (ip + 4 > n) return -1; l = (tag >> 2) + 1; off = in[ip] | ((int64_t)in[ip + 1] << 8) | ((int64_t)in[ip + 2] << 16) | ((int64_t)in[ip + 3] << 24); ip += 4; }
    if (off == 0 || off > op || op + l > (int64_t)ulen) return -1;
    for (int64_t k = 0; k < l; ++k) out[op + k] = out[op + k - off];   /* byte by byte: the copy may overlap itself */
    op += l;
  }
  return op == (int64_t)ulen ? op : -1;
}

/* DELTA_BINARY_PACKED (Apache Parquet Encodings.md, "Delta Encoding"): header <block size> <miniblocks per block> <total count>
 * <first value (zigzag)>; each block <min delta (zigzag)> <one bit width per miniblock> <bit-packed deltas>. One value per call. */
typedef struct { rd_t r; uint64_t vpm, mb, total, produced; uint64_t last, min_delta; uint8_t bw[4096]; uint64_t mini, in_mini; const uint8_t* mp; } delta_t;
static int delta_init(delta_t* d, const uint8_t* p, const uint8_t* end) {
  d->r.p = p; d->r.end = end; d->r.bad = 0;
  const uint64_t bs = rd_varint(&d->r);
  d->mb = rd_varint(&d->r);
  d->total = rd_varint(&d->r);
  d->last = (uint64_t)rd_zz(&d->r);
  if (d->r.bad || d->mb == 0 || d->mb > 4096 || bs == 0 || bs > (1u << 24) || bs % d->mb) return -1;
  d->vpm = bs / d->mb;
  if (d->vpm % 8) return -1;
  d->produced = 0; d->mini = d->mb; d->in_mini = 0; d->mp = NULL;
  return 0;
}
static int delta_next(delta_t* d, uint64_t* out) {
  if (d->produced >= d->total) return -1;
  if (d->produced == 0) { d->produced = 1; *out = d->last; return 0; }
  if (d->mp == NULL || d->in_mini == d->vpm) {
    if (d->mp) { d->r.p = d->mp + d->vpm * d->bw[d->mini] / 8 <= d->r.end ? d->mp + d->vpm * d->bw[d->mini] / 8 : d->r.end; d->mini++; }
    if (d->mini >= d->mb) {   /* next block */
      d->min_delta = (uint64_t)rd_zz(&d->r);
      for (uint64_t m = 0; m < d->mb; ++m) d->bw[m] = rd_u8(&d->r);
      if (d->r.bad) return -1;
      d->mini = 0;
    }
    if (d->bw[d->mini] > 64) return -1;
    d->mp = d->r.p;
    d->in_mini = 0;
  }
  const unsigned b = d->bw[d->mini];
  uint64_t v = 0;
  for (unsigned k = 0; k < b; ++k) {
    const uint64_t bit = d->in_mini * b + k;
    if (d->mp + (bit >> 3) >= d->r.end) return -1;
    v |= (uint64_t)((d->mp[bit >> 3] >> (bit & 7)) & 1) << k;
  }
  d->in_mini++;
  d->last += d->min_delta + v;
  d->produced++;
  *out = d->last;
  return 0;
}

static int g_pq_codec = 0;   /* parquet.thrift CompressionCodec of the chunk orc_pq_decode_codec is working on (0 UNCOMPRESSED, 1 SNAPPY) */
int orc_pq_decode(const uint8_t* chunk, int64_t len, int physical, int type_length, int max_def, int out_type, int64_t cap_rows,
                  uint8_t* out_values, uint8_t* out_valid, int64_t* out_rows, int64_t* out_nulls);
/* the same for a SNAPPY chunk of DATA_PAGE v1 / dictionary pages (what parquet-cpp wrote into the reference's tests/data files): every
 * page payload is inflated first. Long strings (> 12 bytes) of such chunks are refused (-2): their views would have to point into an
 * image this function does not keep. */
int orc_pq_decode_codec(const uint8_t* chunk, int64_t len, int codec, int physical, int type_length, int max_def, int out_type, int64_t cap_rows,
                        uint8_t* out_values, uint8_t* out_valid, int64_t* out_rows, int64_t* out_nulls) {
  if (codec != 0 && codec != 1) return -2;
  g_pq_codec = codec;
  const int rc = orc_pq_decode(chunk, len, physical, type_length, max_def, out_type, cap_rows, out_values, out_valid, out_rows, out_nulls);
  g_pq_codec = 0;
  return rc;
}

int orc_pq_decode(const uint8_t* chunk, int64_t len, int physical, int type_length, int max_def, int out_type, int64_t cap_rows,
                  uint8_t* out_values, uint8_t* out_valid, int64_t* out_rows, int64_t* out_nulls) {
  rd_t r = {chunk, chunk + len, 0};
  uint8_t* inflated[64];
  int n_inflated = 0;
  const int es = esize_of(out_type);
  const int pw = physical == PT_INT32 || physical == PT_FLOAT ? 4 : (physical == PT_INT64 || physical == PT_DOUBLE ? 8 : type_length);
  int64_t rows = 0, nulls = 0;
  int64_t dict_n = -1;
  uint8_t* dict = NULL;   /* dictionary converted to the output type */
  int rc = 0;
  while (r.p < r.end && rc == 0) {
    page_t pg;
    if (read_page(&r, &pg)) { rc = -1; break; }
    if ((int64_t)(r.end - r.p) < pg.csize) { rc = -1; break; }
    const uint8_t* pay = r.p;
    const uint8_t* pend = pay + pg.csize;
    r.p = pend;
    const uint8_t* view_base = chunk;
    if (g_pq_codec == 1) {
      if (pg.type == 3 || pg.usize < 0 || n_inflated >= 64) { rc = -2; break; }   /* v2 pages compress the values only: not in the reference's files */
      uint8_t* buf = (uint8_t*)malloc((size_t)pg.usize + 16);
      inflated[n_inflated++] = buf;
      if (snappy_raw(pay, pg.csize, buf, pg.usize) != pg.usize) { rc = -1; break; }
      pay = buf; pend = buf + pg.usize; view_base = NULL;
    } else if (pg.csize != pg.usize) { rc = -2; break; }
    if (pg.type == 2) {
      if (dict_n >= 0 || pg.nvals < 0 || (pg.enc != 0 && pg.enc != 2)) { rc = pg.enc != 0 && pg.enc != 2 ? -2 : -1; break; }
      dict_n = pg.nvals;
      dict = (uint8_t*)calloc((size_t)(dict_n > 0 ? dict_n : 1), 16);
      const uint8_t* q = pay;
      for (int64_t i = 0; i < dict_n; ++i) {
        if (physical == PT_BYTE_ARRAY) {
          uint32_t l;
          if (pend - q < 4) { rc = -1; break; }
          memcpy(&l, q, 4);
          if ((uint64_t)(pend - q - 4) < l) { rc = -1; break; }
          if (!view_base && l > 12) { rc = -2; break; }   /* a long view would have to point into the inflated page */
          put_view(chunk, (uint64_t)(q + 4 - chunk), dict, i);
          q += 4 + l;
        } else {
          if (pend - q < pw) { rc = -1; break; }
          put_plain(physical, type_length, out_type, q, dict, i);
          q += pw;
        }
      }
      continue;
    }
    if (pg.type != 0 && pg.type != 3) continue;
    if (pg.nvals < 0) { rc = -1; break; }
    if (rows + pg.nvals > cap_rows) { rc = -1; break; }
    const uint8_t* q = pay;
    uint8_t* defs = (uint8_t*)malloc((size_t)pg.nvals + 1);
    memset(defs, 1, (size_t)pg.nvals + 1);
    if (pg.type == 0) {
      if (max_def == 1) {
        uint32_t l;
        if (pg.def_enc != 3) { free(defs); rc = -2; break; }
        if (pend - q < 4) { free(defs); rc = -1; break; }
        memcpy(&l, q, 4);
        if ((uint64_t)(pend - q - 4) < l) { free(defs); rc = -1; break; }
        hyb_t h;
        hyb_init(&h, q + 4, l, 1);
        for (int i = 0; i < pg.nvals; ++i) { uint32_t v; if (hyb_next(&h, &v)) { rc = -1; break; } defs[i] = (uint8_t)v; }
        q += 4 + l;
      }
    } else {
      if (pg.rep_len != 0) { free(defs); rc = -2; break; }
      if (pg.def_len < 0 || pend - q < pg.def_len) { free(defs); rc = -1; break; }
      if (max_def == 1) {
        hyb_t h;
        hyb_init(&h, q, (uint64_t)pg.def_len, 1);
        for (int i = 0; i < pg.nvals; ++i) { uint32_t v; if (hyb_next(&h, &v)) { rc = -1; break; } defs[i] = (uint8_t)v; }
      }
      q += pg.def_len;
    }
    if (rc) { free(defs); break; }
    /* value reader state */
    hyb_t vh;
    delta_t dl;
    int use_hyb = 0, boolbit = 0;
    if (pg.enc == 2 || pg.enc == 8) {
      if (dict_n < 0) { free(defs); rc = -1; break; }
      if (pend - q >= 1) { hyb_init(&vh, q + 1, (uint64_t)(pend - q - 1), q[0]); if (q[0] > 32) rc = -1; }
      else hyb_init(&vh, q, 0, 0);
      use_hyb = 1;
    } else if (pg.enc == 3 && physical == PT_BOOLEAN) {
      uint32_t l;
      if (pend - q < 4) { free(defs); rc = -1; break; }
      memcpy(&l, q, 4);
      if ((uint64_t)(pend - q - 4) < l) { free(defs); rc = -1; break; }
      hyb_init(&vh, q + 4, l, 1);
      use_hyb = 2;
    } else if (pg.enc == 5 && (physical == PT_INT32 || physical == PT_INT64)) {
      if (delta_init(&dl, q, pend)) { free(defs); rc = -1; break; }
      use_hyb = 3;
    } else if (pg.enc != 0) {
      free(defs);
      rc = -2;
      break;
    }
    for (int i = 0; i < pg.nvals && rc == 0; ++i) {
      const int64_t o = rows + i;
      out_valid[o] = defs[i];
      if (!defs[i]) {
        ++nulls;
        if (out_type == T_BOOL) out_values[o] = 0; else memset(out_values + o * es, 0, (size_t)es);
        continue;
      }
      if (use_hyb == 1) {
        uint32_t idx;
        if (hyb_next(&vh, &idx) || (int64_t)idx >= dict_n) { rc = -1; break; }
        memcpy(out_values + o * es, dict + (int64_t)idx * es, (size_t)es);
      } else if (use_hyb == 2) {
        uint32_t v;
        if (hyb_next(&vh, &v)) { rc = -1; break; }
        out_values[o] = (uint8_t)v;
      } else if (use_hyb == 3) {
        uint64_t v;
        uint8_t le[8];
        if (delta_next(&dl, &v)) { rc = -1; break; }
        if (physical == PT_INT32) v = (uint64_t)(uint32_t)v;
        memcpy(le, &v, 8);
        put_plain(physical, type_length, out_type, le, out_values, o);
      } else if (physical == PT_BOOLEAN) {
        if (q + (boolbit >> 3) >= pend) { rc = -1; break; }
        out_values[o] = (uint8_t)((q[boolbit >> 3] >> (boolbit & 7)) & 1);
        ++boolbit;
      } else if (physical == PT_BYTE_ARRAY) {
        uint32_t l;
        if (pend - q < 4) { rc = -1; break; }
        memcpy(&l, q, 4);
        if ((uint64_t)(pend - q - 4) < l) { rc = -1; break; }
        if (!view_base && l > 12) { rc = -2; break; }
        put_view(chunk, (uint64_t)(q + 4 - chunk), out_values, o);
        q += 4 + l;
      } else {
        if (pend - q < pw) { rc = -1; break; }
        put_plain(physical, type_length, out_type, q, out_values, o);
        q += pw;
      }
    }
    free(defs);
    rows += pg.nvals;
  }
  free(dict);
  for (int k = 0; k < n_inflated; ++k) free(inflated[k]);
  *out_rows = rows;
  *out_nulls = nulls;
  return rc;
}

/* List<primitive> (round 5): a leaf under ONE repeated ancestor (max_rep = 1, max_def = list_nullable + 1 + elem_nullable), the
 * three-level LIST of LogicalTypes.md that the reference writes for Array(T) and reads through arrow-rs (deserialize.rs:33-81; the
 * record assembly is parquet's "Dremel" levels: a repetition level of 0 starts a row, a definition level below list_nullable + 1 is a
 * NULL / empty list without an element, max_def is a value, the level in between a NULL element). One level entry at a time:
 *   out_offsets[r] .. out_offsets[r + 1]  the elements of row r (Databend's ArrayColumn offsets), out_list_valid[r] = the list is not NULL,
 *   out_values / out_elem_valid           the elements back to back in the output type (a NULL element: zero bytes).
 * UNCOMPRESSED chunks (SNAPPY v1 pages through orc_pq_decode_list_codec: the List<Int64> column of the reference's
 * tests/data/parquet/multi_page files, fixed-width values only), PLAIN / dictionary / RLE-Boolean / DELTA_BINARY_PACKED values, v1 and
 * v2 pages. -> 0, -1 malformed, -2 not handled */
int orc_pq_decode_list(const uint8_t* chunk, int64_t len, int physical, int type_length, int list_nullable, int elem_nullable, int out_type,
                       int64_t cap_entries, uint64_t* out_offsets, uint8_t* out_list_valid, uint8_t* out_values, uint8_t* out_elem_valid,
                       int64_t* out_rows, int64_t* out_elems);
int orc_pq_decode_list_codec(const uint8_t* chunk, int64_t len, int codec, int physical, int type_length, int list_nullable, int elem_nullable, int out_type,
                             int64_t cap_entries, uint64_t* out_offsets, uint8_t* out_list_valid, uint8_t* out_values, uint8_t* out_elem_valid,
                             int64_t* out_rows, int64_t* out_elems) {
  if ((codec != 0 && codec != 1) || (codec == 1 && physical == PT_BYTE_ARRAY)) return -2;
  g_pq_codec = codec;
  const int rc = orc_pq_decode_list(chunk, len, physical, type_length, list_nullable, elem_nullable, out_type, cap_entries, out_offsets, out_list_valid,
                                    out_values, out_elem_valid, out_rows, out_elems);
  g_pq_codec = 0;
  return rc;
}
int orc_pq_decode_list(const uint8_t* chunk, int64_t len, int physical, int type_length, int list_nullable, int elem_nullable, int out_type,
                       int64_t cap_entries, uint64_t* out_offsets, uint8_t* out_list_valid, uint8_t* out_values, uint8_t* out_elem_valid,
                       int64_t* out_rows, int64_t* out_elems) {
  rd_t r = {chunk, chunk + len, 0};
  const int es = esize_of(out_type);
  const int pw = physical == PT_INT32 || physical == PT_FLOAT ? 4 : (physical == PT_INT64 || physical == PT_DOUBLE ? 8 : type_length);
  const int max_def = list_nullable + 1 + elem_nullable;
  const int dw = max_def > 1 ? 2 : 1;
  uint8_t* inflated[256];
  int n_inflated = 0;
  int64_t rows = 0, elems = 0, entries = 0;
  int64_t dict_n = -1;
  uint8_t* dict = NULL;
  int rc = 0;
  while (r.p < r.end && rc == 0) {
    page_t pg;
    if (read_page(&r, &pg)) { rc = -1; break; }
    if ((int64_t)(r.end - r.p) < pg.csize) { rc = -1; break; }
    const uint8_t* pay = r.p;
    const uint8_t* pend = pay + pg.csize;
    r.p = pend;
    if (g_pq_codec == 1) {   /* SNAPPY v1 / dictionary pages: the whole payload is inflated first */
      if (pg.type == 3 || pg.usize < 0 || n_inflated >= 256) { rc = -2; break; }
      uint8_t* buf = (uint8_t*)malloc((size_t)pg.usize + 16);
      inflated[n_inflated++] = buf;
      if (snappy_raw(pay, pg.csize, buf, pg.usize) != pg.usize) { rc = -1; break; }
      pay = buf; pend = buf + pg.usize;
    } else if (pg.csize != pg.usize) { rc = -2; break; }
    if (pg.type == 2) {
      if (dict_n >= 0 || pg.nvals < 0 || (pg.enc != 0 && pg.enc != 2)) { rc = pg.enc != 0 && pg.enc != 2 ? -2 : -1; break; }
      dict_n = pg.nvals;
      dict = (uint8_t*)calloc((size_t)(dict_n > 0 ? dict_n : 1), 16);
      const uint8_t* q = pay;
      for (int64_t i = 0; i < dict_n; ++i) {
        if (physical == PT_BYTE_ARRAY) {
          uint32_t l;
          if (pend - q < 4) { rc = -1; break; }
          memcpy(&l, q, 4);
          if ((uint64_t)(pend - q - 4) < l) { rc = -1; break; }
          put_view(chunk, (uint64_t)(q + 4 - chunk), dict, i);
          q += 4 + l;
        } else {
          if (pend - q < pw) { rc = -1; break; }
          put_plain(physical, type_length, out_type, q, dict, i);
          q += pw;
        }
      }
      continue;
    }
    if (pg.type != 0 && pg.type != 3) continue;
    if (pg.nvals < 0 || entries + pg.nvals > cap_entries) { rc = -1; break; }
    const uint8_t* q = pay;
    hyb_t rh, dh;
    if (pg.type == 0) {   /* v1: [u32 length][repetition levels][u32 length][definition levels][values] */
      uint32_t l;
      if (pend - q < 4) { rc = -1; break; }
      memcpy(&l, q, 4);
      if ((uint64_t)(pend - q - 4) < l) { rc = -1; break; }
      hyb_init(&rh, q + 4, l, 1);
      q += 4 + l;
      if (pend - q < 4) { rc = -1; break; }
      memcpy(&l, q, 4);
      if ((uint64_t)(pend - q - 4) < l) { rc = -1; break; }
      hyb_init(&dh, q + 4, l, dw);
      q += 4 + l;
    } else {              /* v2: the byte lengths come from the header, no prefixes */
      if (pg.rep_len < 0 || pg.def_len < 0 || pend - q < (int64_t)pg.rep_len + pg.def_len) { rc = -1; break; }
      hyb_init(&rh, q, (uint64_t)pg.rep_len, 1);
      hyb_init(&dh, q + pg.rep_len, (uint64_t)pg.def_len, dw);
      q += pg.rep_len + pg.def_len;
    }
    hyb_t vh;
    delta_t dl;
    int use_hyb = 0, boolbit = 0;
    if (pg.enc == 2 || pg.enc == 8) {
      if (dict_n < 0) { rc = -1; break; }
      if (pend - q >= 1) { hyb_init(&vh, q + 1, (uint64_t)(pend - q - 1), q[0]); if (q[0] > 32) rc = -1; }
      else hyb_init(&vh, q, 0, 0);
      use_hyb = 1;
    } else if (pg.enc == 3 && physical == PT_BOOLEAN) {
      uint32_t l;
      if (pend - q < 4) { rc = -1; break; }
      memcpy(&l, q, 4);
      if ((uint64_t)(pend - q - 4) < l) { rc = -1; break; }
      hyb_init(&vh, q + 4, l, 1);
      use_hyb = 2;
    } else if (pg.enc == 5 && (physical == PT_INT32 || physical == PT_INT64)) {
      if (delta_init(&dl, q, pend)) { rc = -1; break; }
      use_hyb = 3;
    } else if (pg.enc != 0) { rc = -2; break; }
    for (int i = 0; i < pg.nvals && rc == 0; ++i) {
      uint32_t rep, def;
      if (hyb_next(&rh, &rep) || hyb_next(&dh, &def) || rep > 1 || (int)def > max_def) { rc = -1; break; }
      if (rep == 0) {                       /* a new row */
        out_offsets[rows] = (uint64_t)elems;
        out_list_valid[rows] = (uint8_t)((int)def >= list_nullable);
        ++rows;
      } else if (entries + i == 0) { rc = -1; break; }   /* the first entry of a chunk starts a row */
      if ((int)def < list_nullable + 1) continue;        /* NULL or empty list: no element */
      const int64_t o = elems++;
      const int present = (int)def == max_def;
      out_elem_valid[o] = (uint8_t)present;
      if (!present) { if (out_type == T_BOOL) out_values[o] = 0; else memset(out_values + o * es, 0, (size_t)es); continue; }
      if (use_hyb == 1) {
        uint32_t idx;
        if (hyb_next(&vh, &idx) || (int64_t)idx >= dict_n) { rc = -1; break; }
        memcpy(out_values + o * es, dict + (int64_t)idx * es, (size_t)es);
      } else if (use_hyb == 2) {
        uint32_t v;
        if (hyb_next(&vh, &v)) { rc = -1; break; }
        out_values[o] = (uint8_t)v;
      } else if (use_hyb == 3) {
        uint64_t v;
        uint8_t le[8];
        if (delta_next(&dl, &v)) { rc = -1; break; }
        if (physical == PT_INT32) v = (uint64_t)(uint32_t)v;
        memcpy(le, &v, 8);
        put_plain(physical, type_length, out_type, le, out_values, o);
      } else if (physical == PT_BOOLEAN) {
        if (q + (boolbit >> 3) >= pend) { rc = -1; break; }
        out_values[o] = (uint8_t)((q[boolbit >> 3] >> (boolbit & 7)) & 1);
        ++boolbit;
      } else if (physical == PT_BYTE_ARRAY) {
        uint32_t l;
        if (pend - q < 4) { rc = -1; break; }
        memcpy(&l, q, 4);
        if ((uint64_t)(pend - q - 4) < l) { rc = -1; break; }
        put_view(chunk, (uint64_t)(q + 4 - chunk), out_values, o);
        q += 4 + l;
      } else {
        if (pend - q < pw) { rc = -1; break; }
        put_plain(physical, type_length, out_type, q, out_values, o);
        q += pw;
      }
    }
    entries += pg.nvals;
  }
  free(dict);
  for (int i = 0; i < n_inflated; ++i) free(inflated[i]);
  out_offsets[rows] = (uint64_t)elems;
  *out_rows = rows;
  *out_elems = elems;
  return rc;
}

/* Concatenated page payloads (levels + values of every page, dictionary page first) of an UNCOMPRESSED column chunk — the
 * byte stream a compressed twin of the same chunk must decompress to (the product's "image"). Returns the number of bytes
 * written, -1 on a malformed chunk, -2 if a page is compressed. */
int64_t orc_pq_payloads(const uint8_t* chunk, int64_t len, uint8_t* out, int64_t cap) {
  rd_t r = {chunk, chunk + len, 0};
  int64_t o = 0;
  while (r.p < r.end) {
    page_t pg;
    if (read_page(&r, &pg)) return -1;
    if (pg.csize != pg.usize) return -2;
    if ((int64_t)(r.end - r.p) < pg.csize || cap - o < pg.csize) return -1;
    memcpy(out + o, r.p, (size_t)pg.csize);
    o += pg.csize;
    r.p += pg.csize;
  }
  return o;
}

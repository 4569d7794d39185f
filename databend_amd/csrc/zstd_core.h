// zstd_core.h — the Zstandard frame format (RFC 8878) as lane-uniform code: everything that PARSES a frame.
//
// Scan side: TableCompression::Zstd is the reference's default (table_compression.rs:27-28, mapped to
// ParquetCompression::ZSTD(ZstdLevel::default()) at :70); the parquet crate hands every page to the zstd crate (= libzstd) before its
// decoders see it. This header restates the FORMAT (frame / block headers, literals section with its Huffman tree description,
// sequences section with its three FSE tables, the repeat-offset rules), not libzstd's code: the entropy tables are laid out for a
// 64-lane wave that walks one page (LDS-resident, 8-byte sequence entries carrying base value and extra bits so that a sequence costs
// one LDS round trip), and every byte that moves goes through four primitives of the caller:
//     lit_begin(kind, pos, n) the block's literals as one forward stream: in the compressed input (Raw_Literals_Block), decoded by the
//                           Huffman stage into the TAIL of the page's own output region (see decode_block), or one repeated byte
//     put_lit(len)          the next len bytes of that stream
//     put_in(pos, len)      a Raw_Block
//     put_fill(byte, len)   an RLE_Block
//     put_match(off, len)   a back-reference
// The file compiles for the device (k_parquet_dev.hip: W = the wave of dv_inflate_zstd_kernel) and for the host
// (tests/zstd_host_check.cpp: W = plain memory), so that the parse logic is fuzzed against the system's libzstd on the CPU; the
// product only ever instantiates it on the device.
//
// Execution model: every function here is called by ALL lanes of the wave with identical arguments ("uniform"). Values read back
// from the tables are made scalar with W::uni(); table writes are done by the lane W::lead() selects. On the host both are trivial.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define ZC_FN __device__ __forceinline__   /* W lives in registers: nothing takes it through memory */
#define ZC_INL __device__ __forceinline__
#define ZC_MEM __device__ __forceinline__
#define ZC_CONST __device__ const
#else
#define ZC_FN static
#define ZC_INL static inline
#define ZC_MEM inline
#define ZC_CONST static const
#endif

#define ZC_LIKELY(x) __builtin_expect(!!(x), 1)
#define ZC_UNLIKELY(x) __builtin_expect(!!(x), 0)

namespace zc {

enum { OK = 0, CORRUPT_ = 1, UNSUPPORTED = 2 };
// CORRUPT_STRICT marks the checks the format demands and libzstd only makes from some version on (the reference pins zstd-sys
// 2.0.16+zstd.1.5.7, Cargo.lock:20810; older libraries let such frames through) — the host check tells the two apart.
#if defined(ZC_TRACE)
#define CORRUPT (zc_trace(__LINE__, 0), zc::CORRUPT_)
#define CORRUPT_STRICT (zc_trace(__LINE__, 1), zc::CORRUPT_)
#else
#define CORRUPT zc::CORRUPT_
#define CORRUPT_STRICT zc::CORRUPT_
#endif

// Literals_Length_Code / Match_Length_Code -> (baseline, extra bits)          RFC 8878 3.1.1.3.2.1.1
ZC_CONST uint32_t LL_BASE[36] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 18, 20, 22, 24, 28, 32, 40,
                                 48, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768, 65536};
ZC_CONST uint8_t LL_BITS[36] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 3, 3, 4, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};
ZC_CONST uint32_t ML_BASE[53] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29,
                                 30, 31, 32, 33, 34, 35, 37, 39, 41, 43, 47, 51, 59, 67, 83, 99, 131, 259, 515, 1027, 2051, 4099, 8195,
                                 16387, 32771, 65539};
ZC_CONST uint8_t ML_BITS[53] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
                                1, 1, 1, 1, 2, 2, 3, 3, 4, 4, 5, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};
// default distributions (3.1.1.3.2.2), accuracy logs 6 / 5 / 6
ZC_CONST int8_t LL_DEFAULT[36] = {4, 3, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 1, 1, 1, 2, 2, 2, 2, 2, 2, 2, 2, 2, 3, 2, 1, 1, 1, 1, 1, -1, -1, -1, -1};
ZC_CONST int8_t OF_DEFAULT[29] = {1, 1, 1, 1, 1, 1, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1};
ZC_CONST int8_t ML_DEFAULT[53] = {1, 4, 3, 2, 2, 2, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1,
                                  1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1, -1, -1};

enum { K_LL = 0, K_OF = 1, K_ML = 2, K_WEIGHTS = 3 };
enum { TAG_NONE = 0, TAG_PREDEF = 1, TAG_OTHER = 2 };

// scratch layout (bytes) inside W::scr()
constexpr uint32_t SCR_SYMS = 0;        // u8[512]   spread symbols
constexpr uint32_t SCR_NEXT = 512;      // u16[64]   next state number per symbol
constexpr uint32_t SCR_NORM = 640;      // i16[64]   normalised counts
constexpr uint32_t SCR_WTAB = 768;      // u32[64]   FSE table of the Huffman weights
constexpr uint32_t SCR_WEIGHTS = 1024;  // u8[256]   Huffman weights
constexpr uint32_t SCR_RANK = 1280;     // u32[16]   first table index of every weight
constexpr uint32_t SCR_CNT = 1344;      // u32[16]   symbols per weight
constexpr uint32_t SCR_PARK = 1408;     // u32[8]    W::park slots
constexpr uint32_t SCR_BYTES = 1440;

constexpr uint32_t HUF_MAXBITS = 11;
constexpr uint32_t BLOCK_MAX = 128 * 1024;

ZC_INL uint32_t hibit(uint32_t v) { return 31u - (uint32_t)__builtin_clz(v); }   // v != 0

// state of one frame (page) between blocks; all scalar. What the sequence tables and the Huffman table in LDS hold is ONE packed word
// (it is live across the sequence loop, where every scalar register counts): accuracy logs in bits 0-3 / 4-7 / 8-11 (LL / OF / ML),
// tags in bits 12-13 / 14-15 / 16-17, Max_Number_of_Bits of the Huffman table in bits 18-21 (0: none yet).
struct Frame {
  uint32_t rep[3];
  uint32_t tabs;
};
ZC_INL uint32_t tabs_pack(uint32_t ll_al, uint32_t of_al, uint32_t ml_al, uint32_t ll_tag, uint32_t of_tag, uint32_t ml_tag, uint32_t huf_bits) {
  return ll_al | (of_al << 4) | (ml_al << 8) | (ll_tag << 12) | (of_tag << 14) | (ml_tag << 16) | (huf_bits << 18);
}

// forward bit reader over the input (FSE table descriptions): LSB first
template <class W>
struct FwdBits {
  uint32_t pos, end;   // byte position / end of the readable region
  uint32_t bit;        // bits consumed from byte `pos`
  uint64_t c;          // bytes [pos, pos + 8)
  bool over;
  ZC_MEM void init(W& w, uint32_t p, uint32_t e) { pos = p; end = e; bit = 0; over = false; c = p < e ? w.in64(p) : 0; }
  ZC_MEM uint32_t read(W& w, uint32_t n) {   // n <= 16
    if (bit + n > 56) {
      pos += bit >> 3; bit &= 7;
      c = pos < end ? w.in64(pos) : 0;
    }
    const uint32_t v = (uint32_t)(c >> bit) & ((1u << n) - 1);
    bit += n;
    return v;
  }
  ZC_MEM void rewind(uint32_t n) { bit -= n; }   // (never below what the last read consumed)
  // byte position after aligning to the next byte; sets `over` when bits past `end` were consumed
  ZC_MEM uint32_t finish() {
    const uint32_t p = pos + ((bit + 7) >> 3);
    if (p > end) over = true;
    return p;
  }
};

// FSE table description (4.1.1): normalised counts into scr[SCR_NORM]. -> accuracy log (0: corrupt), *nsym, *next = position after it
template <class W>
ZC_FN uint32_t read_ncount(W& w, uint32_t pos, uint32_t end, uint32_t max_al, uint32_t max_sym, uint32_t* nsym, uint32_t* next) {
  int16_t* norm = (int16_t*)(w.scr() + SCR_NORM);
  FwdBits<W> b;
  b.init(w, pos, end);
  const uint32_t al = 5 + b.read(w, 4);
  if (al > max_al) return 0;
  int32_t remaining = 1 << al;
  uint32_t s = 0;
  while (remaining > 0 && s <= max_sym) {
    const uint32_t bits = hibit((uint32_t)remaining + 1) + 1;
    uint32_t val = b.read(w, bits);
    const uint32_t lower_mask = (1u << (bits - 1)) - 1;
    const uint32_t threshold = (1u << bits) - 1 - ((uint32_t)remaining + 1);
    if ((val & lower_mask) < threshold) {
      b.rewind(1);
      val &= lower_mask;
    } else if (val > lower_mask) {
      val -= threshold;
    }
    const int32_t proba = (int32_t)val - 1;
    remaining -= proba < 0 ? 1 : proba;
    if (w.lead()) norm[s] = (int16_t)proba;
    ++s;
    if (proba == 0) {
      uint32_t rep = b.read(w, 2);
      for (;;) {
        for (uint32_t i = 0; i < rep && s <= max_sym; ++i) {
          if (w.lead()) norm[s] = 0;
          ++s;
        }
        if (rep != 3) break;
        rep = b.read(w, 2);
        if (b.pos > end) return 0;
      }
    }
    if (b.pos > end) return 0;
  }
  if (remaining != 0 || s > max_sym + 1) return 0;
  *next = b.finish();
  if (b.over) return 0;
  *nsym = s;
  w.sync();
  return al;
}

// decoding table from normalised counts (4.1.1 "from normalized distribution to decoding tables"). kind selects the entry format:
//   K_LL / K_ML  u64: next_base | nbits << 16 | extra_bits << 24 | base_value << 32
//   K_OF         u32: next_base | nbits << 16 | offset_code << 24
//   K_WEIGHTS    u32: next_base | nbits << 16 | symbol << 24
template <class W>
ZC_FN bool fse_build(W& w, uint32_t nsym, uint32_t al, int kind) {
  const int16_t* norm = (const int16_t*)(w.scr() + SCR_NORM);
  uint8_t* syms = w.scr() + SCR_SYMS;
  uint16_t* nxt = (uint16_t*)(w.scr() + SCR_NEXT);
  const uint32_t size = 1u << al;
  bool ok = true;
  if (w.lead()) {
    uint32_t high = size;
    for (uint32_t s = 0; s < nsym; ++s)
      if (norm[s] == -1) {
        if (high == 0) { ok = false; break; }
        syms[--high] = (uint8_t)s;
        nxt[s] = 1;
      }
    const uint32_t step = (size >> 1) + (size >> 3) + 3, mask = size - 1;
    uint32_t p = 0;
    for (uint32_t s = 0; ok && s < nsym; ++s) {
      const int32_t f = norm[s];
      if (f <= 0) continue;
      nxt[s] = (uint16_t)f;
      for (int32_t i = 0; i < f; ++i) {
        syms[p] = (uint8_t)s;
        do { p = (p + step) & mask; } while (p >= high);
      }
    }
    if (p != 0) ok = false;
    for (uint32_t i = 0; ok && i < size; ++i) {
      const uint32_t s = syms[i];
      const uint32_t d = nxt[s]++;
      const uint32_t nb = al - hibit(d);
      const uint32_t base = (d << nb) - size;
      const uint32_t lo = base | (nb << 16);
      if (kind == K_LL) {
        if (s > 35) { ok = false; break; }
        w.llt()[i] = (uint64_t)(lo | ((uint32_t)LL_BITS[s] << 24)) | ((uint64_t)LL_BASE[s] << 32);
      } else if (kind == K_ML) {
        if (s > 52) { ok = false; break; }
        w.mlt()[i] = (uint64_t)(lo | ((uint32_t)ML_BITS[s] << 24)) | ((uint64_t)ML_BASE[s] << 32);
      } else if (kind == K_OF) {
        if (s > 31) { ok = false; break; }
        w.oft()[i] = lo | (s << 24);
      } else {
        ((uint32_t*)(w.scr() + SCR_WTAB))[i] = lo | (s << 24);
      }
    }
  }
  w.sync();
  return w.bcast(ok);
}

template <class W>
ZC_FN bool fse_default(W& w, int kind) {
  int16_t* norm = (int16_t*)(w.scr() + SCR_NORM);
  const uint32_t n = kind == K_LL ? 36 : kind == K_OF ? 29 : 53;
  if (w.lead())
    for (uint32_t s = 0; s < n; ++s) norm[s] = kind == K_LL ? LL_DEFAULT[s] : kind == K_OF ? OF_DEFAULT[s] : ML_DEFAULT[s];
  w.sync();
  return fse_build(w, n, kind == K_OF ? 5 : 6, kind);
}

// a table whose only symbol is `s` (RLE_Mode): accuracy log 0
template <class W>
ZC_FN bool fse_rle(W& w, uint32_t s, int kind) {
  if (kind == K_LL) {
    if (s > 35) return false;
    if (w.lead()) w.llt()[0] = ((uint64_t)LL_BITS[s] << 24) | ((uint64_t)LL_BASE[s] << 32);
  } else if (kind == K_ML) {
    if (s > 52) return false;
    if (w.lead()) w.mlt()[0] = ((uint64_t)ML_BITS[s] << 24) | ((uint64_t)ML_BASE[s] << 32);
  } else {
    if (s > 31) return false;
    if (w.lead()) w.oft()[0] = s << 24;
  }
  w.sync();
  return true;
}

// backward bit reader (4.1 "bitstreams are read in reverse"): stream = input bytes [base, base + len), `off` = bits left below the
// read position (signed: a read past the start yields zeros, as the format's decoders define, and is told by off < 0)
template <class W>
struct BackBits {
  uint32_t base;
  int32_t off;
  uint64_t c;       // bytes [base + cb, base + cb + 8)
  int32_t cbit;     // 8 * cb
  // false: empty stream or no end mark
  ZC_MEM bool init(W& w, uint32_t b, uint32_t len) {
    base = b;
    if (len == 0) return false;
    const uint32_t last = w.in8(b + len - 1);
    if (last == 0) return false;
    off = (int32_t)(len * 8) - (int32_t)(8 - hibit(last));
    cbit = 1 << 30;   // nothing loaded
    c = 0;
    return true;
  }
  ZC_MEM uint32_t read(W& w, uint32_t n) {   // n <= 32
    if (n == 0) return 0;
    off -= (int32_t)n;
    if (off < 0) {
      // bits below the start of the stream read as zero
      const int32_t have = (int32_t)n + off;   // bits of the stream that belong to this read (top part)
      if (have <= 0) return 0;
      const uint64_t v = w.in64(base) & (((uint64_t)1 << have) - 1);
      return (uint32_t)(v << (uint32_t)(-off));
    }
    if (off < cbit || off + (int32_t)n > cbit + 64) {
      int32_t cb = ((off + (int32_t)n + 7) >> 3) - 8;
      if (cb < 0) cb = 0;
      cbit = cb * 8;
      c = w.in64(base + (uint32_t)cb);
    }
    return (uint32_t)(c >> (uint32_t)(off - cbit)) & (uint32_t)(((uint64_t)1 << n) - 1);
  }
};

// The same stream for the SEQUENCES section, where it is the inner loop. The unread bits sit at the BOTTOM of one scalar register pair:
// c = input bytes [base + bb / 8, + 8), of which stream bits [bb, bb + have) are unread, so a field is `have -= n; (c >> have) & mask`
// (four scalar instructions, nothing to shift back) and the only bookkeeping per field is `have`. The register is refilled (56..63
// bits) when a whole sequence — its three fields and the three state updates, whose widths are all known from the table entries
// before the first bit is read — does not fit what is left: one test per sequence instead of one per field.
template <class W>
struct SeqBits {
  uint32_t base;
  int32_t bb;     // stream bit index of bit 0 of c (a multiple of 8)
  int32_t have;   // unread bits in c
  uint64_t c;
  ZC_MEM int32_t left() const { return bb + have; }   // unread bits of the stream
  ZC_MEM bool init(W& w, uint32_t b, uint32_t len) {
    base = b;
    have = 0; c = 0; bb = 0;
    if (len == 0) return false;
    const uint32_t last = w.in8(b + len - 1);
    if (last == 0) return false;
    have = (int32_t)(len * 8) - (int32_t)(8 - hibit(last));   // (not in c yet: refill() brings the top of the stream in)
    return true;
  }
  ZC_MEM void refill(W& w) {
    const int32_t off = bb + have;
    const int32_t b = off > 56 ? (off - 56) >> 3 : 0;
    bb = 8 * b;
    have = off - bb;                                         // 56..63, or what is left of the stream
    c = w.uni64(w.in64(base + (uint32_t)b));
  }
  // the same after the first refill of a stream: the reads only move DOWN from there, which is all the window has to follow
  ZC_MEM void refill_back(W& w) {
    const int32_t off = bb + have;
    const int32_t b = off > 56 ? (off - 56) >> 3 : 0;
    bb = 8 * b;
    have = off - bb;
    c = w.uni64(w.in64_back(base + (uint32_t)b));
  }
  ZC_MEM uint32_t read(uint32_t n) {   // n <= 31 and n <= have
    have -= (int32_t)n;
    return (uint32_t)(c >> (uint32_t)have) & ((1u << n) - 1u);
  }
};

// Huffman tree description (4.2.1) at input [pos, end) -> table w.huf()[1 << maxbits] of (symbol | nbits << 8); -> bytes used, 0: corrupt
template <class W>
ZC_FN uint32_t huf_read_table(W& w, uint32_t pos, uint32_t end, uint32_t* maxbits_out) {
  uint8_t* weights = w.scr() + SCR_WEIGHTS;
  if (pos >= end) return 0;
  const uint32_t hb = w.in8(pos);
  uint32_t used, nw = 0;
  if (hb >= 128) {
    nw = hb - 127;
    used = 1 + ((nw + 1) >> 1);
    if (pos + used > end) return 0;
    for (uint32_t i = 0; i < nw; i += 2) {
      const uint32_t b = w.in8(pos + 1 + (i >> 1));
      if (w.lead()) {
        weights[i] = (uint8_t)(b >> 4);
        if (i + 1 < nw) weights[i + 1] = (uint8_t)(b & 15);
      }
    }
    w.sync();
  } else {
    used = 1 + hb;
    if (hb == 0 || pos + used > end) return 0;
    uint32_t nsym = 0, next = 0;
    const uint32_t al = read_ncount(w, pos + 1, pos + used, 6, 12, &nsym, &next);
    if (al == 0 || next >= pos + used) return 0;
    if (!fse_build(w, nsym, al, K_WEIGHTS)) return 0;
    const uint32_t* wt = (const uint32_t*)(w.scr() + SCR_WTAB);
    BackBits<W> bs;
    if (!bs.init(w, next, pos + used - next)) return 0;
    uint32_t s1 = bs.read(w, al), s2 = bs.read(w, al);
    if (bs.off < 0) return 0;
    for (;;) {
      // two interleaved states; the stream ends when an update runs past its start: the other state's symbol is the last one
      if (nw >= 254) return 0;
      uint32_t e = w.uni(wt[s1]);
      if (w.lead()) weights[nw] = (uint8_t)(e >> 24);
      ++nw;
      s1 = (e & 0xFFFF) + bs.read(w, (e >> 16) & 0xFF);
      if (bs.off < 0) {
        e = w.uni(wt[s2]);
        if (w.lead()) weights[nw] = (uint8_t)(e >> 24);
        ++nw;
        break;
      }
      if (nw >= 254) return 0;
      e = w.uni(wt[s2]);
      if (w.lead()) weights[nw] = (uint8_t)(e >> 24);
      ++nw;
      s2 = (e & 0xFFFF) + bs.read(w, (e >> 16) & 0xFF);
      if (bs.off < 0) {
        e = w.uni(wt[s1]);
        if (w.lead()) weights[nw] = (uint8_t)(e >> 24);
        ++nw;
        break;
      }
    }
    w.sync();
  }
  if (nw == 0 || nw > 255) return 0;
  // the last weight is implied: the weights' 2^(w-1) sum to a power of two
  uint32_t* rank = (uint32_t*)(w.scr() + SCR_RANK);
  uint32_t total = 0;
  for (uint32_t i = 0; i < nw; ++i) {
    const uint32_t x = w.uni((uint32_t)weights[i]);
    if (x > HUF_MAXBITS) return 0;
    if (x) total += 1u << (x - 1);
  }
  if (total == 0) return 0;
  const uint32_t maxbits = hibit(total) + 1;
  if (maxbits > HUF_MAXBITS) return 0;
  const uint32_t left = (1u << maxbits) - total;
  if (left & (left - 1)) return 0;   // (left != 0 because total < 2^maxbits)
  const uint32_t lastw = hibit(left) + 1;
  bool ok = true;
  if (w.lead()) {
    weights[nw] = (uint8_t)lastw;
    // table index ranges: weight 1 symbols first (one cell each), then weight 2 (two cells each), ...
    uint32_t* cnt = (uint32_t*)(w.scr() + SCR_CNT);
    for (uint32_t x = 0; x <= HUF_MAXBITS + 1; ++x) cnt[x] = 0;
    for (uint32_t i = 0; i <= nw; ++i) cnt[weights[i]]++;
    if (cnt[1] < 2 || (cnt[1] & 1)) ok = false;   // (4.2.1: at least two leaves of the longest code, in pairs)
    uint32_t at = 0;
    for (uint32_t x = 1; x <= maxbits; ++x) { rank[x] = at; at += cnt[x] << (x - 1); }
    if (at != (1u << maxbits)) ok = false;
    uint16_t* T = w.huf();
    for (uint32_t i = 0; ok && i <= nw; ++i) {
      const uint32_t x = weights[i];
      if (!x) continue;
      const uint32_t len = 1u << (x - 1), st = rank[x];
      const uint16_t e = (uint16_t)(i | ((maxbits + 1 - x) << 8));
      for (uint32_t k = 0; k < len; ++k) T[st + k] = e;
      rank[x] = st + len;
    }
  }
  w.sync();
  if (!w.bcast(ok)) return 0;
  *maxbits_out = maxbits;
  return used;
}

// ONE Huffman-coded stream (4.2.2), by ONE lane: input bytes [pos, pos + len) -> nsym symbols into the page output at [out, out + nsym)
// through W::lane_in64 / W::lane_store (per-lane addresses: up to four lanes run this side by side).
template <class W>
ZC_INL bool huf_stream(W& w, const uint16_t* T, uint32_t maxbits, uint32_t pos, uint32_t len, uint32_t nsym, uint32_t out) {
  if (len == 0) return false;
  const uint32_t last = w.lane_in8(pos + len - 1);
  if (last == 0) return false;
  int32_t off = (int32_t)(len * 8) - (int32_t)(8 - hibit(last)) - (int32_t)maxbits;   // bit position of the state's lowest bit
  const uint32_t mask = (1u << maxbits) - 1;
  uint64_t c = 0;
  int32_t cbit = 1 << 30;
  for (uint32_t i = 0; i < nsym; ++i) {
    uint32_t state;
    if (off >= 0) {
      if (off < cbit) {
        int32_t cb = ((off + (int32_t)maxbits + 7) >> 3) - 8;
        if (cb < 0) cb = 0;
        cbit = cb * 8;
        c = w.lane_in64(pos + (uint32_t)cb);
      }
      state = (uint32_t)(c >> (uint32_t)(off - cbit)) & mask;
    } else {
      if (off <= -(int32_t)maxbits) return false;   // more symbols asked than the stream holds
      state = (uint32_t)(w.lane_in64(pos) << (uint32_t)(-off)) & mask;
    }
    const uint32_t e = T[state];
    w.lane_store(out + i, (uint8_t)e);
    off -= (int32_t)(e >> 8);
  }
  return off == -(int32_t)maxbits;   // the stream ends exactly where its last symbol does
}

// one of the three sequence tables of a block, by its Symbol_Compression_Mode (3.1.1.3.2.1)
template <class W, int KIND>
ZC_FN int seq_table(W& w, uint32_t mode, uint32_t& al, uint32_t& tag, uint32_t& p, uint32_t end) {
  if (mode == 0) {
    if (tag != TAG_PREDEF) {
      if (!fse_default(w, KIND)) return CORRUPT;
      tag = TAG_PREDEF;
    }
    al = KIND == K_OF ? 5 : 6;
  } else if (mode == 1) {
    if (p >= end) return CORRUPT;
    if (!fse_rle(w, w.in8(p), KIND)) return CORRUPT;
    p += 1; al = 0; tag = TAG_OTHER;
  } else if (mode == 2) {
    uint32_t nsym = 0, next = 0;
    const uint32_t a = read_ncount(w, p, end, KIND == K_OF ? 8 : 9, KIND == K_LL ? 35 : KIND == K_OF ? 31 : 52, &nsym, &next);
    if (a == 0) return CORRUPT;
    if (!fse_build(w, nsym, a, KIND)) return CORRUPT;
    p = next; al = a; tag = TAG_OTHER;
  } else if (tag == TAG_NONE) {
    return CORRUPT;   // Repeat_Mode with nothing to repeat
  }
  return OK;
}

// What the sequence loop carries from one sequence to the next
template <class W>
struct SeqState {
  SeqBits<W> bs;
  uint64_t le_raw, me_raw;   // the table entries of the coming sequence, as read (LDS) while the sequence before it was executed
  uint32_t oe_raw;
  uint32_t r0, r1, r2;       // repeat offsets
};

// ONE sequence. LAST: the last one of the block, which has no state updates behind its fields (3.1.1.3.2.1.1) — a copy of its own,
// so that the loop over the others carries no test for it.
template <class W, bool LAST>
ZC_FN int seq_one(W& w, SeqState<W>& S) {
  SeqBits<W>& bs = S.bs;
  const uint64_t le = w.uni64(S.le_raw), me = w.uni64(S.me_raw);
  const uint32_t oe = w.uni(S.oe_raw);
  // every width of this sequence is in the three entries: offset code, extra bits of the match / literals length, and — between
  // sequences — the bits of the three state updates (literals length, match length, offset)
  const uint32_t ocode = oe >> 24, mlb = (uint32_t)(me >> 24) & 0xFF, llb = (uint32_t)(le >> 24) & 0xFF;
  const uint32_t lnb = LAST ? 0u : ((uint32_t)le >> 16) & 0xFF, mnb = LAST ? 0u : ((uint32_t)me >> 16) & 0xFF, onb = LAST ? 0u : (oe >> 16) & 0xFF;
  const uint32_t tot = ocode + mlb + llb + lnb + mnb + onb;   // <= 31 + 16 + 16 + 26
  if (ZC_UNLIKELY(bs.have < (int32_t)tot)) {
    bs.refill_back(w);
    if (bs.left() < (int32_t)tot) return CORRUPT_STRICT;   // (libzstd < 1.5 reads zeros past the start and only checks the end; 1.5.7 checks for the exact end)
  }
  uint32_t ofx, mlx, llx, lsx = 0, msx = 0, osx = 0;
  if (ZC_LIKELY(bs.have >= (int32_t)tot)) {
    ofx = bs.read(ocode); mlx = bs.read(mlb); llx = bs.read(llb);
    if (!LAST) { lsx = bs.read(lnb); msx = bs.read(mnb); osx = bs.read(onb); }
  } else {
    // more than 56 bits in one sequence (an offset code above ~14 together with long length codes): field by field
#if defined(ZC_TRACE)
    zc_trace_wide();
#endif
    ofx = bs.read(ocode);
    bs.refill_back(w);
    mlx = bs.read(mlb); llx = bs.read(llb);
    bs.refill_back(w);
    if (!LAST) { lsx = bs.read(lnb); msx = bs.read(mnb); osx = bs.read(onb); }
  }
  const uint32_t ov = (1u << ocode) + ofx;
  const uint32_t ml = (uint32_t)(me >> 32) + mlx;
  const uint32_t ll = (uint32_t)(le >> 32) + llx;
  if (!LAST) {
    S.le_raw = w.ll_at(((uint32_t)le & 0xFFFF) + lsx);
    S.me_raw = w.ml_at(((uint32_t)me & 0xFFFF) + msx);
    S.oe_raw = w.of_at((oe & 0xFFFF) + osx);
  }
  if constexpr (W::RESOLVES_OFFSETS) {
    // (the two-wave producer sends the offset VALUE: its consumer keeps the repeat-offset history and checks the reach)
    return w.seq_raw(ll, ov, ml) ? OK : CORRUPT;
  } else {
    // repeat offsets (3.1.1.5) as selects on j = 0..3 (a repeat code, shifted by one when there are no literals) / 4 (a new offset):
    //   j = 0 offset r0, history unchanged; j = 1 r1, swapped to the front; j = 2 r2, j = 3 r0 - 1, j = 4 ov - 3: pushed to the front
    uint32_t j = ov - 1 + (ll == 0 ? 1u : 0u);
    j = ov > 3 ? 4u : j;
    uint32_t off = ov - 3;
    off = j == 0 ? S.r0 : off;
    off = j == 1 ? S.r1 : off;
    off = j == 2 ? S.r2 : off;
    off = j == 3 ? S.r0 - 1 : off;
    S.r2 = j >= 2 ? S.r1 : S.r2;
    S.r1 = j >= 1 ? S.r0 : S.r1;
    S.r0 = off;
    // execute (a W refuses an offset of 0 — r0 - 1 with r0 = 1 — and literals past the block's; it may do so a few sequences late)
    return w.seq(ll, off, ml) ? OK : CORRUPT;
  }
}

// The sequences of one block: the backward bitstream at input [p, p + len), nseq > 0 sequences, accuracy logs packed as
// ll | of << 4 | ml << 8, the repeat offsets in and out. Everything the loop touches goes through W: in8 / in64 / in64_back (the
// bitstream), ll_at / ml_at / of_at (the tables), seq or seq_raw (one decoded sequence).
template <class W>
ZC_FN int seq_loop(W& w, uint32_t p, uint32_t len, uint32_t nseq, uint32_t als, uint32_t& r0_, uint32_t& r1_, uint32_t& r2_) {
  const uint32_t ll_al = als & 15, of_al = (als >> 4) & 15, ml_al = (als >> 8) & 15;
  SeqState<W> S;
  if (!S.bs.init(w, p, len)) return CORRUPT;
  S.bs.refill(w);
  if (S.bs.left() < (int32_t)(ll_al + of_al + ml_al)) return CORRUPT;
  const uint32_t ls = S.bs.read(ll_al), os = S.bs.read(of_al), ms = S.bs.read(ml_al);   // <= 26 bits
  S.r0 = r0_; S.r1 = r1_; S.r2 = r2_;
  S.le_raw = w.ll_at(ls); S.me_raw = w.ml_at(ms); S.oe_raw = w.of_at(os);
  for (uint32_t left = nseq; left > 1; --left) {
    const int rc = seq_one<W, false>(w, S);
    if (rc) return rc;
  }
  const int rc = seq_one<W, true>(w, S);
  if (rc) return rc;
  r0_ = S.r0; r1_ = S.r1; r2_ = S.r2;
  if (S.bs.left() != 0) return CORRUPT_STRICT;   // every bit of the stream belongs to a sequence
  return OK;
}

// one Compressed_Block (3.1.1.2 / 3.1.1.3): input [pos, pos + bsize)
template <class W>
ZC_FN int decode_block(W& w, Frame& F, uint32_t pos, uint32_t bsize) {
  const uint32_t end = pos + bsize;
  if (bsize < 2) return CORRUPT;
  uint32_t ll_al = F.tabs & 15, of_al = (F.tabs >> 4) & 15, ml_al = (F.tabs >> 8) & 15;
  uint32_t ll_tag = (F.tabs >> 12) & 3, of_tag = (F.tabs >> 14) & 3, ml_tag = (F.tabs >> 16) & 3, huf_bits = (F.tabs >> 18) & 15;
  // ---- literals section header
  const uint32_t b0 = w.in8(pos);
  const uint32_t ltype = b0 & 3, sf = (b0 >> 2) & 3;
  uint32_t regen, comp = 0, hdr, streams = 1;
  if (ltype < 2) {
    if ((sf & 1) == 0) { regen = b0 >> 3; hdr = 1; }
    else if (sf == 1) { if (bsize < 2) return CORRUPT; regen = (b0 >> 4) | (w.in8(pos + 1) << 4); hdr = 2; }
    else { if (bsize < 3) return CORRUPT; regen = (b0 >> 4) | (w.in8(pos + 1) << 4) | (w.in8(pos + 2) << 12); hdr = 3; }
  } else {
    if (bsize < 5) return CORRUPT;
    const uint64_t h = w.in64(pos);
    if (sf == 0 || sf == 1) { regen = (uint32_t)(h >> 4) & 0x3FF; comp = (uint32_t)(h >> 14) & 0x3FF; hdr = 3; streams = sf == 0 ? 1 : 4; }
    else if (sf == 2) { regen = (uint32_t)(h >> 4) & 0x3FFF; comp = (uint32_t)(h >> 18) & 0x3FFF; hdr = 4; streams = 4; }
    else { regen = (uint32_t)(h >> 4) & 0x3FFFF; comp = (uint32_t)(h >> 22) & 0x3FFFF; hdr = 5; streams = 4; }
  }
  if (regen > BLOCK_MAX) return CORRUPT;
  uint32_t p = pos + hdr;
  // where the literals are: 0 = input at lit_pos, 1 = page output at lit_pos, 2 = the byte lit_pos repeated
  uint32_t lit_kind, lit_pos;
  if (ltype == 0) {
    if (regen > end - p) return CORRUPT;
    lit_kind = 0; lit_pos = p; p += regen;
  } else if (ltype == 1) {
    if (p >= end) return CORRUPT;
    lit_kind = 2; lit_pos = w.in8(p); p += 1;
  } else {
    if (comp > end - p || comp == 0) return CORRUPT;
    const uint32_t lend = p + comp;
    if (ltype == 2) {
      uint32_t mb = 0;
      const uint32_t used = huf_read_table(w, p, lend, &mb);
      if (used == 0) return CORRUPT;
      huf_bits = mb;
      p += used;
    } else if (huf_bits == 0) {
      return CORRUPT;   // Treeless_Literals_Block with no earlier tree
    }
    // The decoded literals go to the TAIL of this page's output region, [cap - regen, cap): a sequence never writes past the literals
    // it has not consumed yet (what remains to be written is at least what remains of the literals), so the region is free until read.
    if (regen > w.cap() - w.op()) return CORRUPT;
    lit_kind = 1; lit_pos = w.cap() - regen;
    if (regen) {
      // stream k: input [sp + (l1 + .. + l_k-1), + l_k) -> seg symbols (the last stream: what is left of regen)
      uint32_t sp = p, l1 = lend - p, l2 = 0, l3 = 0, l4 = 0, seg = regen;
      if (streams == 4) {
        if (lend - p < 6) return CORRUPT;
        const uint64_t j = w.in64(p);
        l1 = (uint32_t)j & 0xFFFF; l2 = (uint32_t)(j >> 16) & 0xFFFF; l3 = (uint32_t)(j >> 32) & 0xFFFF;
        const uint32_t body = lend - p - 6;
        if (l1 + l2 + l3 > body) return CORRUPT;
        l4 = body - l1 - l2 - l3;
        seg = (regen + 3) >> 2;
        if (3 * seg > regen) return CORRUPT;
        sp = p + 6;
      }
      // (a stream that ends before its last symbol is refused here; libzstd's double-symbol decoder, when its heuristic picks it,
      // lets the LAST symbol of such a stream through from zero bits — malformed by the format either way)
      if (!w.huf_streams(streams, sp, l1, l2, l3, l4, seg, regen, huf_bits, lit_pos)) return CORRUPT_STRICT;
    }
    p = lend;
  }
  // the literals of the block are one forward stream from here on: put_lit(n) moves its next n bytes to the output
  w.lit_begin(lit_kind, lit_pos, regen);
  // ---- sequences section header
  if (p >= end) return CORRUPT;
  uint32_t nseq = w.in8(p);
  if (nseq == 0) {
    p += 1;
  } else if (nseq < 128) {
    p += 1;
  } else if (nseq < 255) {
    if (end - p < 2) return CORRUPT;
    nseq = ((nseq - 128) << 8) + w.in8(p + 1);
    p += 2;
  } else {
    if (end - p < 3) return CORRUPT;
    nseq = w.in8(p + 1) + (w.in8(p + 2) << 8) + 0x7F00;
    p += 3;
  }
  if (nseq) {
    if (p >= end) return CORRUPT;
    const uint32_t modes = w.in8(p);
    p += 1;
    if (modes & 3) return CORRUPT_STRICT;   // Reserved bits (libzstd >= 1.5.6)
    // (three explicit calls, the fields passed by reference: a loop over pointers to F's members would put F into private memory,
    // whose loads the compiler treats as per-lane values — the whole parser left the scalar unit that way in the first build)
    int trc = seq_table<W, K_LL>(w, (modes >> 6) & 3, ll_al, ll_tag, p, end);
    if (trc) return trc;
    trc = seq_table<W, K_OF>(w, (modes >> 4) & 3, of_al, of_tag, p, end);
    if (trc) return trc;
    trc = seq_table<W, K_ML>(w, (modes >> 2) & 3, ml_al, ml_tag, p, end);
    if (trc) return trc;
    // ---- the sequences: one backward bitstream, walked by W::sequences (= seq_loop above; the two-wave producer runs it as a
    // function of its own so that the loop gets a register allocation of its own)
    F.tabs = tabs_pack(ll_al, of_al, ml_al, ll_tag, of_tag, ml_tag, huf_bits);
    if (p >= end) return CORRUPT;
    uint32_t r0 = F.rep[0], r1 = F.rep[1], r2 = F.rep[2];
    trc = w.sequences(p, end - p, nseq, ll_al | (of_al << 4) | (ml_al << 8), r0, r1, r2);
    if (trc) return trc;
    F.rep[0] = r0; F.rep[1] = r1; F.rep[2] = r2;
  }
  else {
    F.tabs = tabs_pack(ll_al, of_al, ml_al, ll_tag, of_tag, ml_tag, huf_bits);   // (a block without sequences may still bring a Huffman table)
  }
  const uint32_t lit_left = w.lit_rest();
  if (lit_left && !w.put_lit(lit_left)) return CORRUPT;
  return OK;
}

// the whole compressed payload of a page: one or more frames (3.1), skippable frames skipped (3.1.2). The output must come to exactly
// w.cap() bytes (the caller checks op == cap).
template <class W>
ZC_FN int decode_frames(W& w, uint32_t in_len_) {
  uint32_t in_len = in_len_;
  uint32_t p = 0;
  while (p < in_len) {
    if (in_len - p < 4) return CORRUPT;
    const uint32_t magic = (uint32_t)w.in64(p);
    if ((magic & 0xFFFFFFF0u) == 0x184D2A50u) {
      if (in_len - p < 8) return CORRUPT;
      const uint32_t sz = (uint32_t)(w.in64(p) >> 32);
      if (sz > in_len - p - 8) return CORRUPT;
      p += 8 + sz;
      continue;
    }
    if (magic != 0xFD2FB528u) return CORRUPT;
    p += 4;
    if (p >= in_len) return CORRUPT;
    const uint32_t fhd = w.in8(p);
    p += 1;
    const uint32_t fcs_flag = fhd >> 6, single = (fhd >> 5) & 1, did = fhd & 3;
    uint32_t checksum = (fhd >> 2) & 1;
    if (fhd & 0x08) return CORRUPT;   // reserved bit
    uint64_t window = 0;
    if (!single) {
      if (p >= in_len) return CORRUPT;
      const uint32_t wd = w.in8(p);
      p += 1;
      const uint32_t e = 10 + (wd >> 3), m = wd & 7;
      if (e > 31) return CORRUPT;   // (libzstd: frameParameter_windowTooLarge above 2^31)
      window = ((uint64_t)1 << e) + (((uint64_t)1 << e) >> 3) * m;
    }
    if (did) {
      const uint32_t n = did == 3 ? 4 : did;
      if (in_len - p < n) return CORRUPT;
      uint32_t id = 0;
      for (uint32_t k = 0; k < n; ++k) id |= w.in8(p + k) << (8 * k);
      p += n;
      if (id != 0) return UNSUPPORTED;   // a dictionary: the reference's writer never uses one
    }
    uint32_t fn = fcs_flag == 0 ? (single ? 1u : 0u) : fcs_flag == 1 ? 2u : fcs_flag == 2 ? 4u : 8u;
    uint64_t fcs = 0;
    if (fn) {
      if (in_len - p < fn) return CORRUPT;
      for (uint32_t k = 0; k < fn; ++k) fcs |= (uint64_t)w.in8(p + k) << (8 * k);
      if (fn == 2) fcs += 256;
      p += fn;
    }
    if (single) window = fcs;
    uint32_t frame_start = w.op();
    if (fn && fcs > (uint64_t)(w.cap() - w.op())) return CORRUPT;
    w.frame_begin();   // back-references do not reach across frames
    Frame F;
    F.rep[0] = 1; F.rep[1] = 4; F.rep[2] = 8;
    F.tabs = 0;   // no tables yet (TAG_NONE)
    uint32_t bmax = window < BLOCK_MAX ? (uint32_t)window : BLOCK_MAX;
    uint32_t fcs32 = (uint32_t)fcs;   // (checked against the room of the page above: it fits)
    for (;;) {
      if (in_len - p < 3) return CORRUPT;
      const uint32_t bh = (uint32_t)w.in64(p) & 0xFFFFFF;
      p += 3;
      const uint32_t btype = (bh >> 1) & 3;
      uint32_t last = bh & 1, bsize = bh >> 3;
      if (btype == 3) return CORRUPT;
      if (btype == 1) {
        if (p >= in_len) return CORRUPT;
        if (bsize > bmax) return CORRUPT;
        if (bsize && !w.put_fill(w.in8(p), bsize)) return CORRUPT;
        p += 1;
      } else {
        if (bsize > in_len - p || bsize > BLOCK_MAX) return CORRUPT;
        if (btype == 0) {
          if (bsize > bmax) return CORRUPT;
          if (bsize && !w.put_in(p, bsize)) return CORRUPT;
        } else {
          // (what this loop keeps between blocks waits in W's parking slots while the block is decoded: the sequence loop needs
          //  every scalar register it can get, and a value that is only STORED and re-LOADED around it is not live inside it)
          w.park(0, p); w.park(1, in_len); w.park(2, bsize); w.park(3, bmax); w.park(4, fcs32); w.park(5, frame_start);
          w.park(6, fn | (checksum << 4) | (last << 5)); w.park(7, w.op());
          const int rc = decode_block(w, F, p, bsize);
          if (rc) return rc;
          p = w.unpark(0); in_len = w.unpark(1); bsize = w.unpark(2); bmax = w.unpark(3); fcs32 = w.unpark(4); frame_start = w.unpark(5);
          const uint32_t fl = w.unpark(6);
          fn = fl & 15; checksum = (fl >> 4) & 1; last = fl >> 5;
          if (w.op() - w.unpark(7) > bmax) return CORRUPT;
        }
        p += bsize;
      }
      if (last) break;
    }
    if (fn && w.op() - frame_start != fcs32) return CORRUPT;
    if (checksum) {
      if (in_len - p < 4) return CORRUPT;
      p += 4;   // XXH64 of the content: not verified here (the parquet page's own size check stands in)
    }
  }
  return OK;
}

}  // namespace zc

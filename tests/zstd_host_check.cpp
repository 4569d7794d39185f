// zstd_host_check.cpp — TEST INFRASTRUCTURE: runs databend_amd/csrc/zstd_core.h (the frame parser the device kernel instantiates) over
// plain host memory and compares it with the system's libzstd (dlopen, the library behind the reference's `zstd` crate).
//   zstd_host_check <seed> <cases>        generated inputs x compression levels, both directions; prints "ok <n>" or the first mismatch
//   zstd_host_check file <frames> <decoded size> [<out>]   one decode of a file (the reference-held frames of tests/golden/zstd_ref)
// W here is the host stand-in of the wave: the same four put_* primitives, bounds-checked, no LDS.
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <random>
#include <string>
#include <vector>

#define ZC_TRACE 1
static int g_trace_line = 0, g_trace_strict = 0;
static void zc_trace(int line, int strict) { if (!g_trace_line) { g_trace_line = line; g_trace_strict = strict; } }
static long g_wide_sequences = 0;   // sequences that took the field-by-field path of the bit reader (more than 56 bits in one sequence)
static void zc_trace_wide() { ++g_wide_sequences; }
#include "../databend_amd/csrc/zstd_core.h"

struct HostWave {
  static constexpr bool RESOLVES_OFFSETS = false;
  bool seq_raw(uint32_t, uint32_t, uint32_t) { return false; }
  const uint8_t* in;
  uint32_t in_len;
  std::vector<uint8_t> out;
  uint32_t op_ = 0, cap_ = 0, frame0 = 0;
  uint16_t huf_[2048];
  uint64_t llt_[512], mlt_[512];
  uint32_t oft_[256];
  uint8_t scr_[zc::SCR_BYTES];

  bool lead() const { return true; }
  void sync() {}
  bool bcast(bool b) const { return b; }
  uint32_t uni(uint32_t v) const { return v; }
  uint64_t uni64(uint64_t v) const { return v; }
  uint16_t* huf() { return huf_; }
  uint64_t* llt() { return llt_; }
  uint64_t* mlt() { return mlt_; }
  uint32_t* oft() { return oft_; }
  uint64_t ll_at(uint32_t st) const { return llt_[st]; }
  uint64_t ml_at(uint32_t st) const { return mlt_[st]; }
  uint32_t of_at(uint32_t st) const { return oft_[st]; }
  uint8_t* scr() { return scr_; }
  int sequences(uint32_t p, uint32_t len, uint32_t nseq, uint32_t als, uint32_t& r0, uint32_t& r1, uint32_t& r2) {
    return zc::seq_loop(*this, p, len, nseq, als, r0, r1, r2);
  }
  uint32_t parked[8];
  void park(uint32_t i, uint32_t v) { parked[i] = v; }
  uint32_t unpark(uint32_t i) const { return parked[i]; }
  uint32_t op() const { return op_; }
  uint32_t cap() const { return cap_; }
  void frame_begin() { frame0 = op_; }
  uint32_t in8(uint32_t p) const { return p < in_len ? in[p] : (abort(), 0); }
  uint64_t in64(uint32_t p) const {
    if (p >= in_len) abort();   // the parser only starts a read inside the input
    uint64_t v = 0;
    for (uint32_t k = 0; k < 8 && p + k < in_len; ++k) v |= (uint64_t)in[p + k] << (8 * k);
    return v;
  }
  uint32_t lane_in8(uint32_t p) const { return in8(p); }
  uint64_t lane_in64(uint32_t p) const { return in64(p); }
  uint64_t in64_back(uint32_t p) const { return in64(p); }
  void lane_store(uint32_t p, uint8_t b) {
    if (p >= cap_) abort();
    out[p] = b;
  }
  bool put_in(uint32_t pos, uint32_t len) {
    if (len > cap_ - op_ || pos > in_len || len > in_len - pos) return false;
    memcpy(out.data() + op_, in + pos, len);
    op_ += len;
    return true;
  }
  bool put_out(uint32_t pos, uint32_t len) {
    if (len > cap_ - op_ || pos > cap_ || len > cap_ - pos) return false;
    memmove(out.data() + op_, out.data() + pos, len);
    op_ += len;
    return true;
  }
  bool put_fill(uint32_t byte, uint32_t len) {
    if (len > cap_ - op_) return false;
    memset(out.data() + op_, (int)byte, len);
    op_ += len;
    return true;
  }
  uint32_t lkind = 0, lpos = 0, lleft = 0;
  void lit_begin(uint32_t kind, uint32_t pos, uint32_t n) { lkind = kind; lpos = pos; lleft = n; }
  bool put_lit(uint32_t len) {
    if (len > lleft) return false;
    lleft -= len;
    if (lkind == 2) return put_fill(lpos, len);
    const bool ok = lkind == 0 ? put_in(lpos, len) : put_out(lpos, len);
    lpos += len;
    return ok;
  }
  bool seq(uint32_t ll, uint32_t off, uint32_t ml) { return put_lit(ll) && put_match(off, ml); }
  uint32_t lit_rest() const { return lleft; }
  bool put_match(uint32_t off, uint32_t len) {
    if (off == 0 || off > op_ - frame0 || len > cap_ - op_) return false;
    for (uint32_t i = 0; i < len; ++i) out[op_ + i] = out[op_ + i - off];
    op_ += len;
    return true;
  }
  bool huf_streams(uint32_t streams, uint32_t sp, uint32_t l1, uint32_t l2, uint32_t l3, uint32_t l4, uint32_t seg, uint32_t regen,
                   uint32_t maxbits, uint32_t outp) {
    if (streams == 1) return zc::huf_stream(*this, huf_, maxbits, sp, l1, regen, outp);
    const uint32_t pos[4] = {sp, sp + l1, sp + l1 + l2, sp + l1 + l2 + l3}, len[4] = {l1, l2, l3, l4};
    bool ok = true;
    for (int k = 0; k < 4; ++k) ok &= zc::huf_stream(*this, huf_, maxbits, pos[k], len[k], k < 3 ? seg : regen - 3 * seg, outp + k * seg);
    return ok;
  }
};

// -> zc status; out = decoded bytes when OK (and exactly cap of them)
static int decode(const std::vector<uint8_t>& comp, uint32_t cap, std::vector<uint8_t>& out) {
  HostWave w;
  g_trace_line = 0;
  w.in = comp.data();
  w.in_len = (uint32_t)comp.size();
  w.out.assign(cap + 16, 0xEE);
  w.cap_ = cap;
  int rc = zc::decode_frames(w, w.in_len);
  if (rc == zc::OK && w.op_ != cap) rc = zc::CORRUPT_;
  w.out.resize(cap);
  out.swap(w.out);
  return rc;
}

typedef size_t (*compress_fn)(void*, size_t, const void*, size_t, int);
typedef size_t (*bound_fn)(size_t);
typedef size_t (*decompress_fn)(void*, size_t, const void*, size_t);
typedef unsigned (*iserror_fn)(size_t);

static std::vector<uint8_t> gen(std::mt19937_64& r, int kind, size_t n) {
  std::vector<uint8_t> v(n);
  switch (kind) {
    case 0: for (auto& b : v) b = (uint8_t)r(); break;                                   // incompressible
    case 1: break;                                                                       // zeros
    case 2: for (size_t i = 0; i < n; ++i) v[i] = (uint8_t)("abcdefgh"[i & 7]); break;     // short period
    case 3: {                                                                            // int64 prices (3 random bytes, 5 zero)
      for (size_t i = 0; i + 8 <= n; i += 8) { uint64_t x = 90000 + r() % 10404951; memcpy(&v[i], &x, 8); }
      break;
    }
    case 4: {                                                                            // skewed symbols (Huffman-friendly, few matches)
      for (auto& b : v) { uint64_t x = r(); b = (uint8_t)(__builtin_ctzll(x | (1ull << 40)) * 7 + (x >> 60)); }
      break;
    }
    case 5: {                                                                            // words from a small vocabulary
      static const char* W[] = {"lineitem", "orders", "DELIVER IN PERSON", "TRUCK", "AIR", "RAIL", "furiously ", "quickly ", "packages ", "1996-03-13"};
      size_t i = 0;
      while (i < n) { const char* s = W[r() % 10]; for (; *s && i < n; ++s) v[i++] = (uint8_t)*s; }
      break;
    }
    case 6: {                                                                            // long-distance repeats
      const size_t blk = 1 + r() % 5000;
      for (size_t i = 0; i < n; ++i) v[i] = i < blk ? (uint8_t)r() : v[i - blk];
      for (size_t k = 0; k < n / 97; ++k) v[r() % n] ^= 1;
      break;
    }
    case 7: {                                                                            // 2-bit packed dictionary indices
      for (auto& b : v) { uint64_t x = r(); b = (uint8_t)((x & 3) | ((x >> 2 & 3) << 2) | ((x >> 4 & 1) << 4) | ((x >> 5 & 1) << 6)); }
      break;
    }
    default: {                                                                           // runs of random length
      size_t i = 0;
      while (i < n) { size_t len = 1 + r() % 300; uint8_t b = (uint8_t)r(); for (; len && i < n; --len) v[i++] = b; }
    }
  }
  return v;
}

int main(int argc, char** argv) {
  if (argc > 3 && !strcmp(argv[1], "file")) {   // zstd_host_check file <frame file> <decoded size>: one decode, status + first failing check
    FILE* f = fopen(argv[2], "rb");
    if (!f) return 2;
    std::vector<uint8_t> z(1 << 26), out;
    z.resize(fread(z.data(), 1, z.size(), f));
    fclose(f);
    const int rc = decode(z, (uint32_t)atoi(argv[3]), out);
    printf("rc %d zstd_core.h:%d\n", rc, g_trace_line);
    if (rc == 0 && argc > 4) {   // the decoded bytes, for the caller to hash
      FILE* o = fopen(argv[4], "wb");
      if (!o) return 2;
      fwrite(out.data(), 1, out.size(), o);
      fclose(o);
    }
    return rc;
  }
  const uint64_t seed = argc > 1 ? strtoull(argv[1], 0, 10) : 1;
  const int cases = argc > 2 ? atoi(argv[2]) : 200;
  void* h = dlopen("libzstd.so.1", RTLD_NOW | RTLD_LOCAL);
  if (!h) { printf("skip: no libzstd.so.1\n"); return 0; }
  compress_fn comp = (compress_fn)dlsym(h, "ZSTD_compress");
  bound_fn bound = (bound_fn)dlsym(h, "ZSTD_compressBound");
  decompress_fn dec = (decompress_fn)dlsym(h, "ZSTD_decompress");
  iserror_fn iserr = (iserror_fn)dlsym(h, "ZSTD_isError");
  if (!comp || !bound || !dec || !iserr) { printf("skip: libzstd symbols\n"); return 0; }
  std::mt19937_64 r(seed);
  static const int levels[] = {1, 1, 3, 5, 9, 15, 19, -1, -5};
  static const size_t sizes[] = {0, 1, 2, 3, 7, 64, 255, 256, 257, 1000, 4096, 20000, 65536, 131072, 131073, 300000, 1048576, 2500000};
  int done = 0, mutated_ok = 0, mutated_rej = 0, stricter = 0;
  for (int c = 0; c < cases; ++c) {
    const int kind = (int)(r() % 9);
    size_t n = sizes[r() % (sizeof(sizes) / sizeof(sizes[0]))];
    if (r() % 3 == 0) n = r() % 200000;
    if (c % 16 != 0 && n > 400000) n = 100000 + r() % 100000;   // (a few big ones only)
    std::vector<uint8_t> src = gen(r, kind, n);
    const int level = levels[r() % 9];
    std::vector<uint8_t> z(bound(n) + 64);
    size_t zn = comp(z.data(), z.size(), src.data(), n, level);
    if (iserr(zn)) { printf("compress failed\n"); return 1; }
    z.resize(zn);
    if (r() % 5 == 0 && n > 10) {   // two frames back to back
      const size_t cut = 1 + r() % (n - 1);
      std::vector<uint8_t> a(bound(cut) + 64), b(bound(n - cut) + 64);
      const size_t an = comp(a.data(), a.size(), src.data(), cut, level), bn = comp(b.data(), b.size(), src.data() + cut, n - cut, level);
      z.assign(a.begin(), a.begin() + an);
      z.insert(z.end(), b.begin(), b.begin() + bn);
    }
    std::vector<uint8_t> out;
    const int rc = decode(z, (uint32_t)n, out);
    if (rc != zc::OK || out != src) {
      size_t at = 0;
      while (rc == zc::OK && at < n && out[at] == src[at]) ++at;
      printf("MISMATCH case %d kind %d n %zu level %d zn %zu rc %d (zstd_core.h:%d) first bad byte %zu\n", c, kind, n, level, z.size(), rc, g_trace_line, at);
      return 1;
    }
    ++done;
    // a wrong declared size is an error, never an overrun
    if (n > 0) {
      std::vector<uint8_t> o2;
      if (decode(z, (uint32_t)n - 1, o2) == zc::OK) { printf("short cap accepted, case %d\n", c); return 1; }
      if (decode(z, (uint32_t)n + 1, o2) == zc::OK) { printf("long cap accepted, case %d\n", c); return 1; }
    }
    // mutations: whatever libzstd makes of the bytes, the parser agrees whenever libzstd yields exactly n bytes, and never faults
    for (int m = 0; m < 6 && z.size() > 6; ++m) {
      std::vector<uint8_t> zm = z;
      const int how = (int)(r() % 3);
      if (how == 0) zm[r() % zm.size()] ^= (uint8_t)(1u << (r() % 8));
      else if (how == 1) zm.resize(1 + r() % zm.size());
      else zm[4 + r() % (zm.size() - 4)] = (uint8_t)r();
      std::vector<uint8_t> ref(n + 1), o3;
      const size_t rn = dec(ref.data(), n, zm.data(), zm.size());
      const int rc3 = decode(zm, (uint32_t)n, o3);
      if (!iserr(rn) && rn == n) {
        ref.resize(n);
        // (a frame with a content checksum is verified by libzstd and not here: none of these frames carries one)
        if (rc3 != zc::OK && g_trace_strict) {
          ++stricter;   // a check of the format that this libzstd version does not make
        } else if (rc3 != zc::OK || o3 != ref) {
          if (FILE* f = fopen("/tmp/zfail.bin", "wb")) { fwrite(zm.data(), 1, zm.size(), f); fclose(f); }
          printf("n = %zu\n", n); printf("MUTATION: libzstd accepts, parser rc %d at zstd_core.h:%d (case %d m %d how %d)\n", rc3, g_trace_line, c, m, how); return 1; }
        ++mutated_ok;
      } else {
        if (rc3 == zc::OK) {
          // libzstd refused or produced another size; accepting is only legitimate if libzstd's refusal is about something this
          // parser does not check. None is known: report it.
          printf("MUTATION: libzstd refuses (%zu), parser accepts (case %d m %d how %d)\n", rn, c, m, how);
          return 1;
        }
        ++mutated_rej;
      }
    }
  }
  // sequences wider than the bit register (a far offset + two long length codes + the state updates): none of the generators above
  // makes one. 1.3 MB of noise, then blocks of 40,000 new bytes followed by 40,000 bytes copied from ~1.3 MB back, at level 19.
  {
    const size_t head = 1300000, piece = 40000, n = head + 12 * 2 * piece;
    std::vector<uint8_t> src(n);
    for (size_t i = 0; i < head; ++i) src[i] = (uint8_t)r();
    for (size_t i = head; i < n; i += 2 * piece) {
      for (size_t k = 0; k < piece; ++k) src[i + k] = (uint8_t)r();
      memcpy(&src[i + piece], &src[i + piece - head], piece);
    }
    std::vector<uint8_t> z(bound(n) + 64), out;
    const size_t zn = comp(z.data(), z.size(), src.data(), n, 19);
    if (iserr(zn)) { printf("compress failed\n"); return 1; }
    z.resize(zn);
    const long before = g_wide_sequences;
    const int rc = decode(z, (uint32_t)n, out);
    if (rc != zc::OK || out != src) { printf("MISMATCH wide-sequence case rc %d (zstd_core.h:%d)\n", rc, g_trace_line); return 1; }
    if (g_wide_sequences == before) { printf("the wide-sequence case did not reach the field-by-field path\n"); return 1; }
    for (int m = 0; m < 40; ++m) {   // and its mutants
      std::vector<uint8_t> zm = z, ref(n + 1), o3;
      zm[zm.size() - 1 - r() % 4000] ^= (uint8_t)(1u << (r() % 8));
      const size_t rn = dec(ref.data(), n, zm.data(), zm.size());
      const int rc3 = decode(zm, (uint32_t)n, o3);
      ref.resize(n);
      if (!iserr(rn) && rn == n) {
        if (rc3 != zc::OK && g_trace_strict) ++stricter;
        else if (rc3 != zc::OK || o3 != ref) { printf("MUTATION (wide): libzstd accepts, parser rc %d at zstd_core.h:%d\n", rc3, g_trace_line); return 1; }
        else ++mutated_ok;
      } else {
        if (rc3 == zc::OK) { printf("MUTATION (wide): libzstd refuses, parser accepts\n"); return 1; }
        ++mutated_rej;
      }
    }
    ++done;
  }
  printf("ok %d cases (%ld wide sequences), mutations: %d agreed-accept %d agreed-reject %d rejected-by-format-checks-this-libzstd-lacks\n", done, g_wide_sequences, mutated_ok, mutated_rej, stricter);
  return 0;
}

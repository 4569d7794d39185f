// dv_wave.h — one wave decompresses one page (k_parquet_dev.hip): how the wave reads its input and where its output lives.
//
//   input    look-ahead WINDOWS of the compressed payload held in VGPRs (lane l = dword l): headers, FSE / Huffman table descriptions
//            and the backward bitstream of the ZSTD sequences are read with v_readlane (scalar results, no memory round trip per
//            field); literal bytes are moved from a window to the output with ds_bpermute. A window slides with one coalesced load;
//            the FORWARD streams (the literals of a ZSTD block, the whole LZ4 / Snappy payload) keep the next 256 bytes loaded ahead,
//            so a sequence never waits for HBM.
//   output   an LDS RING of the last W bytes (8 KiB; ZSTD's tables take another 14 KiB of LDS)
//            written to the HBM image in 16-byte stores; a back-reference that reaches further than the ring reads the image itself —
//            those bytes left the ring, so they were flushed long ago (the wave waits for its own stores once per far reference).
// 22 / 8 KiB of LDS per wave: seven / twenty pages resident per CU (the round-4 kernel kept a 64 KiB window + 8 KiB input ring: two).
#pragma once
#include "zstd_core.h"

struct DvJob {
  const uint8_t* src;   // the page's payload as stored
  uint8_t* dst;         // where its decompressed image goes
  uint32_t comp_len, uncomp_len;
  uint32_t lev_len;     // DATA_PAGE_V2: uncompressed level bytes in front
  uint32_t compressed;
  uint32_t src_safe;    // bytes readable from src (up to the 16-byte boundary past the end of the chunk)
  uint32_t codec;
  uint32_t* ctl;        // the chunk's control words (first failure wins)
};

namespace {

// global-memory accesses through pointers that came out of a struct (the compiler would use flat instructions, which tie up the LDS
// counter as well)
typedef const __attribute__((address_space(1))) uint8_t* gcptr8;
__device__ __forceinline__ uint32_t gload32(const uint8_t* p) { return *(const __attribute__((address_space(1))) uint32_t*)(uintptr_t)p; }
__device__ __forceinline__ uint8_t gload8(const uint8_t* p) { return *(const __attribute__((address_space(1))) uint8_t*)(uintptr_t)p; }
__device__ __forceinline__ uint64_t gload64u(const uint8_t* p) {   // unaligned
  uint64_t v;
  __builtin_memcpy(&v, (const __attribute__((address_space(1))) uint8_t*)(uintptr_t)p, 8);
  return v;
}
__device__ __forceinline__ void gstore8(uint8_t* p, uint8_t v) { *(__attribute__((address_space(1))) uint8_t*)(uintptr_t)p = v; }
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void gstore128(uint8_t* p, u32x4 v) { *(__attribute__((address_space(1))) u32x4*)(uintptr_t)p = v; }

// Divergence hygiene: the parser state of a wave (positions, counters, window origins) is SCALAR. The compiler keeps it on the scalar unit
// only while no scalar variable is assigned inside a region whose join coincides with the join of a per-lane branch — one guarded
// per-lane load inside a uniform `if` turned `wlo` into a vector PHI and with it every header field and branch of the kernel (464
// exec-mask branches in the first build). Hence: guarded loads are branch-free (clamped address + select), and scalar updates happen in
// straight-line code after the per-lane part.
// dword at gA + o, or (past the readable region) the last readable dword again; safe is a multiple of 4 and >= 4. No select on the loaded
// value: it would make the wave wait for the load where it is ISSUED, and the chunk loaded ahead would not be ahead any more. What lies
// past the payload is never interpreted as anything but bytes of a (then malformed) stream.
__device__ __forceinline__ uint32_t guarded32(const uint8_t* gA, uint32_t o, uint32_t safe) {
  return gload32(gA + (o + 4 <= safe ? o : safe - 4));
}

__device__ __forceinline__ uint32_t rfl(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ uint32_t rdl(uint32_t v, uint32_t l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)l); }
// v of lane `l` (per-lane l; only its low six bits count)
__device__ __forceinline__ uint32_t bperm(uint32_t l, uint32_t v) { return (uint32_t)__builtin_amdgcn_ds_bpermute((int)(l << 2), (int)v); }

// A FORWARD stream of global bytes read through two 256-byte chunks held in registers (chunk k = bytes [256 k, 256 k + 256) from the
// 4-byte aligned address gA; the chunk after it is loaded when the stream enters chunk k, i.e. one chunk ahead of its use).
struct FwdStream {
  const uint8_t* gA;
  uint32_t safe;       // bytes readable from gA (a multiple of 4)
  uint32_t k;          // chunk held in `cur`
  uint32_t cur, nxt;   // this lane's dword of chunk k / k + 1
  uint32_t lane;
  __device__ __forceinline__ uint32_t fetch(uint32_t chunk) const {
    return guarded32(gA, chunk * 256 + 4 * lane, safe);
  }
  __device__ __forceinline__ void open(const uint8_t* g, uint32_t safe_bytes, uint32_t a, uint32_t ln) {
    gA = g; safe = safe_bytes & ~3u; lane = ln;
    k = rfl(a >> 8);
    cur = fetch(k); nxt = fetch(k + 1);
  }
  // make byte offset a (from gA) lie in chunk k
  __device__ __forceinline__ void seek(uint32_t a) {
    const uint32_t want = rfl(a >> 8);
    if (want != k) {
      if (want == k + 1) { cur = nxt; nxt = fetch(want + 1); }
      else { cur = fetch(want); nxt = fetch(want + 1); }
    }
    k = want;
  }
  // 8 bytes at offset a (scalar). The chunk loaded ahead is only touched when the read reaches into it (its load may still be in flight).
  __device__ __forceinline__ uint64_t u64(uint32_t a) {
    seek(a);
    const uint32_t i = rfl((a & 255) >> 2);
    uint32_t d0, d1, d2;
    if (i < 62) {
      d0 = rdl(cur, i); d1 = rdl(cur, i + 1); d2 = rdl(cur, i + 2);
    } else {
      d0 = rdl(cur, i);
      d1 = i == 62 ? rdl(cur, 63) : rdl(nxt, 0);
      d2 = rdl(nxt, i - 62);
    }
    const uint64_t lo = (uint64_t)d0 | ((uint64_t)d1 << 32);
    const uint32_t s = 8 * (a & 3);
    return s ? (lo >> s) | ((uint64_t)d2 << (64 - s)) : lo;
  }
  // byte a + li of the stream for a per-lane li < 64 (a in chunk k after seek(a); the bytes asked for lie in [a, a + n), which may reach
  // into chunk k + 1)
  __device__ __forceinline__ uint32_t lane_byte_at(uint32_t a, uint32_t li, uint32_t n) const {
    const uint32_t t = (a & 255) + li;                        // 0..318
    const uint32_t sel = t & 0xFCu;                           // (dword index within its chunk) * 4
    uint32_t v = (uint32_t)__builtin_amdgcn_ds_bpermute((int)sel, (int)cur);
    if ((a & 255) + n > 256) {
      const uint32_t x = (uint32_t)__builtin_amdgcn_ds_bpermute((int)sel, (int)nxt);
      v = t < 256 ? v : x;
    }
    return (v >> (8 * (t & 3))) & 0xFFu;
  }
  // byte a + lane for the first n lanes (a in chunk k after seek(a); a + n may reach into chunk k + 1)
  __device__ __forceinline__ uint8_t lane_byte(uint32_t a, uint32_t n) const {
    const uint32_t t = (a & 255) + lane;                      // 0..318
    const uint32_t idx = (t >> 2) & 63;
    uint32_t v = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(idx << 2), (int)cur);
    if ((a & 255) + n > 256) {
      const uint32_t x = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(idx << 2), (int)nxt);
      v = t < 256 ? v : x;
    }
    return (uint8_t)(v >> (8 * (t & 3)));
  }
};

struct ZWave {
  // ---- input
  const uint8_t* srcA;   // 4-byte aligned address at or below the payload
  uint32_t a0;           // payload start - srcA
  uint32_t in_len;
  uint32_t safeA;        // bytes readable from srcA, a multiple of 4
  uint32_t la, wlo;      // the window: this lane holds bytes [wlo + 4 lane, + 4) from srcA; wlo is a multiple of 4
  // ---- the literal stream of the current block (ZSTD)
  FwdStream lit;
  uint32_t lit_kind, lit_at, lit_left;   // kind 2: lit_at is the byte; else offset of the next literal from lit.gA
  // ---- output
  uint8_t* dst;
  uint32_t cap_, op_, flushed, sh, frame0;
  uint8_t* win;
  uint32_t WM, WF;       // ring mask / flush threshold
  bool stores_pending;   // ring bytes were stored to the image since the wave last waited for its stores
  // ---- LDS tables
  uint8_t* tab;
  uint32_t lane;

  static constexpr bool RESOLVES_OFFSETS = false;   // zstd_core.h hands seq() the resolved offset
  __device__ __forceinline__ bool seq_raw(uint32_t, uint32_t, uint32_t) { return false; }
  __device__ __forceinline__ bool lead() const { return lane == 0; }
  __device__ __forceinline__ void sync() const { __builtin_amdgcn_wave_barrier(); }
  __device__ __forceinline__ bool bcast(bool b) const { return __builtin_amdgcn_readfirstlane((int)b) != 0; }
  __device__ __forceinline__ uint32_t uni(uint32_t v) const { return rfl(v); }
  __device__ __forceinline__ uint64_t uni64(uint64_t v) const { return (uint64_t)rfl((uint32_t)v) | ((uint64_t)rfl((uint32_t)(v >> 32)) << 32); }
  __device__ __forceinline__ uint16_t* huf() const { return (uint16_t*)tab; }
  __device__ __forceinline__ uint64_t* llt() const { return (uint64_t*)(tab + 4096); }
  __device__ __forceinline__ uint64_t* mlt() const { return (uint64_t*)(tab + 8192); }
  __device__ __forceinline__ uint32_t* oft() const { return (uint32_t*)(tab + 12288); }
  __device__ __forceinline__ uint8_t* scr() const { return tab + 13312; }
  // entry of a sequence table at a (uniform) state number
  __device__ __forceinline__ uint64_t ll_at(uint32_t st) const { return llt()[st]; }
  __device__ __forceinline__ uint64_t ml_at(uint32_t st) const { return mlt()[st]; }
  __device__ __forceinline__ uint32_t of_at(uint32_t st) const { return oft()[st]; }
  // parking slots (zstd_core.h decode_frames): stored by lane 0, read back by the wave; volatile so that the value is really re-loaded
  __device__ __forceinline__ void park(uint32_t i, uint32_t v) const { if (lane == 0) ((volatile uint32_t*)(scr() + zc::SCR_PARK))[i] = v; }
  __device__ __forceinline__ uint32_t unpark(uint32_t i) const { return rfl(((volatile uint32_t*)(scr() + zc::SCR_PARK))[i]); }
  __device__ __forceinline__ uint32_t op() const { return op_; }
  __device__ __forceinline__ uint32_t cap() const { return cap_; }
  __device__ __forceinline__ void frame_begin() { frame0 = op_; }

  // ---- the window (headers, table descriptions, backward bitstreams)
  __device__ __forceinline__ void slide(uint32_t wl) {
    wlo = rfl(wl);
    la = guarded32(srcA, wlo + 4 * lane, safeA);
  }
  // bytes [a, a + n) from srcA inside the window (n <= 8)
  __device__ __forceinline__ void ensure(uint32_t a, uint32_t n) {
    if (a < wlo) {
      const uint32_t e = (a + n + 3) & ~3u;   // a reader that walks backward: the window ends just past what it asks for
      slide(e > 256 ? e - 256 : 0);
    } else if (a + n > wlo + 256) {
      slide(a & ~3u);
    }
  }
  __device__ __forceinline__ uint32_t in8(uint32_t pos) {
    const uint32_t a = rfl(pos + a0);
    ensure(a, 1);
    return (rdl(la, (a - wlo) >> 2) >> (8 * (a & 3))) & 0xFF;
  }
  __device__ __forceinline__ uint64_t in64(uint32_t pos) {
    const uint32_t a = rfl(pos + a0);
    ensure(a, 8);
    const uint32_t i = (a - wlo) >> 2;
    const uint32_t d0 = rdl(la, i), d1 = rdl(la, i + 1), d2 = rdl(la, i + 2 < 64 ? i + 2 : 63);
    const uint64_t lo = (uint64_t)d0 | ((uint64_t)d1 << 32);
    const uint32_t s = 8 * (a & 3);
    return s ? (lo >> s) | ((uint64_t)d2 << (64 - s)) : lo;
  }
  // in64 for a reader that only moves DOWN from a position the window already covers (the sequence bitstream after its first
  // refill): the window's upper end needs no test
  __device__ __forceinline__ uint64_t in64_back(uint32_t pos) {
    const uint32_t a = rfl(pos + a0);
    if (a < wlo) {
      const uint32_t e = (a + 8 + 3) & ~3u;
      slide(e > 256 ? e - 256 : 0);
    }
    const uint32_t i = (a - wlo) >> 2;
    const uint32_t d0 = rdl(la, i), d1 = rdl(la, i + 1), d2 = rdl(la, i + 2 < 64 ? i + 2 : 63);
    const uint64_t lo = (uint64_t)d0 | ((uint64_t)d1 << 32);
    const uint32_t s = 8 * (a & 3);
    return s ? (lo >> s) | ((uint64_t)d2 << (64 - s)) : lo;
  }
  // per-lane reads of the payload (the Huffman streams: up to four lanes, each at its own address)
  __device__ __forceinline__ uint32_t lane_in8(uint32_t pos) const { return pos + a0 < safeA ? gload8(srcA + a0 + pos) : 0u; }
  __device__ __forceinline__ uint64_t lane_in64(uint32_t pos) const {
    const uint8_t* p = srcA + a0 + pos;
    if (pos + a0 + 8 <= safeA) return gload64u(p);
    uint64_t v = 0;
    for (uint32_t k = 0; k < 8; ++k)
      if (pos + a0 + k < safeA) v |= (uint64_t)gload8(p + k) << (8 * k);
    return v;
  }
  __device__ __forceinline__ void lane_store(uint32_t p, uint8_t b) const {
    if (p < cap_) gstore8(dst + p, b);
  }

  // ---- the ring
  __device__ __forceinline__ void flush(bool force) {
    const uint32_t target = op_;
    // head: up to the next 16-byte boundary of the image
    const uint32_t a = (flushed + sh) & 15u;
    const uint32_t room = target - flushed;
    const uint32_t h = a ? ((16 - a) < room ? (16 - a) : room) : 0u;
    if (lane < h) gstore8(dst + flushed + lane, win[(flushed + sh + lane) & WM]);
    const uint32_t f1 = flushed + h;
    // body: whole 16-byte vectors
    const uint32_t nvec = ((f1 + sh) & 15u) == 0 ? (target - f1) >> 4 : 0u;
#pragma clang loop unroll(disable)
    for (uint32_t v = lane; v < nvec; v += 64)
      gstore128(dst + f1 + 16 * v, *(const u32x4*)(win + ((f1 + sh + 16 * v) & WM)));
    const uint32_t f2 = f1 + (nvec << 4);
    // tail (only at the end of the page)
    const uint32_t tail = force ? target - f2 : 0u;
    for (uint32_t i = lane; i < tail; i += 64) gstore8(dst + f2 + i, win[(f2 + i + sh) & WM]);
    flushed = rfl(f2 + tail);
    stores_pending = true;
  }
  __device__ __forceinline__ void wait_stores() {
    if (stores_pending) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      stores_pending = false;
    }
  }
  __device__ __forceinline__ void advance(uint32_t n) {
    op_ = rfl(op_ + n);
    if (op_ - flushed >= WF) flush(false);
  }

  // `len` bytes from global memory at g (long literals; Raw_Blocks)
  __device__ __forceinline__ void copy_in(const uint8_t* g, uint32_t len) {
#pragma clang loop unroll(disable)
    for (uint32_t i = 0; i < len; i += 256) {
      const uint32_t n = len - i < 256 ? len - i : 256;
      uint8_t b[4];
#pragma unroll
      for (uint32_t k = 0; k < 4; ++k) b[k] = (64 * k + lane < n) ? gload8(g + i + 64 * k + lane) : (uint8_t)0;
#pragma unroll
      for (uint32_t k = 0; k < 4; ++k)
        if (64 * k + lane < n) win[(op_ + 64 * k + lane + sh) & WM] = b[k];
      __builtin_amdgcn_wave_barrier();
      advance(n);
    }
  }
  template <bool CK = true>
  __device__ __forceinline__ bool put_in(uint32_t pos, uint32_t len) {
    if (CK && (len > cap_ - op_ || pos > in_len || len > in_len - pos)) return false;
    copy_in(srcA + a0 + pos, len);
    return true;
  }
  template <bool CK = true>
  __device__ __forceinline__ bool put_fill(uint32_t byte, uint32_t len) {
    if (CK && len > cap_ - op_) return false;
#pragma clang loop unroll(disable)
    for (uint32_t i = 0; i < len; i += 64) {
      const uint32_t n = len - i < 64 ? len - i : 64;
      if (lane < n) win[(op_ + lane + sh) & WM] = (uint8_t)byte;
      __builtin_amdgcn_wave_barrier();
      advance(n);
    }
    return true;
  }
  // <= 64 bytes of a forward stream -> the output
  __device__ __forceinline__ void put_stream64(FwdStream& f, uint32_t a, uint32_t n) {
    f.seek(a);
    const uint8_t v = f.lane_byte(a, n);
    if (lane < n) win[(op_ + lane + sh) & WM] = v;
    __builtin_amdgcn_wave_barrier();
    advance(n);
  }

  // ---- the literals of a ZSTD block
  __device__ __forceinline__ void lit_begin(uint32_t kind, uint32_t pos, uint32_t n) {
    lit_kind = kind; lit_left = n; lit_at = pos;
    if (kind != 2 && n) {
      // kind 0: the payload at `pos`; kind 1: the page's own image at `pos` (huf_streams waited for the stores that put them there)
      const uint8_t* g = kind == 0 ? srcA + a0 + pos : dst + pos;
      const uint32_t x = (uint32_t)((uintptr_t)g & 3u);
      // readable: the payload to the end of the chunk's slack / the image to the end of the page (+ the buffer's 16 bytes of slack)
      const uint32_t room = kind == 0 ? safeA - (a0 + pos) + x : cap_ - pos + x + 4;
      lit_at = x;
      lit.open(g - x, room, x, lane);
    }
  }
  // ONE SEQUENCE of a ZSTD block whose ll literals and ml match bytes fit the wave (ll + ml <= 64, ll may be 0), the source inside the
  // ring (off <= W - 128), the literals a stream (lit_kind 0 / 1) — replayed by the consumer of the two-wave kernel, whose producer has
  // validated it: no checks here. Everything per byte is per LANE: lane l < ll takes literal l; lane ll + m takes match byte m from
  // `off` back — in the ring (one ds_read for all lanes) or, when the match overlaps the literals of this very sequence, one of those
  // literals (out[x] = out[x - off (1 + floor(m / off))]). The literal bytes come out of the stream's register window with ONE
  // ds_bpermute at a per-lane index (round 5 took 8 bytes as a scalar: ~30 scalar instructions per sequence, and the scalar unit is
  // what the two waves of every page on the CU share).
  template <bool FLUSH = true>
  __device__ __forceinline__ void seq_small(uint32_t ll, uint32_t off, uint32_t ml) {
    const int32_t m = (int32_t)lane - (int32_t)ll;                 // match byte index (negative: a literal lane)
    uint32_t mm = m > 0 ? (uint32_t)m : 0u;
    // mm mod off without a division (exact: see put_match; for off >= 64 the quotient is 0)
    mm = mm - off * (uint32_t)(((float)mm + 0.5f) * __builtin_amdgcn_rcpf((float)off));
    const int32_t rel = (int32_t)ll - (int32_t)off + (int32_t)mm;  // source relative to op_: < 0 in the ring, >= 0 a literal of this sequence
    const uint32_t from_ring = win[(op_ + (uint32_t)rel + sh) & WM];
    uint32_t from_lit = 0;
    if (ll) {
      lit.seek(lit_at);
      from_lit = lit.lane_byte_at(lit_at, (m < 0 ? lane : (uint32_t)rel) & 63u, ll);
      lit_left -= ll;
      lit_at = rfl(lit_at + ll);
    }
    const uint32_t v = (m < 0 || rel >= 0) ? from_lit : from_ring;
    if (lane < ll + ml) win[(op_ + lane + sh) & WM] = (uint8_t)v;
    __builtin_amdgcn_wave_barrier();
    if (FLUSH) advance(ll + ml);
    else op_ += ll + ml;           // (the caller has flushed and keeps to 64 of these: see zq_consume)
  }
  // literals + match of one ZSTD sequence (ll > 0)
  __device__ __forceinline__ bool put_seq(uint32_t ll, uint32_t off, uint32_t ml) {
    if (ll <= 8 && ll + ml <= 64 && lit_kind != 2 && off <= WM - 127) {
      if (ll > lit_left || ll + ml > cap_ - op_ || off == 0 || off > op_ - frame0 + ll) return false;
      const uint64_t bits = lit.u64(lit_at);
      lit_left -= ll;
      lit_at = rfl(lit_at + ll);
      pair_small(bits, ll, off, ml);
      return true;
    }
    return put_lit(ll) && put_match(off, ml);
  }
  // one sequence of a ZSTD block (zstd_core.h): ll literals (possibly none), then ml bytes from `off` back
  __device__ __forceinline__ bool seq(uint32_t ll, uint32_t off, uint32_t ml) { return ll ? put_seq(ll, off, ml) : put_match(off, ml); }
  __device__ __forceinline__ uint32_t lit_rest() const { return lit_left; }
  __device__ __forceinline__ int sequences(uint32_t p, uint32_t len, uint32_t nseq, uint32_t als, uint32_t& r0, uint32_t& r1, uint32_t& r2) {
    return zc::seq_loop(*this, p, len, nseq, als, r0, r1, r2);
  }
  template <bool CK = true>
  __device__ __forceinline__ bool put_lit(uint32_t len) {
    if (CK && (len > lit_left || len > cap_ - op_)) return false;
    lit_left -= len;
    if (lit_kind == 2) return put_fill<CK>(lit_at, len);
    if (len <= 64) {
      put_stream64(lit, lit_at, len);
    } else {
      copy_in(lit.gA + lit_at, len);
    }
    lit_at = rfl(lit_at + len);
    return true;
  }

  // THE COMMON SEQUENCE in one LDS round trip: up to 8 literal bytes that the caller holds as a scalar (`litbits`, byte i = bits [8 i, 8 i + 8))
  // followed by a back-reference of `mlen` bytes, lit + mlen <= 64, the source inside the ring (1 <= off <= W - 128, off <= bytes written
  // + lit: checked by the caller). Lane l < lit writes literal byte l; lane lit + m writes match byte m, whose source lies `off` back:
  // either among the bytes already in the ring (one ds_read for all lanes) or among the literal bytes of this very sequence (taken from
  // the scalar, no LDS access) — by periodicity out[x] = out[x - off (1 + floor(m / off))].
  __device__ __forceinline__ void pair_small(uint64_t litbits, uint32_t lit, uint32_t off, uint32_t mlen) {
    const int32_t m = (int32_t)lane - (int32_t)lit;                 // match byte index (negative: a literal lane)
    uint32_t mm = m > 0 ? (uint32_t)m : 0u;
    if (off < 64) {
      const float r = __builtin_amdgcn_rcpf((float)off);          // mm mod off (see put_match)
      mm = mm - off * (uint32_t)(((float)mm + 0.5f) * r);
    }
    const int32_t rel = (int32_t)lit - (int32_t)off + (int32_t)mm;  // source relative to op_: < 0 in the ring, >= 0 a literal of this sequence
    const uint32_t from_ring = win[(op_ + (uint32_t)rel + sh) & WM];
    const uint32_t li = m < 0 ? lane : (uint32_t)rel;             // which literal byte this lane takes if it takes one
    const uint32_t from_lit = (uint32_t)(litbits >> (8 * (li & 7))) & 0xFF;
    const uint32_t v = (m < 0 || rel >= 0) ? from_lit : from_ring;
    if (lane < lit + mlen) win[(op_ + lane + sh) & WM] = (uint8_t)v;
    __builtin_amdgcn_wave_barrier();
    advance(lit + mlen);
  }

  // A BATCH OF m <= 64 SEQUENCES REPLAYED BY THE BYTE (round 6). Lane i < m holds sequence i: `ll` literals, then `ml` bytes from `off`
  // back (off resolved; ml may be 0, then off is not looked at as long as it is 1). Its literals are bytes of the stream `ls`:
  //   SCAN_LITS  the batch's literals follow each other from `lp` (uniform) on — a ZSTD block; `rle`: they are all the byte `lp`;
  //   otherwise  [lp, lp + ll) per lane — LZ4 / Snappy, whose literals lie between the tokens.
  // The caller has checked that the batch fits the page and that the literals exist, and keeps the batch's output below 2^31 bytes.
  // One sequence per step is a chain of LDS round trips (ring read -> literal permute -> ring write: ~1 200 cycles each with the CU's
  // other waves in the way); the sequences of TPC-H's decimal pages are 8 bytes long, so a step of 64 lanes has room for eight of them:
  //   B  wave scans over the batch: where a sequence's output starts / ends (where its literals start). The reach test (offset <=
  //      bytes of the frame before the match) is one vector compare; a batch that fails it is not executed at all (-> false: the
  //      page is malformed, what its image holds does not matter).
  //   C  the output of the batch, 64 bytes per step, one lane per byte: the lane finds its sequence (the scalar end positions of the
  //      few sequences that touch the step against its own position), takes the sequence's fields with ds_bpermute and is either a
  //      literal (its byte comes out of the stream's register window) or a match byte `off` back: in the ring (written by an
  //      earlier step), in the image (far offsets) or — the source is a byte of this very step — another lane, followed by pointer
  //      jumping (<= 6 rounds; none when no match of the step reaches into the step itself).
  // Any sequence length and any offset takes this path: a long match is simply many steps of one sequence.
  template <bool SCAN_LITS>
  __device__ __forceinline__ bool replay(FwdStream& ls, uint32_t m, uint32_t ll, uint32_t ml, uint32_t off, uint32_t lp, bool rle, uint32_t& lits_total) {
    // ---- B
    const bool act = lane < m;
    const uint32_t llv = act ? ll : 0u, lenv = act ? ll + ml : 0u;
    uint32_t Ei = lenv, Li = llv;                                        // inclusive scans
#pragma unroll
    for (uint32_t d = 1; d < 64; d <<= 1) {
      const uint32_t pe = bperm(lane - d, Ei);
      Ei += lane >= d ? pe : 0u;
      if (SCAN_LITS) {
        const uint32_t pl = bperm(lane - d, Li);
        Li += lane >= d ? pl : 0u;
      }
    }
    const uint32_t Sx = Ei - lenv;
    const uint32_t Lx = SCAN_LITS ? (rle ? 0u : lp + Li - llv) : lp;     // where the sequence's literals start (from ls.gA)
    const uint32_t T = rdl(Ei, 63);
    if (SCAN_LITS) lits_total = rdl(Li, 63);
    // off == 0 or off > bytes of this frame written before the match  <=>  off - 1 >= that count (unsigned)
    if (__builtin_amdgcn_ballot_w64(act && off - 1u >= op_ - frame0 + Sx + llv) != 0) return false;
    // ---- C
    const uint32_t lim = WM - 127;
#pragma clang loop unroll(disable)
    for (uint32_t c0 = 0; c0 < T; c0 += 64) {
      const uint32_t n = T - c0 < 64 ? T - c0 : 64;
      const uint32_t b = c0 + lane;
      const bool live = lane < n;
      // the sequences that touch this step: f .. g - 1 (those before f ended at or before c0, those from g on start behind it)
      const uint32_t f = (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(act && Ei <= c0));
      const uint32_t g = (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(act && Sx < c0 + n));
      uint32_t idx = f;
#pragma clang loop unroll(disable)
      for (uint32_t k = f; k + 1 < g; ++k) idx += b >= rdl(Ei, k) ? 1u : 0u;
      const uint32_t Si = bperm(idx, Sx), Lxi = bperm(idx, Lx), lli = bperm(idx, llv), offi = bperm(idx, off);
      const uint32_t kk = b - Si;            // byte of the sequence
      const bool isl = live && kk < lli;
      const uint32_t lidx = Lxi + kk;        // a literal lane: its byte of the stream
      uint32_t from_lit = lp;                // (RLE literals: lp is the byte)
      uint64_t lmask = __builtin_amdgcn_ballot_w64(isl);
      if (!(SCAN_LITS && rle)) {
        // the literals of a step lie in stream order: the first literal lane holds the lowest position, the last one the highest; between
        // them lie, besides literals, only the tokens of the sequences that start in the step — one window of 256 bytes from the first
        // position on holds them all for whatever an encoder writes (a Snappy stream of one-byte literals with five-byte headers is
        // legal, though: the loop takes as many windows as it needs)
        while (lmask != 0) {
          const uint32_t la0 = rdl(lidx, (uint32_t)__builtin_ctzll(lmask));
          const uint32_t reach = rdl(lidx, 63u - (uint32_t)__builtin_clzll(lmask)) - la0 + 1;
          const bool now = isl && lidx - la0 < 256u;
          ls.seek(la0);
          const uint32_t x = ls.lane_byte_at(la0, (lidx - la0) & 255u, reach < 256u ? reach : 256u);
          from_lit = now ? x : from_lit;
          lmask &= ~__builtin_amdgcn_ballot_w64(now);
        }
      }
      const bool ism = live && !isl;
      const bool inch = ism && offi <= lane;     // the source is a byte of this very step (lane - offi)
      const bool far = ism && offi > lim;        // the source left the ring: it is in the image, flushed long ago (see put_match)
      uint32_t from_out = win[(op_ + lane - offi + sh) & WM];
      if (__builtin_amdgcn_ballot_w64(far) != 0) {
        wait_stores();
        if (far) from_out = gload8(dst + op_ + lane - offi);
      }
      uint32_t v = isl ? from_lit : from_out;
      if (__builtin_amdgcn_ballot_w64(inch) != 0) {
        uint32_t pend = inch ? 1u : 0u, sl = lane - offi;
        while (__builtin_amdgcn_ballot_w64(pend != 0) != 0) {
          const uint32_t pv = bperm(sl, v), pp = bperm(sl, pend), ps = bperm(sl, sl);
          v = (pend && !pp) ? pv : v;
          sl = (pend && pp) ? ps : sl;
          pend = (pend && !pp) ? 0u : pend;
        }
      }
      if (live) win[(op_ + lane + sh) & WM] = (uint8_t)v;
      __builtin_amdgcn_wave_barrier();
      advance(n);
    }
    return true;
  }

  template <bool CK = true>
  __device__ __forceinline__ bool put_match(uint32_t off, uint32_t len) {
    if (CK && (off == 0 || off > op_ - frame0 || len > cap_ - op_)) return false;
    if (off <= WM - 63) {
      // the source lies in the ring. By periodicity out[cur + l] = out[cur - off + (l mod off)]: every source byte lies before `cur`,
      // so the 64 lanes read (one LDS instruction) before any of them writes
      uint32_t j = lane;
      if (off < 64) {
        // lane mod off without an integer division: (lane + 0.5) / off is at least 0.5 / 63 away from an integer, far more than the
        // error of v_rcp_f32, so the truncation is exact
        const float r = __builtin_amdgcn_rcpf((float)off);
        j = lane - off * (uint32_t)(((float)lane + 0.5f) * r);
      }
#pragma clang loop unroll(disable)
      for (uint32_t i = 0; i < len; i += 64) {
        const uint32_t n = len - i < 64 ? len - i : 64;
        if (lane < n) {
          const uint8_t v = win[(op_ - off + j + sh) & WM];
          win[(op_ + lane + sh) & WM] = v;
        }
        __builtin_amdgcn_wave_barrier();
        advance(n);
      }
    } else {
      // the source left the ring: it is in the image (flushed at least WF + 64 bytes ago: WF <= W - 512)
#pragma clang loop unroll(disable)
      for (uint32_t i = 0; i < len; i += 64) {
        const uint32_t n = len - i < 64 ? len - i : 64;
        wait_stores();
        if (lane < n) win[(op_ + lane + sh) & WM] = gload8(dst + op_ - off + lane);
        __builtin_amdgcn_wave_barrier();
        advance(n);
      }
    }
    return true;
  }

  // the Huffman-coded literal streams of a block, one lane each, into the image at [outp, outp + regen)
  __device__ __forceinline__ bool huf_streams(uint32_t streams, uint32_t sp, uint32_t l1, uint32_t l2, uint32_t l3, uint32_t l4, uint32_t seg,
                                              uint32_t regen, uint32_t maxbits, uint32_t outp) {
    bool ok = true;
    if (lane < streams) {
      uint32_t pos = sp, len = l1, n = seg;
      if (lane == 1) { pos = sp + l1; len = l2; }
      if (lane == 2) { pos = sp + l1 + l2; len = l3; }
      if (lane == 3) { pos = sp + l1 + l2 + l3; len = l4; n = regen - 3 * seg; }
      // (lengths were checked against the literals section by the caller: every stream lies inside the payload)
      ok = zc::huf_stream(*this, huf(), maxbits, pos, len, n, outp + lane * seg);
    }
    const bool all = __builtin_amdgcn_ballot_w64(!ok) == 0;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the literals are read back (lit_begin) by other lanes
    stores_pending = false;
    return all;
  }

  __device__ __forceinline__ void begin(const DvJob& P, uint8_t* lds, uint32_t ring_bytes, uint32_t ln) {
    const uint32_t lev = P.lev_len;
    const uint8_t* s0 = P.src + lev;
    a0 = (uint32_t)((uintptr_t)s0 & 3u);
    srcA = s0 - a0;
    in_len = P.comp_len - lev;
    safeA = (P.src_safe - lev + a0) & ~3u;
    lane = ln;
    slide(0);
    dst = P.dst + lev;
    cap_ = P.uncomp_len - lev; op_ = 0; flushed = 0; frame0 = 0;
    sh = (uint32_t)((uintptr_t)dst & 15u);
    win = lds; WM = ring_bytes - 1; WF = ring_bytes / 2;
    tab = lds + ring_bytes;
    stores_pending = false;
    lit_kind = 2; lit_at = 0; lit_left = 0;
  }
};

// ---------------------------------------------------------------------------------------------
// ZSTD with the parse and the copy on two waves (round 5, VERDICT r04 #2b). A page is a serial chain — decode a sequence (two FSE state
// walks, three bit reads, the repeat-offset rules), then move its bytes — and one wave walking both halves took 28-30 ms for a
// 160 KB page of 20 000 sequences. Here wave 0 of a 128-thread workgroup (the PRODUCER: ZProd) runs the whole format — headers,
// tables, Huffman literals, the sequence bitstream — but executes nothing: it checks every command against its own count of the
// output position and queues it; wave 1 (the CONSUMER) replays the commands against the ring with the executor of the one-wave
// kernel. The queue is a single-producer / single-consumer ring in LDS: 64 commands collect in the producer's registers
// (a compare and four selects per command), leave as one 16-byte LDS store per lane, and are taken 64 at a time (one LDS load per lane, v_readlane per
// command); `tail` / `head` are LDS words written by one side each. Because the producer validates, the consumer cannot fail: it
// always drains to the END command, the producer always sends one — neither wave can wait for something that will not come.
// One rendezvous per block: Huffman literals are decoded into the TAIL of the page's output region, where the previous block's
// literals may still be waiting to be consumed, so the producer lets the queue drain before it writes them.
// ---------------------------------------------------------------------------------------------
constexpr uint32_t ZW_RING = 4096;                          // ZSTD: ring bytes (r06 sweep on the 7-column set, tools/probes/r06_zstd_ring.sh: 16 KiB 22.4 ms, 8 KiB 20.0, 4 KiB 16.8, 2 KiB 17.8 — pages per CU by LDS)
constexpr uint32_t ZW_TABLES = 13312 + zc::SCR_BYTES;       // Huffman 4 KiB + LL 4 KiB + ML 4 KiB + OF 1 KiB + scratch
constexpr uint32_t ZW_LDS = ZW_RING + ZW_TABLES;
#ifndef DBHIP_ZQ_CAP
#define DBHIP_ZQ_CAP 64
#endif
constexpr uint32_t ZQ_CAP = DBHIP_ZQ_CAP;    // commands in the queue (16 bytes each): ONE batch — the consumer takes a batch into registers and gives
                                             // the slots back before it executes it, the producer collects the next one in registers meanwhile
constexpr uint32_t ZQ_BYTES = ZQ_CAP * 16 + 16;
constexpr uint32_t ZW2_LDS = ZW_RING + ZW_TABLES + ZQ_BYTES;   // the two-wave kernel: + the command queue
enum { ZC_SEQ = 0, ZC_LIT_BEGIN = 1, ZC_LIT = 2, ZC_IN = 3, ZC_FILL = 4, ZC_END = 5, ZC_FRAME = 6 };
typedef uint32_t u32x4q __attribute__((ext_vector_type(4)));

struct ZQueue {
  u32x4q* slots;      // [ZQ_CAP]
  uint32_t* ctl;      // [0] tail (written by the producer), [1] head: commands TAKEN, [3] commands EXECUTED (written by the consumer)
  __device__ __forceinline__ uint32_t tail() const { return __hip_atomic_load(&ctl[0], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP); }
  __device__ __forceinline__ uint32_t head() const { return __hip_atomic_load(&ctl[1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP); }
  __device__ __forceinline__ uint32_t done() const { return __hip_atomic_load(&ctl[3], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP); }
  __device__ __forceinline__ void take(uint32_t h) const { __hip_atomic_store(&ctl[1], h, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); }
  __device__ __forceinline__ void executed(uint32_t h) const { __hip_atomic_store(&ctl[3], h, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); }
  // Belt and braces: a wait that sees no progress for ~2^24 polls (seconds) raises ctl[2]; both waves then leave and the page is
  // reported malformed — a logic error in the hand-shake must not be able to hang the device.
  __device__ __forceinline__ bool dead() const { return __hip_atomic_load(&ctl[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0; }
  __device__ __forceinline__ void kill() const { __hip_atomic_store(&ctl[2], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
};
constexpr uint32_t ZQ_POLLS = 1u << 24;

// The producer's sequence loop as a function of its own (round 6). The decoder is one inlined function whose outer loops keep ~65
// loop-invariant scalars alive; inlined into it, the sequence loop had its own values — the bit register, the repeat offsets, the
// sequence counter — spilled to VGPR lanes and read back every iteration (71 v_readlane / v_writelane in the loop). A function that is
// NOT inlined gets an allocation of its own: the caller's scalars wait in callee-saved registers, which the function parks in VGPR
// lanes once, at its entry. The state it needs crosses the call through the LDS scratch area of zstd_core.h, which only the table
// builders use and which is idle while sequences are decoded: bytes [0, 1280) five per-lane words (the input window and the four
// words of the collected commands), [1280, 1408) the scalars.
__device__ __attribute__((noinline)) void zprod_sequences(uint32_t tab_lds);
enum { ZH_SRC_LO = 0, ZH_SRC_HI, ZH_A0, ZH_SAFE, ZH_WLO, ZH_QN, ZH_QTAIL, ZH_OP, ZH_FRAME0, ZH_CAP, ZH_LIT_LEFT, ZH_VIOL, ZH_FAILED, ZH_R0, ZH_R1,
       ZH_R2, ZH_P, ZH_LEN, ZH_NSEQ, ZH_ALS, ZH_RC, ZH_XW_LO, ZH_XW_HI, ZH_WORDS };
static_assert(1280 + 4 * ZH_WORDS <= zc::SCR_PARK, "the hand-over block must fit the scratch area below the parking slots");

struct ZProd : ZWave {
  ZQueue q;
  uint32_t qn;                 // commands collected in registers (uniform)
  uint32_t qtail;              // this side's copy of the tail
  uint32_t b0, b1, b2, b3;     // lane i: words of collected command i
  // Sequences are validated BY THE BATCH (round 6): a sequence only adds to op_ and takes from lit_left (kept signed); the batch of
  // <= 64 collected commands is judged as a whole when it leaves for the queue — op_ <= cap_, lit_left >= 0 (`viol` keeps what a
  // block's end found) — and a batch that fails is dropped, so the consumer never sees a command that writes past the page or takes
  // literals the block does not have. (Everything is monotonic between two flushes: op_ grows by < 2^18 per sequence and cap_ < 2^31,
  // lit_left starts below 2^17 and falls by < 2^17 per sequence.) The third test — the offset stays inside the frame — is the
  // consumer's, together with the repeat-offset history (seq_raw below).
  uint32_t viol;
  bool failed;                 // a batch was refused: every later call fails
#ifdef DBHIP_EXPERIMENTS
  uint64_t x_wait = 0;         // cycles spent waiting for the consumer (queue full / drain)
#define ZX_T0 const uint64_t zx_t0 = __builtin_readcyclecounter();
#define ZX_ADD(acc) acc += __builtin_readcyclecounter() - zx_t0;
#else
#define ZX_T0
#define ZX_ADD(acc)
#endif

  __device__ __forceinline__ void qbegin(const ZQueue& Q) { q = Q; qn = 0; qtail = 0; b0 = b1 = b2 = b3 = 0; viol = cap_ >> 31; failed = false; }
  __device__ __forceinline__ bool batch_ok() const {   // (selects between integers: a bool turned into an integer leaves the scalar unit)
    uint32_t bad = viol;
    bad = op_ > cap_ ? 1u : bad;
    bad = (int32_t)lit_left < 0 ? 1u : bad;
    return bad == 0;
  }
  // the collected commands -> the queue (waits for room), unchecked
  __device__ __forceinline__ void qsend() {
    if (qn == 0) return;
    ZX_T0
    for (uint32_t polls = 0; qtail + qn - q.head() > ZQ_CAP; ++polls) {
      if (polls >= ZQ_POLLS) q.kill();
      if (q.dead()) { qn = 0; return; }
      __builtin_amdgcn_s_sleep(2);
    }
    ZX_ADD(x_wait)
    if (lane < qn) q.slots[(qtail + lane) & (ZQ_CAP - 1)] = u32x4q{b0, b1, b2, b3};
    qtail = rfl(qtail + qn);
    qn = 0;
    __hip_atomic_store(&q.ctl[0], qtail, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  __device__ __forceinline__ void qflush() {
    if (!batch_ok()) { failed = true; qn = 0; return; }
    qsend();
  }
  __device__ __forceinline__ void push(uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3) {
    // (one compare + four selects: this clang has no writelane builtin. r05: writing each command to the queue at once — lane 0, one
    //  16-byte LDS store per sequence — measured 49.8 ms against 41.8 ms on the 7-column set: the store sits in the same LDS queue as
    //  the table reads of the next sequence)
    const bool mine = lane == qn;
    b0 = mine ? w0 : b0; b1 = mine ? w1 : b1; b2 = mine ? w2 : b2; b3 = mine ? w3 : b3;
    qn = rfl(qn + 1);
    if (qn == 64) qflush();
  }
  // every queued command has been executed
  __device__ __forceinline__ void drain() {
    qflush();
    if (failed) return;
    ZX_T0
    for (uint32_t polls = 0; q.done() != qtail; ++polls) {
      if (polls >= ZQ_POLLS) q.kill();
      if (q.dead()) return;
      __builtin_amdgcn_s_sleep(2);
    }
    ZX_ADD(x_wait)
  }

  // ---- the output half of the interface zstd_core.h expects: validate against the producer's own position, queue, count
  __device__ __forceinline__ void lit_begin(uint32_t kind, uint32_t pos, uint32_t n) {
    viol = (int32_t)lit_left < 0 ? 1u : viol;   // (the block before this one took more literals than it had)
    lit_left = n;
    push(ZC_LIT_BEGIN | (kind << 8), pos, 0, n);
  }
  // a sequence: command words (ll, ml, ov != 0); the consumer ignores the fourth word of a lane that holds one. ov is the offset VALUE
  // of the format (1..3: a repeat code, else offset + 3): the repeat-offset history lives in the CONSUMER, which also checks that the
  // offset stays inside the frame — 23 scalar instructions per sequence that the wave with time to spare now carries (round 6: the
  // page's latency is the producer's; with those it was 115 : 35 instructions per sequence, now 92 : 58).
  static constexpr bool RESOLVES_OFFSETS = true;
  __device__ __forceinline__ bool seq_raw(uint32_t ll, uint32_t ov, uint32_t ml) {
    lit_left -= ll;
    const bool mine = lane == qn;
    b0 = mine ? ll : b0; b1 = mine ? ml : b1; b2 = mine ? ov : b2;
    op_ += ll + ml;
    qn += 1;
    if (__builtin_expect(qn == 64, 0)) qflush();
    return true;   // (a refused batch is reported when the block's sequences are done: `failed` — what follows it is dropped batch by batch)
  }
  __device__ __forceinline__ bool seq(uint32_t, uint32_t, uint32_t) { return false; }   // (not used: RESOLVES_OFFSETS)
  __device__ __forceinline__ void frame_begin() {
    frame0 = op_;
    push(ZC_FRAME, 0, 0, 0);
  }
  // the same with the address arithmetic on the VECTOR unit (the state number is copied to a VGPR behind an opaque asm: shift + add
  // are then one v_lshl_add_u32 instead of four scalar instructions per table, and scalar issue slots are what this wave runs out of)
  __device__ __forceinline__ uint64_t ll_at(uint32_t st) const { uint32_t v = st; asm("" : "+v"(v)); return llt()[v]; }
  __device__ __forceinline__ uint64_t ml_at(uint32_t st) const { uint32_t v = st; asm("" : "+v"(v)); return mlt()[v]; }
  __device__ __forceinline__ uint32_t of_at(uint32_t st) const { uint32_t v = st; asm("" : "+v"(v)); return oft()[v]; }
  // ---- the hand-over (see zprod_sequences)
  __device__ __forceinline__ uint32_t* hot_words() const { return (uint32_t*)(scr() + 1280); }
  __device__ __forceinline__ void hot_put_lanes() const {
    uint32_t* L = (uint32_t*)scr();
    L[lane] = la; L[64 + lane] = b0; L[128 + lane] = b1; L[192 + lane] = b2; L[256 + lane] = b3;
  }
  __device__ __forceinline__ void hot_get_lanes() {
    const uint32_t* L = (const uint32_t*)scr();
    la = L[lane]; b0 = L[64 + lane]; b1 = L[128 + lane]; b2 = L[192 + lane]; b3 = L[256 + lane];
  }
  // what the sequence loop changes: written by one side of the call, read by the other
  __device__ __forceinline__ void hot_put_state() const {
    uint32_t* H = hot_words();
    if (lane == 0) {
      H[ZH_WLO] = wlo; H[ZH_QN] = qn; H[ZH_QTAIL] = qtail; H[ZH_OP] = op_; H[ZH_LIT_LEFT] = lit_left; H[ZH_VIOL] = viol;
      H[ZH_FAILED] = failed ? 1u : 0u;
#ifdef DBHIP_EXPERIMENTS
      H[ZH_XW_LO] = (uint32_t)x_wait; H[ZH_XW_HI] = (uint32_t)(x_wait >> 32);
#endif
    }
    hot_put_lanes();
  }
  __device__ __forceinline__ void hot_get_state() {
    const uint32_t* H = hot_words();
    wlo = rfl(H[ZH_WLO]); qn = rfl(H[ZH_QN]); qtail = rfl(H[ZH_QTAIL]); op_ = rfl(H[ZH_OP]); lit_left = rfl(H[ZH_LIT_LEFT]);
    viol = rfl(H[ZH_VIOL]); failed = rfl(H[ZH_FAILED]) != 0;
#ifdef DBHIP_EXPERIMENTS
    x_wait = (uint64_t)rfl(H[ZH_XW_LO]) | ((uint64_t)rfl(H[ZH_XW_HI]) << 32);
#endif
    hot_get_lanes();
  }
  __device__ __forceinline__ int sequences(uint32_t p, uint32_t len, uint32_t nseq, uint32_t als, uint32_t& r0, uint32_t& r1, uint32_t& r2) {
    uint32_t* H = hot_words();
    if (lane == 0) {
      H[ZH_SRC_LO] = (uint32_t)(uintptr_t)srcA; H[ZH_SRC_HI] = (uint32_t)((uintptr_t)srcA >> 32); H[ZH_A0] = a0; H[ZH_SAFE] = safeA;
      H[ZH_FRAME0] = frame0; H[ZH_CAP] = cap_; H[ZH_R0] = r0; H[ZH_R1] = r1; H[ZH_R2] = r2;
      H[ZH_P] = p; H[ZH_LEN] = len; H[ZH_NSEQ] = nseq; H[ZH_ALS] = als;
    }
    hot_put_state();
    zprod_sequences((uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)tab);
    hot_get_state();
    r0 = rfl(H[ZH_R0]); r1 = rfl(H[ZH_R1]); r2 = rfl(H[ZH_R2]);
    const int rc = (int)rfl(H[ZH_RC]);
    return failed && rc == zc::OK ? (int)zc::CORRUPT_ : rc;
  }
  __device__ __forceinline__ uint32_t lit_rest() {
    viol = (int32_t)lit_left < 0 ? 1u : viol;
    return (int32_t)lit_left < 0 ? 0u : lit_left;
  }
  __device__ __forceinline__ bool put_lit(uint32_t len) {
    if (failed || (lit_left >> 31) || len > lit_left || op_ > cap_ || len > cap_ - op_) return false;
    lit_left -= len;
    push(ZC_LIT, len, 0, 0);
    op_ = rfl(op_ + len);
    return !failed;
  }
  __device__ __forceinline__ bool put_in(uint32_t pos, uint32_t len) {
    if (failed || op_ > cap_ || len > cap_ - op_ || pos > in_len || len > in_len - pos) return false;
    push(ZC_IN, pos, 0, len);
    op_ = rfl(op_ + len);
    return !failed;
  }
  __device__ __forceinline__ bool put_fill(uint32_t byte, uint32_t len) {
    if (failed || op_ > cap_ || len > cap_ - op_) return false;
    push(ZC_FILL, byte, 0, len);
    op_ = rfl(op_ + len);
    return !failed;
  }
  __device__ __forceinline__ bool huf_streams(uint32_t streams, uint32_t sp, uint32_t l1, uint32_t l2, uint32_t l3, uint32_t l4, uint32_t seg,
                                              uint32_t regen, uint32_t maxbits, uint32_t outp) {
    drain();   // the tail of the output region may still hold the previous block's literals
    if (failed || regen > cap_ || outp != cap_ - regen) return false;
    return ZWave::huf_streams(streams, sp, l1, l2, l3, l4, seg, regen, maxbits, outp);
  }
  __device__ __forceinline__ void end(uint32_t status) {
    if (failed || !batch_ok()) {   // (what is still collected belongs to a refused batch)
      qn = 0;
      if (status == (uint32_t)zc::OK) status = (uint32_t)zc::CORRUPT_;
    }
    push(ZC_END, status, 0, 0);
    qsend();
  }
};

__device__ __attribute__((noinline)) void zprod_sequences(uint32_t tab_lds) {
  ZProd a;
  a.tab = (uint8_t*)(__attribute__((address_space(3))) uint8_t*)(uintptr_t)rfl(tab_lds);
  a.lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
  uint32_t* H = a.hot_words();
  a.srcA = (const uint8_t*)(uintptr_t)((uint64_t)rfl(H[ZH_SRC_LO]) | ((uint64_t)rfl(H[ZH_SRC_HI]) << 32));
  a.a0 = rfl(H[ZH_A0]); a.safeA = rfl(H[ZH_SAFE]); a.frame0 = rfl(H[ZH_FRAME0]); a.cap_ = rfl(H[ZH_CAP]);
  a.q.slots = (u32x4q*)(a.tab + ZW_TABLES);
  a.q.ctl = (uint32_t*)(a.tab + ZW_TABLES + ZQ_CAP * 16);
  a.hot_get_state();
  uint32_t r0 = rfl(H[ZH_R0]), r1 = rfl(H[ZH_R1]), r2 = rfl(H[ZH_R2]);
  const int rc = zc::seq_loop(a, rfl(H[ZH_P]), rfl(H[ZH_LEN]), rfl(H[ZH_NSEQ]), rfl(H[ZH_ALS]), r0, r1, r2);
  if (a.lane == 0) { H[ZH_R0] = r0; H[ZH_R1] = r1; H[ZH_R2] = r2; H[ZH_RC] = (uint32_t)rc; }
  a.hot_put_state();
}

// the consumer: replays the commands until END; -> the producer's status
__device__ __forceinline__ uint32_t zq_consume(ZWave& w, const ZQueue& q, uint32_t xmode, uint64_t* xwait = nullptr) {
  uint32_t head = 0;
  uint32_t r0 = 1, r1 = 4, r2 = 8;   // the repeat offsets of the frame being replayed
  uint32_t bad = 0;                  // an offset reached before its frame: nothing is executed from there on
  for (;;) {
    uint32_t tail = q.tail();
#ifdef DBHIP_EXPERIMENTS
    const uint64_t zx_c0 = __builtin_readcyclecounter();
#endif
    for (uint32_t polls = 0; tail == head; ++polls) {
      if (polls >= ZQ_POLLS) q.kill();
      if (q.dead()) return (uint32_t)zc::CORRUPT_;
      __builtin_amdgcn_s_sleep(16);   // (a batch of 64 commands takes the producer ~10^5 cycles: a poll every ~1000 costs the CU's shared scalar unit next to nothing)
      tail = q.tail();
    }
#ifdef DBHIP_EXPERIMENTS
    if (xwait) *xwait += __builtin_readcyclecounter() - zx_c0;
#endif
    const uint32_t m = rfl(tail - head < 64 ? tail - head : 64);
    const u32x4q e = q.slots[(head + (w.lane < m ? w.lane : 0)) & (ZQ_CAP - 1)];
    head = rfl(head + m);
    q.take(head);       // (a release store: the slots have been read; what follows works on `e`)
    // THE COMMON BATCH: nothing but sequences — replayed BY THE BYTE, not by the sequence (ZWave::replay; one sequence per step made the
    // consumer the wave the page waited for). What is left to do here is the only serial part, phase A: the repeat-offset history
    // (RFC 8878 3.1.1.5) as one scalar pass over the batch without a memory access; the resolved offset of sequence i lands in lane i.
    if (__builtin_amdgcn_ballot_w64(w.lane < m && e.z == 0) == 0 && !(xmode & 1)) {
#ifdef DBHIP_EXPERIMENTS
      if (xwait) xwait[1] += m;
#endif
      // ---- A: the repeat-offset history of the batch as a SCAN (lane i = sequence i). A sequence maps the history (r0, r1, r2) to a new
      // one whose components are each an old component (+ 0 or - 1) or a constant:
      //   j = 0: (r0, r1, r2)   1: (r1, r0, r2)   2: (r2, r0, r1)   3: (r0 - 1, r0, r1)   4, a new offset c: (c, r0, r1)
      // Such maps compose — component = (which old component | constant, value to add | the constant) — so six rounds of a wave scan
      // give every lane the history in front of its sequence; a scalar loop over the batch (16 scalar instructions per sequence) was
      // the consumer's largest share of the scalar unit, which every wave of the CU waits for.
      const bool act_a = w.lane < m;
      const uint32_t ov_l = act_a ? e.z : 1u, ll_l = act_a ? e.x : 1u;            // (idle lanes: j = 0, the identity)
      const uint32_t jj = ov_l > 3 ? 4u : ov_l - 1 + (ll_l == 0 ? 1u : 0u);
      // this sequence's map: packed component kinds (2 bits each: 0..2 = that old component, 3 = a constant) and the three values
      uint32_t mk = jj == 0 ? 0x24u : jj == 1 ? 0x21u : jj == 2 ? 0x12u : jj == 3 ? 0x10u : 0x13u;
      uint32_t v0 = jj == 3 ? 0xFFFFFFFFu : jj == 4 ? ov_l - 3 : 0u, v1 = 0, v2 = 0;
#pragma unroll
      for (uint32_t d = 1; d < 64; d <<= 1) {
        // the map of the d lanes in front (already combined), applied FIRST: component c of the result = own component c with its
        // reference into the old history followed through the other map
        const uint32_t pk = bperm(w.lane - d, mk), p0 = bperm(w.lane - d, v0), p1 = bperm(w.lane - d, v1), p2 = bperm(w.lane - d, v2);
        if (w.lane >= d) {
          uint32_t nk = 0, n0, n1, n2;
#define ZQ_COMPOSE(C_, VC_, NC_)                                                                   \
          {                                                                                          \
            const uint32_t k = (mk >> (2 * C_)) & 3u;                                                \
            const uint32_t pkind = k == 0 ? pk & 3u : k == 1 ? (pk >> 2) & 3u : (pk >> 4) & 3u;      \
            const uint32_t pval = k == 0 ? p0 : k == 1 ? p1 : p2;                                    \
            nk |= (k == 3 ? 3u : pkind) << (2 * C_);                                                 \
            NC_ = k == 3 ? VC_ : pval + VC_;                                                         \
          }
          ZQ_COMPOSE(0, v0, n0)
          ZQ_COMPOSE(1, v1, n1)
          ZQ_COMPOSE(2, v2, n2)
#undef ZQ_COMPOSE
          mk = nk; v0 = n0; v1 = n1; v2 = n2;
        }
      }
      // history behind this lane's sequence (inclusive) and in front of it (the lane before; lane 0: the batch's start)
      const uint32_t k0 = mk & 3u, k1 = (mk >> 2) & 3u, k2 = (mk >> 4) & 3u;
      const uint32_t a0 = (k0 == 3 ? 0u : k0 == 0 ? r0 : k0 == 1 ? r1 : r2) + v0;
      const uint32_t a1 = (k1 == 3 ? 0u : k1 == 0 ? r0 : k1 == 1 ? r1 : r2) + v1;
      const uint32_t a2 = (k2 == 3 ? 0u : k2 == 0 ? r0 : k2 == 1 ? r1 : r2) + v2;
      // (the permutes by every lane: a lane that is masked off gives its neighbour nothing to read)
      const uint32_t q0 = bperm(w.lane - 1, a0), q1 = bperm(w.lane - 1, a1), q2 = bperm(w.lane - 1, a2);
      const uint32_t b0 = w.lane ? q0 : r0, b1 = w.lane ? q1 : r1, b2 = w.lane ? q2 : r2;
      const uint32_t offv = jj == 0 ? b0 : jj == 1 ? b1 : jj == 2 ? b2 : jj == 3 ? b0 - 1 : ov_l - 3;
      r0 = rdl(a0, 63); r1 = rdl(a1, 63); r2 = rdl(a2, 63);                         // (idle lanes are identities: lane 63 holds the batch's end)
      // ---- B, C
      if (!bad) {
        uint32_t TL = 0;
        if (!w.replay<true>(w.lit, m, e.x, e.y, offv, w.lit_at, w.lit_kind == 2, TL)) bad = 1;
        else {
          if (w.lit_kind != 2) w.lit_at = rfl(w.lit_at + TL);
          w.lit_left -= TL;
        }
      }
      q.executed(head);
      continue;
    }
    for (uint32_t i = 0; i < m; ++i) {
      const uint32_t w0 = rdl(e.x, i), w1 = rdl(e.y, i), ov = rdl(e.z, i), w3 = rdl(e.w, i);
      if (ov) {                        // a sequence: w0 literals, then w1 bytes from an offset back
        if (xmode & 1) continue;       // (experiments build only: xmode is 0 otherwise)
        // repeat offsets (RFC 8878 3.1.1.5) as selects on j = 0..3 (a repeat code, shifted by one when there are no literals) / 4
        // (a new offset): j = 0 offset r0, history unchanged; j = 1 r1, swapped to the front; j = 2 r2, j = 3 r0 - 1, j = 4 ov - 3
        uint32_t j = ov - 1 + (w0 == 0 ? 1u : 0u);
        j = ov > 3 ? 4u : j;
        uint32_t off = ov - 3;
        off = j == 0 ? r0 : off;
        off = j == 1 ? r1 : off;
        off = j == 2 ? r2 : off;
        off = j == 3 ? r0 - 1 : off;
        r2 = j >= 2 ? r1 : r2;
        r1 = j >= 1 ? r0 : r1;
        r0 = off;
        // off == 0 or off > bytes of this frame written so far + the literals  <=>  off - 1 >= that sum (unsigned). From the first
        // such sequence on nothing is executed any more (the queue is still drained: the producer must never wait for good)
        bad = (off - 1u >= w.op_ - w.frame0 + w0) ? 1u : bad;
        if (bad) continue;
        if (w0 + w1 <= 64 && off <= w.WM - 127 && w.lit_kind != 2) {
          w.seq_small(w0, off, w1);
        } else {                       // (validated by the producer: the unchecked forms)
          if (w0) (void)w.put_lit<false>(w0);
          (void)w.put_match<false>(off, w1);
        }
        continue;
      }
      const uint32_t cmd = w0 & 0xFF;
      if (cmd == ZC_END) {
        q.executed(head);
        return bad && w1 == (uint32_t)zc::OK ? (uint32_t)zc::CORRUPT_ : w1;
      }
      if (bad) continue;
      if (cmd == ZC_LIT) (void)w.put_lit<false>(w1);
      else if (cmd == ZC_LIT_BEGIN) w.lit_begin(w0 >> 8, w1, w3);
      else if (cmd == ZC_IN) (void)w.put_in<false>(w1, w3);
      else if (cmd == ZC_FILL) (void)w.put_fill<false>(w1, w3);
      else if (cmd == ZC_FRAME) { w.frame0 = w.op_; r0 = 1; r1 = 4; r2 = 8; }   // back-references and the history do not reach across frames
    }
    q.executed(head);
  }
}

constexpr uint32_t LZ_RING = 8192;                          // LZ4 / Snappy: ring bytes (nothing else in LDS). r05 sweep on the 7-column lineitem set:
                                                            // 16 KiB 19.2 / 21.1 ms (LZ4 / Snappy), 8 KiB 14.3 / 16.3, 4 KiB 14.8 / 17.1

// ---------------------------------------------------------------------------------------------
// LZ4 block format (lz4_Block_format.md) and Snappy raw format (format_description.txt) on the same wave: the payload is ONE forward
// stream — tokens / tags are read from its register chunks (scalar), literals move from the chunks to the ring.
// ---------------------------------------------------------------------------------------------
// LZ4 parses UP TO 64 SEQUENCES AHEAD — scalar work on the stream's register window, no LDS round trip — collecting (literals, match
// length, offset, where the literals lie in the payload) in lane i of four registers, and hands the batch to ZWave::replay, which moves
// 64 output bytes per step through a second view of the same stream (the literals lie behind the parser's position). With 20 one-wave
// pages per CU the INSTRUCTION COUNT rules, not the latency: the common token (<= 5 literals, no length extension, everything inside
// the chunk the stream holds) has a loop of its own with one exit; every other case leaves it for one pass of the general code.
struct LzBatch {
  uint32_t m;                // sequences collected (uniform)
  uint32_t ll, ml, off, lp;  // lane i: sequence i
  uint32_t room;             // bytes the page still has room for behind the collected sequences
  __device__ __forceinline__ void begin(const ZWave& w) { m = 0; ll = ml = off = lp = 0; room = w.cap_ - w.op_; }
  __device__ __forceinline__ void put(uint32_t lane, uint32_t l, uint32_t n, uint32_t o, uint32_t p) {   // (the caller has checked l + n <= room)
    const bool mine = lane == m;
    ll = mine ? l : ll; ml = mine ? n : ml; off = mine ? o : off; lp = mine ? p : lp;
    room -= l + n;
    m += 1;
  }
  // -> false: the sequence does not fit the page
  __device__ __forceinline__ bool push(uint32_t lane, uint32_t l, uint32_t n, uint32_t o, uint32_t p) {
    if (l > room || n > room - l) return false;
    put(lane, l, n, o, p);
    return true;
  }
  __device__ __forceinline__ bool run(ZWave& w, FwdStream& lits) {
    uint32_t unused;
    const bool ok = m == 0 || w.replay<false>(lits, m, ll, ml, off, lp, false, unused);
    begin(w);
    return ok;
  }
};

// -> false: malformed
__device__ __forceinline__ bool lz4_block(ZWave& w, FwdStream& in) {
  uint32_t p = 0;
  const uint32_t n = w.in_len, a0 = w.a0;
  const int32_t plim = (int32_t)n - 24;
  FwdStream lits = in;
  LzBatch B;
  B.begin(w);
  for (;;) {
    // the common token, as long as it lasts
    const uint32_t kbase = in.k << 8;   // (the chunk the stream holds: it only moves in the general code below)
    for (;;) {
      const uint32_t a = a0 + p;
      // a batch is full / fewer than 24 bytes are left / the 8 bytes at `a` do not lie in dwords 0..63 of the chunk
      if (B.m >= 64) break;
      if ((int32_t)p > plim) break;
      if (a - kbase > 244u) break;
      const uint32_t i = (a & 255u) >> 2;
      const uint32_t d0 = rdl(in.cur, i), d1 = rdl(in.cur, i + 1), d2 = rdl(in.cur, i + 2);
      const uint32_t sft = 8 * (a & 3u);
      const uint64_t h = (((uint64_t)d0 | ((uint64_t)d1 << 32)) >> sft) | (((uint64_t)d2 << 1) << (63 - sft));
      const uint32_t lit = ((uint32_t)h >> 4) & 15u, ml = (uint32_t)h & 15u;
      if (lit > 5 || ml == 15 || lit + ml + 4 > B.room) break;
      B.put(w.lane, lit, ml + 4, (uint32_t)(h >> (8 + 8 * lit)) & 0xFFFF, a + 1);
      p += lit + 3;
    }
    if (B.m >= 64) {
      if (!B.run(w, lits)) return false;
      continue;
    }
    if (p >= n) break;
    // one sequence the general way
    const uint64_t h = in.u64(a0 + p);
    const uint32_t tok = (uint32_t)h & 0xFF;
    uint32_t lit = tok >> 4, ml = tok & 15u;
    p += 1;
    if (lit == 15) {
      for (;;) {
        if (p >= n) return false;
        const uint32_t b = (uint32_t)in.u64(a0 + p) & 0xFF;
        p += 1;
        if (lit > 0x7FFFFFFFu - b) return false;
        lit += b;
        if (b != 255) break;
      }
    }
    if (lit > n - p) return false;
    const uint32_t lp = a0 + p;
    p = rfl(p + lit);
    if (p >= n) {              // the last sequence ends with its literals
      if (lit && !B.push(w.lane, lit, 0, 1, lp)) return false;
      break;
    }
    if (n - p < 2) return false;
    const uint32_t off = (uint32_t)in.u64(a0 + p) & 0xFFFF;
    p += 2;
    if (ml == 15) {
      for (;;) {
        if (p >= n) return false;
        const uint32_t b = (uint32_t)in.u64(a0 + p) & 0xFF;
        p += 1;
        if (ml > 0x7FFFFFFFu - b - 4) return false;
        ml += b;
        if (b != 255) break;
      }
    }
    if (!B.push(w.lane, lit, ml + 4, off, lp)) return false;
  }
  return B.run(w, lits);
}

__device__ __forceinline__ bool snappy_raw(ZWave& w, FwdStream& in) {
  uint32_t p = 0;
  const uint32_t n = w.in_len, a0 = w.a0;
  // preamble: the uncompressed length as a varint
  uint64_t total = 0;
  bool fin = false;
  for (int k = 0; k < 5 && p < n; ++k) {
    const uint32_t b = (uint32_t)in.u64(a0 + p) & 0xFF;
    p += 1;
    total |= (uint64_t)(b & 0x7F) << (7 * k);
    if (!(b & 0x80)) { fin = true; break; }
  }
  if (!fin || total != (uint64_t)w.cap_) return false;
  // A sequence = a literal element (possibly none) + the copy element behind it (possibly none). Parsed ahead like LZ4's: the common
  // pair (a literal of <= 4 bytes, a copy with a one- or two-byte offset, all inside the 8 bytes at the tag) in a loop of its own.
  const int32_t plim = (int32_t)n - 24;
  FwdStream lits = in;
  LzBatch B;
  B.begin(w);
  for (;;) {
    const uint32_t kbase = in.k << 8;
    for (;;) {
      const uint32_t a = a0 + p;
      if (B.m >= 64) break;
      if ((int32_t)p > plim) break;
      if (a - kbase > 244u) break;
      const uint32_t i = (a & 255u) >> 2;
      const uint32_t d0 = rdl(in.cur, i), d1 = rdl(in.cur, i + 1), d2 = rdl(in.cur, i + 2);
      const uint32_t sft = 8 * (a & 3u);
      const uint64_t h = (((uint64_t)d0 | ((uint64_t)d1 << 32)) >> sft) | (((uint64_t)d2 << 1) << (63 - sft));
      const uint32_t tag = (uint32_t)h & 0xFF;
      const bool has_lit = (tag & 3u) == 0;
      const uint32_t ll = has_lit ? (tag >> 2) + 1 : 0u;          // (> 4: not this loop's)
      const uint32_t adv = has_lit ? 1 + ll : 0u;
      if (ll > 4) break;
      const uint32_t hb = (uint32_t)(h >> (8 * adv));              // the copy's tag and the two bytes behind it
      const uint32_t t2 = hb & 0xFF, k2 = t2 & 3u;
      if (k2 == 0 || k2 == 3) break;
      const uint32_t mlen = k2 == 1 ? ((t2 >> 2) & 7u) + 4 : (t2 >> 2) + 1;
      const uint32_t off = k2 == 1 ? ((t2 >> 5) << 8) | ((hb >> 8) & 0xFF) : (hb >> 8) & 0xFFFF;
      if (off == 0 || ll + mlen > B.room) break;
      B.put(w.lane, ll, mlen, off, a + 1);
      p += adv + 1 + k2;
    }
    if (B.m >= 64) {
      if (!B.run(w, lits)) return false;
      continue;
    }
    if (p >= n) break;
    // one sequence the general way
    uint64_t h = in.u64(a0 + p);
    uint32_t tag = (uint32_t)h & 0xFF, ll = 0, lp = 0;
    if ((tag & 3u) == 0) {
      uint32_t len = (tag >> 2) + 1, hdr = 1;
      if (len > 60) {
        const uint32_t extra = len - 60;   // 1..4 length bytes
        if (n - p < 1 + extra) return false;
        const uint32_t v = (uint32_t)(h >> 8) & (extra == 4 ? 0xFFFFFFFFu : ((1u << (8 * extra)) - 1));
        if (v == 0xFFFFFFFFu) return false;
        len = v + 1;
        hdr = 1 + extra;
      }
      if (len > n - p - hdr) return false;
      lp = a0 + p + hdr;
      ll = len;
      p = rfl(p + hdr + len);
      bool lone = p >= n;                  // nothing behind it,
      if (!lone) {
        h = in.u64(a0 + p);
        tag = (uint32_t)h & 0xFF;
        lone = (tag & 3u) == 0;            // or another literal: a sequence without a copy
      }
      if (lone) {
        if (!B.push(w.lane, ll, 0, 1, lp)) return false;
        continue;
      }
    }
    const uint32_t kind = tag & 3u, left = n - p;
    uint32_t len, off;
    if (kind == 1) {
      if (left < 2) return false;
      len = ((tag >> 2) & 7u) + 4; off = ((tag >> 5) << 8) | ((uint32_t)(h >> 8) & 0xFF);
      p += 2;
    } else if (kind == 2) {
      if (left < 3) return false;
      len = (tag >> 2) + 1; off = (uint32_t)(h >> 8) & 0xFFFF;
      p += 3;
    } else {
      if (left < 5) return false;
      len = (tag >> 2) + 1; off = (uint32_t)(h >> 8);
      p += 5;
    }
    if (off == 0) return false;
    if (!B.push(w.lane, ll, len, off, lp)) return false;
  }
  return B.run(w, lits);
}

}  // namespace

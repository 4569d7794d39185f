// dv_wave.h — one wave decompresses one page (k_parquet_dev.hip): how the wave reads its input and where its output lives.
//
//   input    a 256-byte look-ahead WINDOW of the compressed payload held in one VGPR (lane l = dword l): headers, FSE / Huffman table
//            descriptions and the backward bitstream of the sequences are read with v_readlane (scalar results, no memory round trip per
//            field); the window slides forward or backward with one coalesced 256-byte load.
//   output   an LDS RING of the last W bytes (W = 8 KiB for ZSTD, whose tables take another 14 KiB of LDS) written to the HBM image in
//            16-byte stores; a back-reference that reaches further than the ring reads the image itself — those bytes left the ring, so
//            they were flushed long ago (the wave waits for its own stores once per far reference, nothing else).
// With 22 KiB of LDS per wave seven pages are resident per CU (the round-4 kernel kept a 64 KiB window + 8 KiB input ring: two).
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

struct ZWave {
  // ---- input
  const uint8_t* srcA;   // 4-byte aligned address at or below the payload
  uint32_t a0;           // payload start - srcA
  uint32_t in_len;
  uint32_t safeA;        // bytes readable from srcA, a multiple of 4
  uint32_t la, wlo;      // the window: this lane holds bytes [wlo + 4 lane, + 4) from srcA; wlo is a multiple of 4
  // ---- output
  uint8_t* dst;
  uint32_t cap_, op_, flushed, sh, frame0;
  uint8_t* win;
  uint32_t WM, WF;       // ring mask / flush threshold
  bool stores_pending;   // ring bytes were stored to the image since the wave last waited for its stores
  // ---- LDS tables
  uint8_t* tab;
  uint32_t lane;

  __device__ __forceinline__ bool lead() const { return lane == 0; }
  __device__ __forceinline__ void sync() const { __builtin_amdgcn_wave_barrier(); }
  __device__ __forceinline__ bool bcast(bool b) const { return __builtin_amdgcn_readfirstlane((int)b) != 0; }
  __device__ __forceinline__ uint32_t uni(uint32_t v) const { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
  __device__ __forceinline__ uint64_t uni64(uint64_t v) const {
    return (uint64_t)uni((uint32_t)v) | ((uint64_t)uni((uint32_t)(v >> 32)) << 32);
  }
  __device__ __forceinline__ uint16_t* huf() const { return (uint16_t*)tab; }
  __device__ __forceinline__ uint64_t* llt() const { return (uint64_t*)(tab + 4096); }
  __device__ __forceinline__ uint64_t* mlt() const { return (uint64_t*)(tab + 8192); }
  __device__ __forceinline__ uint32_t* oft() const { return (uint32_t*)(tab + 12288); }
  __device__ __forceinline__ uint8_t* scr() const { return tab + 13312; }
  __device__ __forceinline__ uint32_t op() const { return op_; }
  __device__ __forceinline__ uint32_t cap() const { return cap_; }
  __device__ __forceinline__ void frame_begin() { frame0 = op_; }

  // ---- the window
  __device__ __forceinline__ void slide(uint32_t wl) {
    wlo = wl;
    const uint32_t o = wl + 4 * lane;
    la = (o + 4 <= safeA) ? *(const uint32_t*)(srcA + o) : 0u;
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
    const uint32_t a = uni(pos + a0);
    ensure(a, 1);
    const uint32_t d = (uint32_t)__builtin_amdgcn_readlane((int)la, (int)((a - wlo) >> 2));
    return (d >> (8 * (a & 3))) & 0xFF;
  }
  __device__ __forceinline__ uint64_t in64(uint32_t pos) {
    const uint32_t a = uni(pos + a0);
    ensure(a, 8);
    const uint32_t i = (a - wlo) >> 2;
    const uint32_t d0 = (uint32_t)__builtin_amdgcn_readlane((int)la, (int)i);
    const uint32_t d1 = (uint32_t)__builtin_amdgcn_readlane((int)la, (int)(i + 1));
    const uint32_t i2 = i + 2 < 64 ? i + 2 : 63;
    const uint32_t d2 = (uint32_t)__builtin_amdgcn_readlane((int)la, (int)i2);
    const uint64_t lo = (uint64_t)d0 | ((uint64_t)d1 << 32);
    const uint32_t s = 8 * (a & 3);
    return s ? (lo >> s) | ((uint64_t)d2 << (64 - s)) : lo;
  }
  // per-lane reads of the payload (the Huffman streams: up to four lanes, each at its own address)
  __device__ __forceinline__ uint32_t lane_in8(uint32_t pos) const { return pos + a0 < safeA ? srcA[a0 + pos] : 0u; }
  __device__ __forceinline__ uint64_t lane_in64(uint32_t pos) const {
    const uint8_t* p = srcA + a0 + pos;
    uint64_t v = 0;
    if (pos + a0 + 8 <= safeA) {
      __builtin_memcpy(&v, p, 8);
    } else {
      for (uint32_t k = 0; k < 8; ++k)
        if (pos + a0 + k < safeA) v |= (uint64_t)p[k] << (8 * k);
    }
    return v;
  }
  __device__ __forceinline__ void lane_store(uint32_t p, uint8_t b) const {
    if (p < cap_) dst[p] = b;
  }

  // ---- the ring
  __device__ __forceinline__ void flush(bool force) {
    const uint32_t target = op_;
    const uint32_t a = (flushed + sh) & 15u;
    if (a && flushed < target) {
      const uint32_t h = (16 - a) < (target - flushed) ? (16 - a) : (target - flushed);
      if (lane < h) dst[flushed + lane] = win[(flushed + sh + lane) & WM];
      flushed += h;
    }
    const uint32_t nvec = (target - flushed) >> 4;
    if (((flushed + sh) & 15u) == 0 && nvec) {
#pragma clang loop unroll(disable)
      for (uint32_t v = lane; v < nvec; v += 64)
        *(uint4*)(dst + flushed + 16 * v) = *(const uint4*)(win + ((flushed + sh + 16 * v) & WM));
      flushed += nvec << 4;
    }
    if (force) {
      for (uint32_t i = flushed + lane; i < target; i += 64) dst[i] = win[(i + sh) & WM];
      flushed = target;
    }
    stores_pending = true;
  }
  __device__ __forceinline__ void wait_stores() {
    if (stores_pending) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      stores_pending = false;
    }
  }
  __device__ __forceinline__ void advance(uint32_t n) {
    op_ += n;
    if (op_ - flushed >= WF) flush(false);
  }

  // `len` bytes from global memory at g (the payload, or the literals the Huffman stage left in the image's tail)
  __device__ __forceinline__ void copy_in(const uint8_t* g, uint32_t len) {
#pragma clang loop unroll(disable)
    for (uint32_t i = 0; i < len; i += 256) {
      const uint32_t n = len - i < 256 ? len - i : 256;
      uint8_t b[4];
#pragma unroll
      for (uint32_t k = 0; k < 4; ++k) b[k] = (64 * k + lane < n) ? g[i + 64 * k + lane] : (uint8_t)0;
#pragma unroll
      for (uint32_t k = 0; k < 4; ++k)
        if (64 * k + lane < n) win[(op_ + 64 * k + lane + sh) & WM] = b[k];
      __builtin_amdgcn_wave_barrier();
      advance(n);
    }
  }
  __device__ __forceinline__ bool put_in(uint32_t pos, uint32_t len) {
    if (len > cap_ - op_ || pos > in_len || len > in_len - pos) return false;
    copy_in(srcA + a0 + pos, len);
    return true;
  }
  __device__ __forceinline__ bool put_out(uint32_t pos, uint32_t len) {
    if (len > cap_ - op_ || pos > cap_ || len > cap_ - pos) return false;
    copy_in(dst + pos, len);   // (huf_streams waited for the stores that put them there)
    return true;
  }
  __device__ __forceinline__ bool put_fill(uint32_t byte, uint32_t len) {
    if (len > cap_ - op_) return false;
#pragma clang loop unroll(disable)
    for (uint32_t i = 0; i < len; i += 64) {
      const uint32_t n = len - i < 64 ? len - i : 64;
      if (lane < n) win[(op_ + lane + sh) & WM] = (uint8_t)byte;
      __builtin_amdgcn_wave_barrier();
      advance(n);
    }
    return true;
  }
  __device__ __forceinline__ bool put_match(uint32_t off, uint32_t len) {
    if (off == 0 || off > op_ - frame0 || len > cap_ - op_) return false;
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
      // the source left the ring: it is in the image (flushed at least WF + 64 bytes ago: WF <= W - 256)
      wait_stores();
#pragma clang loop unroll(disable)
      for (uint32_t i = 0; i < len; i += 64) {
        const uint32_t n = len - i < 64 ? len - i : 64;
        if (lane < n) win[(op_ + lane + sh) & WM] = dst[op_ - off + lane];
        __builtin_amdgcn_wave_barrier();
        advance(n);
        wait_stores();   // (a flush inside advance(): the next piece may read what it stored — only when off < 64 + WF, never here, but cheap)
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
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the literals are read back (put_out) by other lanes
    stores_pending = false;
    return all;
  }
};

constexpr uint32_t ZW_RING = 8192;                          // ZSTD: ring bytes
constexpr uint32_t ZW_TABLES = 13312 + zc::SCR_BYTES;       // Huffman 4 KiB + LL 4 KiB + ML 4 KiB + OF 1 KiB + scratch
constexpr uint32_t ZW_LDS = ZW_RING + ZW_TABLES;

}  // namespace

// dev_scan.h — device-wide exclusive scan of u32 counts into u64 offsets
// (block sums -> single-block scan of the sums -> apply), shared by join and sort.
#pragma once
#include "dev_common.h"
#include "runtime.h"

namespace dbscan {

// exclusive scan of u32 counts into u64 offsets: block sums, single-block scan, apply
#define SCAN_TILE 1024
static __global__ __launch_bounds__(256) void scan_block_sums(const uint32_t* cnt, int64_t n, uint64_t* blk) {
  int64_t base = (int64_t)blockIdx.x * SCAN_TILE;
  uint64_t s = 0;
  for (int k = threadIdx.x; k < SCAN_TILE; k += 256)
    if (base + k < n) s += cnt[base + k];
  s = wave_sum_u64(s);
  __shared__ uint64_t part[4];
  if (lane_id() == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) blk[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}
static __global__ __launch_bounds__(1024) void scan_blk_kernel(uint64_t* blk, int64_t nblk) {
  __shared__ uint64_t wave_tot[16];
  __shared__ uint64_t carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int64_t base = 0; base < nblk; base += 1024) {
    int64_t i = base + threadIdx.x;
    uint64_t v = i < nblk ? blk[i] : 0, incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      uint64_t t = __shfl_up(incl, d, 64);
      if (lane_id() >= d) incl += t;
    }
    if (lane_id() == 63) wave_tot[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint64_t wbase = 0;
    for (int k = 0; k < (threadIdx.x >> 6); ++k) wbase += wave_tot[k];
    uint64_t c = carry;
    if (i < nblk) blk[i] = c + wbase + incl - v;
    __syncthreads();
    if (threadIdx.x == 1023) carry = c + wbase + incl;
    __syncthreads();
  }
}
static __global__ __launch_bounds__(256) void scan_apply_kernel(const uint32_t* cnt, int64_t n, const uint64_t* blk,
                                                         uint64_t* off) {
  // one wave per 64-element group, 16 groups per tile handled sequentially by wave 0..3
  int64_t base = (int64_t)blockIdx.x * SCAN_TILE;
  __shared__ uint64_t run;
  if (threadIdx.x == 0) run = blk[blockIdx.x];
  __syncthreads();
  for (int chunk = 0; chunk < SCAN_TILE; chunk += 256) {
    int64_t i = base + chunk + threadIdx.x;
    uint64_t v = i < n ? cnt[i] : 0, incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      uint64_t t = __shfl_up(incl, d, 64);
      if (lane_id() >= d) incl += t;
    }
    __shared__ uint64_t wt[4];
    if (lane_id() == 63) wt[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint64_t wbase = 0;
    for (int k = 0; k < (threadIdx.x >> 6); ++k) wbase += wt[k];
    uint64_t r = run;
    if (i < n) off[i] = r + wbase + incl - v;
    __syncthreads();
    if (threadIdx.x == 255) run = r + wbase + incl;
    __syncthreads();
  }
}


// counts[n] -> offsets[n]; blk must hold ceil(n / SCAN_TILE) + 1 u64
inline int32_t exclusive_scan_u32(const uint32_t* cnt, int64_t n, uint64_t* blk, uint64_t* off, hipStream_t s) {
  if (n == 0) return DBHIP_OK;
  const int64_t nblk = dbhip::ceil_div(n, SCAN_TILE);
  hipLaunchKernelGGL(scan_block_sums, dim3((unsigned)nblk), dim3(256), 0, s, cnt, n, blk);
  hipLaunchKernelGGL(scan_blk_kernel, dim3(1), dim3(1024), 0, s, blk, nblk);
  hipLaunchKernelGGL(scan_apply_kernel, dim3((unsigned)nblk), dim3(256), 0, s, cnt, n, blk, off);
  DBHIP_LAUNCH_CHECK();
  return DBHIP_OK;
}

}  // namespace dbscan

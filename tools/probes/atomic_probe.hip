// atomic_probe.hip — what do read-modify-write atomics on an L2-resident table cost on gfx950, by scope?
// (experiment behind the XCD-local aggregation tables; built by tools/probes/build.sh, run on the GPU box)
//   variant 0: ONE table, agent-scope atomics (what gb_accum_kernel does today)
//   variant 1: one table PER XCD (selected with HW_REG_XCC_ID), workgroup-scope atomics (no sc1: resolved in the XCD's L2)
//   variant 2: like 1, plus the slot's key word read with an agent-scope (sc1, L2-served) load first — the probe of a real table
// rows = 60 M, per row: u32 key in [0, G), u64 value; table slot = 4 words {key, hash, sum, count}.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ uint32_t xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | 20) & 15u; }   // HW_REG_XCC_ID, bits 0..3

template <int VARIANT>
__global__ __launch_bounds__(256) void agg_kernel(const uint32_t* __restrict__ keys, const uint64_t* __restrict__ vals, int64_t n, uint64_t* table,
                                                   int64_t slots, unsigned long long* bad) {
  uint64_t* t = table;
  if (VARIANT >= 1) t = table + (uint64_t)xcc_id() * slots * 4;
  uint64_t mism = 0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const uint32_t k = keys[i];
    const uint64_t v = vals[i];
    uint64_t* s = t + (uint64_t)k * 4;
    if (VARIANT == 2) {
      const uint64_t have = __hip_atomic_load(s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      mism += have != (uint64_t)k;
    }
    if (VARIANT == 0) {
      __hip_atomic_fetch_add(s + 2, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_fetch_add(s + 3, 1ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      __hip_atomic_fetch_add(s + 2, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      __hip_atomic_fetch_add(s + 3, 1ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  }
  if (mism) atomicAdd(bad, (unsigned long long)mism);
}

int main(int argc, char** argv) {
  const int64_t n = argc > 1 ? atoll(argv[1]) : 60000000;
  CK(hipSetDevice(0));
  std::vector<uint32_t> hk((size_t)n);
  std::vector<uint64_t> hv((size_t)n);
  uint32_t* dk; uint64_t* dv; unsigned long long* bad;
  CK(hipMalloc(&dk, n * 4)); CK(hipMalloc(&dv, n * 8)); CK(hipMalloc(&bad, 8));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int64_t Gs[] = {1000, 10000, 100000, 1000000, 4};
  for (int64_t G : Gs) {
    uint64_t st = 88172645463325252ULL;
    uint64_t exp_sum = 0;
    for (int64_t i = 0; i < n; ++i) {
      st ^= st << 13; st ^= st >> 7; st ^= st << 17;
      hk[i] = (uint32_t)(st % (uint64_t)G);
      hv[i] = (st >> 20) & 0xFFFFF;
      exp_sum += hv[i];
    }
    CK(hipMemcpy(dk, hk.data(), n * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dv, hv.data(), n * 8, hipMemcpyHostToDevice));
    uint64_t* tab;
    const size_t tbytes = (size_t)G * 4 * 8 * 8;
    CK(hipMalloc(&tab, tbytes));
    std::vector<uint64_t> init((size_t)G * 4 * 8, 0);
    for (int x = 0; x < 8; ++x) for (int64_t g = 0; g < G; ++g) init[((size_t)x * G + g) * 4] = (uint64_t)g;
    for (int variant = 0; variant < 3; ++variant) {
      float best = 1e9f;
      uint64_t got_sum = 0, got_cnt = 0;
      unsigned long long hb = 0;
      for (int rep = 0; rep < 3; ++rep) {
        CK(hipMemcpy(tab, init.data(), tbytes, hipMemcpyHostToDevice));
        CK(hipMemset(bad, 0, 8));
        CK(hipEventRecord(e0, 0));
        if (variant == 0) hipLaunchKernelGGL(agg_kernel<0>, dim3(2048), dim3(256), 0, 0, dk, dv, n, tab, G, bad);
        else if (variant == 1) hipLaunchKernelGGL(agg_kernel<1>, dim3(2048), dim3(256), 0, 0, dk, dv, n, tab, G, bad);
        else hipLaunchKernelGGL(agg_kernel<2>, dim3(2048), dim3(256), 0, 0, dk, dv, n, tab, G, bad);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
        std::vector<uint64_t> out((size_t)G * 4 * 8);
        CK(hipMemcpy(out.data(), tab, tbytes, hipMemcpyDeviceToHost));
        CK(hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost));
        got_sum = 0; got_cnt = 0;
        for (size_t s = 0; s < (size_t)G * 8; ++s) { got_sum += out[s * 4 + 2]; got_cnt += out[s * 4 + 3]; }
      }
      printf("{\"groups\": %lld, \"variant\": %d, \"ms_per_%lldM_rows\": %.3f, \"sum_ok\": %d, \"count_ok\": %d, \"key_mismatches\": %llu}\n", (long long)G, variant,
             (long long)(n / 1000000), best, (int)(got_sum == exp_sum), (int)(got_cnt == (uint64_t)n), hb);
      fflush(stdout);
    }
    CK(hipFree(tab));
  }
  return 0;
}

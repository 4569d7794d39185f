// round 6 probe: what a small pinned-host -> device hipMemcpyAsync costs the CALLING thread when T host threads each keep a stream busy
// (the pipelined fused aggregation uploads one block table per multi-block launch). hipcc --offload-arch=gfx950 -O2 -o h2d_small_copy h2d_small_copy.hip -lpthread
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>
__global__ void spin_kernel(uint64_t* p, int iters) {
  uint64_t x = threadIdx.x;
  for (int i = 0; i < iters; ++i) x = x * 6364136223846793005ULL + 1442695040888963407ULL;
  if (x == 42) p[0] = x;
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  for (int T : {1, 8})
    for (size_t bytes : {(size_t)4096, (size_t)16384, (size_t)32768, (size_t)65536}) {
      std::vector<double> host_us(T, 0.0), wall_ms(T, 0.0);
      std::vector<std::thread> th;
      std::atomic<int> ready{0}; std::atomic<bool> go{false};
      for (int t = 0; t < T; ++t) th.emplace_back([&, t] {
        hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
        void *h, *d; uint64_t* dummy;
        hipHostMalloc(&h, bytes, hipHostMallocDefault); hipMalloc(&d, bytes); hipMalloc((void**)&dummy, 64);
        hipEvent_t ev; hipEventCreateWithFlags(&ev, hipEventDisableTiming);
        for (int w = 0; w < 5; ++w) { hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, s); hipLaunchKernelGGL(spin_kernel, dim3(256), dim3(256), 0, s, dummy, 2000); }
        hipStreamSynchronize(s);
        ++ready; while (!go.load()) {}
        const double t0 = now(); double inside = 0;
        for (int i = 0; i < 50; ++i) {
          const double a = now();
          hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, s);
          hipEventRecord(ev, s);
          inside += now() - a;
          hipLaunchKernelGGL(spin_kernel, dim3(256), dim3(256), 0, s, dummy, 2000);
        }
        hipStreamSynchronize(s);
        wall_ms[t] = (now() - t0) * 1e3; host_us[t] = inside * 1e6 / 50;
      });
      while (ready.load() < T) {}
      go.store(true);
      for (auto& x : th) x.join();
      double hu = 0, wm = 0; for (int t = 0; t < T; ++t) { hu += host_us[t] / T; wm = wall_ms[t] > wm ? wall_ms[t] : wm; }
      printf("threads %d  copy %6zu B : %8.2f us inside hipMemcpyAsync+hipEventRecord per call, %8.3f ms wall for 50 (copy, kernel) pairs\n", T, bytes, hu, wm);
    }
  return 0;
}

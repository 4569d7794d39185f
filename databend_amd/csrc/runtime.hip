// runtime.hip — device binding, memory, streams, events, error reporting.
// There is deliberately no CPU fallback anywhere in this library: without a
// visible gfx950 device dbhip_init fails and every other entry point fails too.
#include "runtime.h"

#include <stdarg.h>
#include <string.h>

#include <stdlib.h>

#include <sys/syscall.h>
#include <unistd.h>

#include <atomic>
#include <map>
#include <mutex>
#include <new>
#include <unordered_map>
#include <vector>

namespace dbhip {

static thread_local char g_err[512] = "";
static hipStream_t g_stream = nullptr;
static int g_device = -1;
static std::mutex g_mu;

struct Scratch {
  void* p = nullptr;
  size_t cap = 0;
};
// Scratch is keyed by (thread, stream): two asynchronous calls of one thread on two streams never share a buffer, so a thread
// may keep several streams busy at once. A buffer that has to grow is released after ITS stream has drained, not after the whole
// device. The registry is process-wide (one mutex, taken per lookup and never held across a HIP call that can wait) so that the
// entries have an owner who can release them:
//   * dbhip_stream_destroy / dbhip_stream_release_scratch free every thread's entry of that stream (after draining it);
//   * a thread that exits frees its entries (thread_local sentinel below; the main thread at process exit leaves it to the OS —
//     the HIP runtime may already be shutting down);
//   * a thread holds at most SCRATCH_STREAMS_PER_THREAD entries: the least recently used one is evicted (device drained first),
//     so a host that takes a fresh stream per query — or torch's stream pool — cannot grow device memory without bound, and a
//     recycled hipStream_t value never inherits more than that.
struct StreamScratch {
  Scratch slot[24];
  uint64_t last_use = 0;
};
constexpr size_t SCRATCH_STREAMS_PER_THREAD = 8;
struct ScratchKey {
  uint64_t tid;
  hipStream_t stream;
  bool operator<(const ScratchKey& o) const { return tid != o.tid ? tid < o.tid : (uintptr_t)stream < (uintptr_t)o.stream; }
};
static std::mutex g_scratch_mu;
static std::map<ScratchKey, StreamScratch>* g_scratch = nullptr;   // (leaked on purpose: no static destructor races a late call)
static std::atomic<uint64_t> g_scratch_clock{0};
static std::atomic<uint64_t> g_next_tid{1};

static void free_entry(StreamScratch& e) {
  for (Scratch& s : e.slot) {
    if (s.p) (void)hipFree(s.p);
    s.p = nullptr;
    s.cap = 0;
  }
}
// every entry whose key matches (tid == 0: any thread; any_stream: every stream of that thread); caller holds no lock
static void release_scratch_entries(uint64_t tid, hipStream_t stream, bool any_stream) {
  std::lock_guard<std::mutex> lk(g_scratch_mu);
  if (!g_scratch) return;
  for (auto it = g_scratch->begin(); it != g_scratch->end();) {
    const bool hit = (tid == 0 || it->first.tid == tid) && (any_stream || it->first.stream == stream);
    if (hit) { free_entry(it->second); it = g_scratch->erase(it); } else ++it;
  }
}
struct ThreadScratchOwner {
  uint64_t tid = g_next_tid.fetch_add(1);
  bool is_main = false;
  ThreadScratchOwner() { is_main = (long)getpid() == (long)syscall(SYS_gettid); }
  ~ThreadScratchOwner() {
    if (is_main) return;   // process exit: the driver reclaims everything, and HIP may be tearing down
    (void)hipDeviceSynchronize();
    release_scratch_entries(tid, nullptr, true);
  }
};
static thread_local ThreadScratchOwner t_scratch_owner;

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int grid_cap_override() {
  static const int cap = [] { const char* e = exp_env("DBHIP_GRID_CAP"); int v = e ? atoi(e) : 0; return v > 0 ? v : 0; }();
  return cap;
}

int32_t hip_fail(hipError_t e, const char* what) {
  set_error("HIP error %d (%s) in %s", (int)e, hipGetErrorString(e), what);
  return DBHIP_ERR_HIP;
}

hipStream_t resolve_stream(void* stream) {
  if (stream) return (hipStream_t)stream;
  return g_stream;
}

// The map is only looked up / changed under g_scratch_mu; the device work (synchronise, free, allocate) happens AFTER the lock is
// dropped, on state only this thread uses: an entry belongs to the (thread, stream) pair of its key, std::map nodes do not move,
// and the one cross-thread path — dbhip_stream_release_scratch / dbhip_stream_destroy — requires that no thread is inside a dbhip
// call on that stream (dbhip.h). A host thread that grows a buffer behind a long kernel therefore no longer blocks every other
// thread's scratch() call.
void* scratch(size_t bytes, int slot, hipStream_t stream) {
  const uint64_t tid = t_scratch_owner.tid;
  Scratch* sp = nullptr;
  StreamScratch evicted;
  bool have_evicted = false;
  {
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    if (!g_scratch) g_scratch = new (std::nothrow) std::map<ScratchKey, StreamScratch>();
    if (!g_scratch) { set_error("scratch: out of host memory"); return nullptr; }
    const ScratchKey key{tid, stream};
    auto it = g_scratch->find(key);
    if (it == g_scratch->end()) {
      // a new (thread, stream) pair: evict this thread's least recently used entry once it holds too many
      size_t mine = 0;
      auto lru = g_scratch->end();
      for (auto j = g_scratch->lower_bound(ScratchKey{tid, nullptr}); j != g_scratch->end() && j->first.tid == tid; ++j) {
        ++mine;
        if (lru == g_scratch->end() || j->second.last_use < lru->second.last_use) lru = j;
      }
      if (mine >= SCRATCH_STREAMS_PER_THREAD && lru != g_scratch->end()) {
        evicted = lru->second;   // (plain pointers: freed below, outside the lock)
        have_evicted = true;
        g_scratch->erase(lru);
      }
      it = g_scratch->emplace(key, StreamScratch()).first;
    }
    it->second.last_use = g_scratch_clock.fetch_add(1) + 1;
    sp = &it->second.slot[slot];
  }
  if (have_evicted) {
    (void)hipDeviceSynchronize();   // (the evicted stream may be gone already: drain the device, not the stream)
    free_entry(evicted);
  }
  Scratch& s = *sp;
  if (s.cap < bytes) {
    if (s.p) {
      (void)hipStreamSynchronize(stream);
      (void)hipFree(s.p);
    }
    size_t cap = bytes + (bytes >> 2) + 4096;
    if (hipMalloc(&s.p, cap) != hipSuccess) {
      s.p = nullptr;
      s.cap = 0;
      set_error("scratch allocation of %zu bytes failed", cap);
      return nullptr;
    }
    s.cap = cap;
  }
  return s.p;
}

void release_stream_scratch(hipStream_t stream) { release_scratch_entries(0, stream, false); }

// ---- cancellation marks (dbhip_stream_cancel) ----
static std::atomic<int> g_cancel_any{0};
static std::mutex g_cancel_mu;
static std::unordered_map<hipStream_t, int>* g_cancelled = nullptr;
bool cancel_requested(hipStream_t stream) {
  if (g_cancel_any.load(std::memory_order_relaxed) == 0) return false;
  std::lock_guard<std::mutex> lk(g_cancel_mu);
  return g_cancelled && g_cancelled->count(stream) != 0;
}
static void set_cancel(hipStream_t stream, bool on) {
  std::lock_guard<std::mutex> lk(g_cancel_mu);
  if (!g_cancelled) g_cancelled = new (std::nothrow) std::unordered_map<hipStream_t, int>();
  if (!g_cancelled) return;
  if (on) (*g_cancelled)[stream] = 1; else g_cancelled->erase(stream);
  g_cancel_any.store((int)g_cancelled->size(), std::memory_order_relaxed);
}

static thread_local uint64_t* g_pinned = nullptr;
uint64_t* pinned_words(int slot) {
  if (!g_pinned) {
    void* p = nullptr;
    if (hipHostMalloc(&p, 8 * 64 * sizeof(uint64_t), hipHostMallocDefault) != hipSuccess) return nullptr;
    g_pinned = (uint64_t*)p;
  }
  return g_pinned + (slot & 7) * 64;
}

static thread_local hipEvent_t g_t0 = nullptr, g_t1 = nullptr;
static thread_local bool g_timed = false;

void kernel_timer_start(hipStream_t s) {
  if (!g_t0) {
    if (hipEventCreate(&g_t0) != hipSuccess || hipEventCreate(&g_t1) != hipSuccess) { g_t0 = g_t1 = nullptr; return; }
  }
  (void)hipEventRecord(g_t0, s);
  g_timed = false;
}

void kernel_timer_stop(hipStream_t s) {
  if (!g_t1) return;
  (void)hipEventRecord(g_t1, s);
  g_timed = true;
}

}  // namespace dbhip

using namespace dbhip;

extern "C" {

int32_t dbhip_abi_version(void) { return DBHIP_ABI_VERSION; }

const char* dbhip_last_error(void) { return g_err; }

int32_t dbhip_device_count(int32_t* out_count_host) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) n = 0;
  if (out_count_host) *out_count_host = n;
  return DBHIP_OK;
}

int32_t dbhip_init(int32_t device) {
  std::lock_guard<std::mutex> lk(g_mu);
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) {
    set_error("no HIP device visible (hipGetDeviceCount -> %d, n=%d): libdbhip has no CPU fallback",
              (int)e, n);
    return DBHIP_ERR_NO_DEVICE;
  }
  if (device < 0 || device >= n) {
    set_error("device %d out of range (0..%d)", device, n - 1);
    return DBHIP_ERR_INVALID;
  }
  DBHIP_CHECK(hipSetDevice(device));
  hipDeviceProp_t prop;
  DBHIP_CHECK(hipGetDeviceProperties(&prop, device));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    set_error("device %d is %s; this library is built for gfx950 only", device, prop.gcnArchName);
    return DBHIP_ERR_NO_DEVICE;
  }
  if (g_device != device || !g_stream) {
    if (g_stream) (void)hipStreamDestroy(g_stream);
    DBHIP_CHECK(hipStreamCreateWithFlags(&g_stream, hipStreamNonBlocking));
    g_device = device;
  }
  return DBHIP_OK;
}

// Column buffers come and go with every operator of a plan; hipMalloc/hipFree of GB-sized
// buffers cost milliseconds each, so freed blocks of >= 1 MiB are kept in size-class free
// lists (classes: 8 steps per power of two, <= 12.5 % internal slack) up to a byte budget
// (DBHIP_CACHE_BYTES, default 96 GiB of the 288 GB) and handed out again. A block is only
// re-used after the device went idle at free time (hipFree's own semantics), so re-use is
// safe for any stream. dbhip_trim() returns the cached blocks to the driver.
namespace {
struct AllocCache {
  std::mutex mu;
  std::unordered_map<void*, size_t> live;                 // ptr -> class bytes (cached classes only)
  std::unordered_map<size_t, std::vector<void*>> free_;  // class bytes -> blocks
  size_t cached = 0, budget = 0;
  bool init = false;
} g_cache;

size_t size_class(size_t bytes) {
  if (bytes < (1u << 20)) return 0;
  int top = 63 - __builtin_clzll((unsigned long long)bytes);
  size_t step = (size_t)1 << (top - 3);
  return (bytes + step - 1) / step * step;
}
}  // namespace

int32_t dbhip_alloc(size_t bytes, void** out) {
  DBHIP_REQUIRE(out, "dbhip_alloc: out is NULL");
  if (bytes == 0) bytes = 16;
  size_t cls = size_class(bytes);
  if (cls) {
    std::lock_guard<std::mutex> lk(g_cache.mu);
    if (!g_cache.init) {
      const char* e = getenv("DBHIP_CACHE_BYTES");
      g_cache.budget = e ? (size_t)strtoull(e, nullptr, 10) : ((size_t)96 << 30);
      g_cache.init = true;
    }
    auto it = g_cache.free_.find(cls);
    if (it != g_cache.free_.end() && !it->second.empty()) {
      *out = it->second.back();
      it->second.pop_back();
      g_cache.cached -= cls;
      g_cache.live[*out] = cls;
      return DBHIP_OK;
    }
    hipError_t e = hipMalloc(out, cls);
    if (e != hipSuccess) {  // give the cache back to the driver and retry once
      for (auto& kv : g_cache.free_) { for (void* p : kv.second) (void)hipFree(p); kv.second.clear(); }
      g_cache.cached = 0;
      DBHIP_CHECK(hipMalloc(out, cls));
    }
    g_cache.live[*out] = cls;
    return DBHIP_OK;
  }
  DBHIP_CHECK(hipMalloc(out, bytes));
  return DBHIP_OK;
}

int32_t dbhip_free(void* p) {
  if (!p) return DBHIP_OK;
  {
    std::unique_lock<std::mutex> lk(g_cache.mu);
    auto it = g_cache.live.find(p);
    if (it != g_cache.live.end()) {
      size_t cls = it->second;
      g_cache.live.erase(it);
      if (g_cache.cached + cls <= g_cache.budget) {
        lk.unlock();
        DBHIP_CHECK(hipDeviceSynchronize());  // nothing in flight may still touch the block
        lk.lock();
        g_cache.free_[cls].push_back(p);
        g_cache.cached += cls;
        return DBHIP_OK;
      }
    }
  }
  DBHIP_CHECK(hipFree(p));
  return DBHIP_OK;
}

int32_t dbhip_trim(void) {
  std::lock_guard<std::mutex> lk(g_cache.mu);
  DBHIP_CHECK(hipDeviceSynchronize());
  for (auto& kv : g_cache.free_) { for (void* p : kv.second) (void)hipFree(p); kv.second.clear(); }
  g_cache.cached = 0;
  return DBHIP_OK;
}

int32_t dbhip_memcpy_h2d(void* dst, const void* src, size_t bytes, void* stream) {
  if (bytes == 0) return DBHIP_OK;
  hipStream_t s = resolve_stream(stream);
  DBHIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, s));
  DBHIP_CHECK(hipStreamSynchronize(s));
  return DBHIP_OK;
}

int32_t dbhip_memcpy_d2h(void* dst, const void* src, size_t bytes, void* stream) {
  if (bytes == 0) return DBHIP_OK;
  hipStream_t s = resolve_stream(stream);
  DBHIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, s));
  DBHIP_CHECK(hipStreamSynchronize(s));
  return DBHIP_OK;
}

int32_t dbhip_memcpy_d2d(void* dst_dev, const void* src_dev, size_t bytes, void* stream) {
  if (bytes == 0) return DBHIP_OK;
  DBHIP_REQUIRE(dst_dev && src_dev, "dbhip_memcpy_d2d: NULL pointer");
  DBHIP_CHECK(hipMemcpyAsync(dst_dev, src_dev, bytes, hipMemcpyDeviceToDevice, dbhip::resolve_stream(stream)));
  return DBHIP_OK;
}
int32_t dbhip_memset(void* dst, int32_t byte, size_t bytes, void* stream) {
  if (bytes == 0) return DBHIP_OK;
  DBHIP_CHECK(hipMemsetAsync(dst, byte, bytes, resolve_stream(stream)));
  return DBHIP_OK;
}

int32_t dbhip_stream_create(void** out) {
  DBHIP_REQUIRE(out, "dbhip_stream_create: out is NULL");
  hipStream_t s;
  DBHIP_CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  *out = (void*)s;
  return DBHIP_OK;
}

int32_t dbhip_scratch_stats(uint64_t* out2_host) {
  DBHIP_REQUIRE(out2_host, "dbhip_scratch_stats: NULL argument");
  std::lock_guard<std::mutex> lk(g_scratch_mu);
  out2_host[0] = out2_host[1] = 0;
  if (!g_scratch) return DBHIP_OK;
  for (auto& kv : *g_scratch) {
    out2_host[0] += 1;
    for (const Scratch& sl : kv.second.slot) out2_host[1] += sl.cap;
  }
  return DBHIP_OK;
}

int32_t dbhip_stream_release_scratch(void* stream) {
  hipStream_t s = resolve_stream(stream);
  DBHIP_CHECK(hipStreamSynchronize(s));
  release_stream_scratch(s);
  return DBHIP_OK;
}

int32_t dbhip_stream_destroy(void* stream) {
  if (!stream) return DBHIP_OK;
  DBHIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
  release_stream_scratch((hipStream_t)stream);   // every thread's scratch of this stream goes with it
  set_cancel((hipStream_t)stream, false);          // (a recycled handle value must not inherit a mark)
  DBHIP_CHECK(hipStreamDestroy((hipStream_t)stream));
  return DBHIP_OK;
}

int32_t dbhip_stream_cancel(void* stream) { set_cancel(resolve_stream(stream), true); return DBHIP_OK; }
int32_t dbhip_stream_cancel_clear(void* stream) { set_cancel(resolve_stream(stream), false); return DBHIP_OK; }

int32_t dbhip_stream_sync(void* stream) {
  DBHIP_CHECK(hipStreamSynchronize(resolve_stream(stream)));
  return DBHIP_OK;
}

int32_t dbhip_event_create(void** out) {
  DBHIP_REQUIRE(out, "dbhip_event_create: out is NULL");
  hipEvent_t e;
  DBHIP_CHECK(hipEventCreate(&e));
  *out = (void*)e;
  return DBHIP_OK;
}

int32_t dbhip_event_record(void* event, void* stream) {
  DBHIP_CHECK(hipEventRecord((hipEvent_t)event, resolve_stream(stream)));
  return DBHIP_OK;
}

int32_t dbhip_event_elapsed_ms(void* start, void* stop, float* out_ms) {
  DBHIP_CHECK(hipEventSynchronize((hipEvent_t)stop));
  DBHIP_CHECK(hipEventElapsedTime(out_ms, (hipEvent_t)start, (hipEvent_t)stop));
  return DBHIP_OK;
}

int32_t dbhip_last_kernel_ms(float* out_ms) {
  DBHIP_REQUIRE(out_ms, "dbhip_last_kernel_ms: NULL out");
  DBHIP_REQUIRE(g_timed, "dbhip_last_kernel_ms: no timed kernel was launched on this thread");
  DBHIP_CHECK(hipEventSynchronize(g_t1));
  DBHIP_CHECK(hipEventElapsedTime(out_ms, g_t0, g_t1));
  return DBHIP_OK;
}

int32_t dbhip_event_destroy(void* event) {
  if (!event) return DBHIP_OK;
  DBHIP_CHECK(hipEventDestroy((hipEvent_t)event));
  return DBHIP_OK;
}

}  // extern "C"

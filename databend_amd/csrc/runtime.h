// runtime.h — host-side plumbing shared by the C-ABI translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "../../include/dbhip.h"

namespace dbhip {

void set_error(const char* fmt, ...);
hipStream_t resolve_stream(void* stream);  // NULL -> library stream
int32_t hip_fail(hipError_t e, const char* what);

// Scratch arena bound to the library (grown on demand, reused across calls on
// the same (thread, stream); callers must not hold it across dbhip calls). Bounded: at most 8 streams' worth per thread (LRU),
// released with the stream (dbhip_stream_destroy), on request (dbhip_stream_release_scratch) and when the thread exits.
void* scratch(size_t bytes, int slot, hipStream_t stream);
// Frees every thread's scratch of `stream` (the caller has drained it): dbhip_stream_destroy / dbhip_stream_release_scratch.
void release_stream_scratch(hipStream_t stream);

// dbhip_stream_cancel: true once the stream has been marked (one relaxed atomic load when nothing is cancelled anywhere)
bool cancel_requested(hipStream_t stream);

// Pinned host words for small device -> host read-backs that are queued asynchronously (64 u64 per slot, 8 slots per
// thread). Two async copies into PAGEABLE memory in flight at once — a kernel's control block, then the queued merge's —
// made the runtime lock / unlock the same stack page twice and the second copy faulted ("write access to a read-only
// page") once the first had completed and unlocked it; pinned memory needs no lock.
uint64_t* pinned_words(int slot);

// HIP-event bracket around the dominant kernel of a call (dbhip_last_kernel_ms).
void kernel_timer_start(hipStream_t s);
void kernel_timer_stop(hipStream_t s);

// Tuning knobs of the development sweeps (grid sizes, kernel variants, thresholds; DESIGN.md names them where it quotes a sweep): compiled
// in only with -DDBHIP_EXPERIMENTS (`make EXPERIMENTS=1`). The shipped library does not read them — its behaviour is a function of its
// arguments and of the documented configuration variables (DBHIP_TRACE, DBHIP_JIT_CACHE_DIR, DBHIP_JIT_ARCH, DBHIP_FAGG_JIT,
// DBHIP_COMM_TIMEOUT_S, DBHIP_CACHE_BYTES) only, and no variable of either kind skips work (tests/test_abi.py checks the binary).
#ifdef DBHIP_EXPERIMENTS
inline const char* exp_env(const char* name) { return getenv(name); }
#else
inline const char* exp_env(const char*) { return nullptr; }
#endif

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Grid for HBM-bound grid-stride kernels: enough workgroups to fill 256 CUs
// with 8 blocks each (guide §6 G11), never more than the work needs.
// `cap`: workgroups of a grid-stride streaming kernel. 2048 (8 per CU) suits kernels with 16 B per lane in flight;
// kernels that keep 32+ B per operand per lane in flight stream FASTER from fewer workgroups (r01x sweep: `plus`
// 0.69 -> 0.71 at 1024, fused sum(a+b*c) 0.67 -> 0.79 at 512), always whole multiples of the 256 CUs.
// env DBHIP_GRID_CAP overrides every kernel (experiments).
int grid_cap_override();
inline int grid_for(int64_t n_items_per_thread_units, int block, int cap = 2048) {
  int64_t need = ceil_div(n_items_per_thread_units, block);
  if (need < 1) need = 1;
  const int o = grid_cap_override();
  if (o > 0) cap = o;
  if (need > cap) need = cap;
  return (int)need;
}

inline int type_size(int32_t t) {
  switch (t) {
    case DBHIP_T_I8: case DBHIP_T_U8: return 1;
    case DBHIP_T_I16: case DBHIP_T_U16: return 2;
    case DBHIP_T_I32: case DBHIP_T_U32: case DBHIP_T_F32: case DBHIP_T_DATE: return 4;
    case DBHIP_T_I64: case DBHIP_T_U64: case DBHIP_T_F64: case DBHIP_T_TIMESTAMP:
    case DBHIP_T_DEC64: return 8;
    case DBHIP_T_DEC128: case DBHIP_T_STRING: return 16;
    case DBHIP_T_DEC256: return 32;
    default: return 0;
  }
}

}  // namespace dbhip

// k_scatter.hip, for k_comm.hip's exchange: the rows-per-destination histogram left on the device (counts_dev[scatter_size + 1], the last word
// counts indices that are out of range), and dbhip_scatter_columns given those counts already on the host (no histogram, no drain for them)
int32_t dbhip_scatter_count_internal(const uint32_t* index, int64_t n, uint32_t scatter_size, uint64_t* counts_dev, hipStream_t s);
int32_t dbhip_scatter_columns_counted_internal(const dbhip_col* cols, int32_t ncols, const uint32_t* index, int64_t n, uint32_t scatter_size,
                                               void* const* out_data_host, uint8_t* const* out_validity_host, int64_t* out_row_starts_host,
                                               const uint64_t* known_counts_host, void* stream);

#define DBHIP_CHECK(expr)                                   \
  do {                                                      \
    hipError_t _e = (expr);                                 \
    if (_e != hipSuccess) return dbhip::hip_fail(_e, #expr); \
  } while (0)

#define DBHIP_REQUIRE(cond, msg)      \
  do {                                \
    if (!(cond)) {                    \
      dbhip::set_error("%s", msg);    \
      return DBHIP_ERR_INVALID;       \
    }                                 \
  } while (0)

#define DBHIP_LAUNCH_CHECK() DBHIP_CHECK(hipGetLastError())

// between two launches of a multi-launch operator
#define DBHIP_POLL_CANCEL(stream, what)                                              \
  do {                                                                               \
    if (dbhip::cancel_requested(stream)) {                                           \
      dbhip::set_error("%s: cancelled (dbhip_stream_cancel)", what);                 \
      return DBHIP_ERR_CANCELLED;                                                    \
    }                                                                                \
  } while (0)

// k_comm.hip — the communicator behind the boundary (north_star: "the host calls the kernels ... with RCCL
// all-reduce / all-to-all over xGMI for the final merge"). One process per GPU; the binding creates ONE dbhip_comm per rank
// (rank 0 draws the unique id, the host's own control plane ships its 128 bytes to the other ranks) and the exchange entry points
// run flush -> collective -> merge on ONE stream without a host round trip between them:
//   dbhip_groupby_exchange_allgather   low-cardinality final merge: every rank's table as one fixed-size block, ncclAllGather,
//                                      merge of the other ranks' blocks (aggregate_exchange_injector.rs:57-147 broadcast shape)
//   dbhip_groupby_exchange_alltoall    hash-partitioned final merge: rows scattered by hash % world into `world` fixed-size
//                                      blocks (scan_hash_partition_transfer, payload.rs:548-589), all-to-all, the table rebuilt
//                                      from the received blocks — rank r then owns the groups with hash % world == r
//   dbhip_comm_allgather / _alltoall / _allreduce_sum_u64   the plain collectives (ANN shard top-k all-gather, result checks)
// RCCL is loaded with dlopen at dbhip_comm_create: libdbhip.so carries no link-time dependency on it, a single-GPU binding never
// loads it. xGMI is point to point (7 links per GPU): the all-to-all is ncclSend / ncclRecv pairs inside one group so that all
// links carry traffic at once; blocks are fixed-size so the collective needs no size negotiation.
#include "runtime.h"
#include "dev_scan.h"

#include <dlfcn.h>
#include <string.h>

#include <chrono>
#include <condition_variable>
#include <string>
#include <map>
#include <mutex>
#include <new>
#include <vector>

using namespace dbhip;

namespace {

// the slice of rccl.h this file needs (ABI of ROCm 7: opaque communicator, 128-byte unique id, int enums)
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
enum { NCCL_SUCCESS = 0 };
enum { NCCL_INT8 = 0, NCCL_UINT8 = 1, NCCL_INT32 = 2, NCCL_UINT32 = 3, NCCL_INT64 = 4, NCCL_UINT64 = 5 };
enum { NCCL_SUM = 0 };

struct Rccl {
  void* lib = nullptr;
  int (*GetUniqueId)(ncclUniqueId*) = nullptr;
  int (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  int (*CommDestroy)(ncclComm_t) = nullptr;
  int (*CommAbort)(ncclComm_t) = nullptr;          // (optional)
  int (*AllGather)(const void*, void*, size_t, int, ncclComm_t, hipStream_t) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  int (*Send)(const void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  int (*Recv)(void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
};
Rccl g_rccl;

int32_t load_rccl() {
  if (g_rccl.lib) return DBHIP_OK;
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  void* h = nullptr;
  // an RCCL the host process already holds (a PyTorch host ships its own librccl.so) is the one to use: two copies in one process
  // tear each other's state down at exit (seen as "double free or corruption" when the tests of this file and torch-using ones share
  // a process)
  for (const char* n : names) if ((h = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;
  if (!h)
    for (const char* n : names) if ((h = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
  if (!h) { set_error("dbhip_comm: librccl.so not found (%s)", dlerror()); return DBHIP_ERR_UNSUPPORTED; }
#define SYM(field, name)                                                                                   \
  do {                                                                                                     \
    *(void**)(&g_rccl.field) = dlsym(h, name);                                                             \
    if (!g_rccl.field) { set_error("dbhip_comm: librccl.so lacks %s", name); dlclose(h); return DBHIP_ERR_UNSUPPORTED; } \
  } while (0)
  SYM(GetUniqueId, "ncclGetUniqueId"); SYM(CommInitRank, "ncclCommInitRank"); SYM(CommDestroy, "ncclCommDestroy");
  SYM(AllGather, "ncclAllGather"); SYM(AllReduce, "ncclAllReduce"); SYM(Send, "ncclSend"); SYM(Recv, "ncclRecv");
  SYM(GroupStart, "ncclGroupStart"); SYM(GroupEnd, "ncclGroupEnd"); SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
  g_rccl.CommAbort = (int (*)(ncclComm_t))dlsym(h, "ncclCommAbort");
  g_rccl.lib = h;
  return DBHIP_OK;
}

int32_t nccl_fail(int r, const char* what) {
  set_error("dbhip_comm: %s failed: %s", what, g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?");
  return DBHIP_ERR_HIP;
}
#define NCCL_CHECK(expr)                         \
  do {                                           \
    const int _r = (expr);                       \
    if (_r != NCCL_SUCCESS) return nccl_fail(_r, #expr); \
  } while (0)

}  // namespace

// An in-process world (dbhip_comm_create_loopback): `world` communicators of ONE process — one per host thread, all on the same GPU —
// whose collectives are device-to-device copies made by the last rank to arrive at a rendezvous. It exists so that the multi-rank
// protocols of this file (block exchange, shard top-k merge, partial-state exchange) run with world > 1 on the one-GPU boxes the
// tests get; it is not a transport anybody should ship data over.
struct LoopOp {
  int kind = 0;                       // 1 all-gather (bytes per rank), 2 grouped pieces
  const uint8_t* send = nullptr; uint8_t* recv = nullptr; size_t bytes = 0;
  const void* pieces = nullptr;       // std::vector<XPiece>*
};
struct LoopGroup {
  std::mutex mu;
  std::condition_variable cv;
  int world = 0, arrived = 0, refs = 0;
  uint64_t generation = 0;
  int32_t status = 0;
  std::string status_msg;        // the message behind `status` (set_error is per thread: the ranks that waited get it from here)
  bool failed = false;           // a rank gave up (dbhip_comm_abort, a failed exchange, a rendezvous that timed out): every waiter and
  std::string why;               // every later collective of the group returns an error instead of waiting for a rank that will not come
  std::vector<LoopOp> ops;
};

struct dbhip_comm {
  ncclComm_t comm = nullptr;
  LoopGroup* loop = nullptr;
  int rank = 0, world = 1;
  bool aborted = false;      // dbhip_comm_abort ran on an RCCL communicator: every later collective returns an error (a NULL `comm` alone
                             // means "a local world of one", whose collectives are copies)
  void* send = nullptr;      // exchange staging (blocks), grown on demand
  void* recv = nullptr;
  size_t cap = 0;
};

namespace {

// `comm == nullptr` is the single-rank copy path ONLY for a world of one; an aborted communicator answers every collective with an error
int32_t comm_usable(const dbhip_comm* c, const char* what) {
  if (c->aborted) { set_error("%s: the communicator was aborted (dbhip_comm_abort)", what); return DBHIP_ERR_INVALID; }
  if (!c->comm && !c->loop && c->world != 1) { set_error("%s: the communicator of a world of %d has no RCCL handle", what, c->world); return DBHIP_ERR_INVALID; }
  return DBHIP_OK;
}

int32_t ensure_staging(dbhip_comm* c, size_t bytes) {
  if (c->cap >= bytes) return DBHIP_OK;
  if (c->send) (void)dbhip_free(c->send);
  if (c->recv) (void)dbhip_free(c->recv);
  c->send = c->recv = nullptr; c->cap = 0;
  int32_t rc = dbhip_alloc(bytes, &c->send);
  if (rc == DBHIP_OK) rc = dbhip_alloc(bytes, &c->recv);
  if (rc) return rc;
  c->cap = bytes;
  return DBHIP_OK;
}

// equal-split all-to-all of `bytes_per_peer` bytes per rank pair: one group of send / recv pairs (every xGMI link busy at once)
struct XPiece;
int32_t alltoall_bytes_loop(dbhip_comm* c, const void* send, void* recv, size_t bytes_per_peer, hipStream_t s);
int32_t alltoall_bytes(dbhip_comm* c, const void* send, void* recv, size_t bytes_per_peer, hipStream_t s) {
  if (c->loop) return alltoall_bytes_loop(c, send, recv, bytes_per_peer, s);
  if (const int32_t u = comm_usable(c, "all-to-all")) return u;
  if (!c->comm) {   // a local world of one: the exchange is a copy
    if (send != recv) DBHIP_CHECK(hipMemcpyAsync(recv, send, bytes_per_peer, hipMemcpyDeviceToDevice, s));
    return DBHIP_OK;
  }
  NCCL_CHECK(g_rccl.GroupStart());
  // (a failing Send / Recv must not leave the thread's group open: every later collective of this thread would be queued into
  // the dangling group and never launched — the first error is kept, the group is always closed)
  int first = NCCL_SUCCESS;
  const char* what = "";
  for (int p = 0; p < c->world && first == NCCL_SUCCESS; ++p) {
    int r = g_rccl.Send((const uint8_t*)send + (size_t)p * bytes_per_peer, bytes_per_peer, NCCL_UINT8, p, c->comm, s);
    if (r != NCCL_SUCCESS) { first = r; what = "ncclSend"; break; }
    r = g_rccl.Recv((uint8_t*)recv + (size_t)p * bytes_per_peer, bytes_per_peer, NCCL_UINT8, p, c->comm, s);
    if (r != NCCL_SUCCESS) { first = r; what = "ncclRecv"; }
  }
  const int e = g_rccl.GroupEnd();
  if (first != NCCL_SUCCESS) return nccl_fail(first, what);
  if (e != NCCL_SUCCESS) return nccl_fail(e, "ncclGroupEnd");
  return DBHIP_OK;
}

// variable-size pieces between all ranks in ONE group: piece p of every entry goes to / comes from rank p (a local world of one: copies)
struct XPiece { const uint8_t* send; uint8_t* recv; const size_t* send_off; const size_t* send_bytes; const size_t* recv_off; const size_t* recv_bytes; };

int32_t loop_failed(LoopGroup* g) {
  set_error("dbhip_comm loopback: the group was aborted (%s)", g->why.c_str());
  return DBHIP_ERR_INVALID;
}
// marks the group failed and wakes every rank waiting in a rendezvous
void loop_abort(dbhip_comm* c, const char* why) {
  LoopGroup* g = c->loop;
  if (!g) return;
  std::lock_guard<std::mutex> lk(g->mu);
  if (!g->failed) {
    g->failed = true;
    char buf[400];
    snprintf(buf, sizeof(buf), "rank %d: %s", c->rank, why ? why : "");
    g->why = buf;
  }
  g->cv.notify_all();
}
int loop_timeout_seconds() {
  static const int t = [] { const char* e = getenv("DBHIP_COMM_TIMEOUT_S"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 600; }();
  return t;
}

// loopback rendezvous: every rank drains its stream and posts its operation; the last one to arrive makes all the copies. A rank
// that will not arrive must not leave the others waiting: an aborted group (loop_abort) and a wait past DBHIP_COMM_TIMEOUT_S
// (default 600 s) end the rendezvous with an error on every rank.
int32_t loop_collective(dbhip_comm* c, const LoopOp& mine, hipStream_t s) {
  LoopGroup* g = c->loop;
  DBHIP_CHECK(hipStreamSynchronize(s));
  std::unique_lock<std::mutex> lk(g->mu);
  if (g->failed) return loop_failed(g);
  g->ops[c->rank] = mine;
  const uint64_t gen = g->generation;
  if (++g->arrived == g->world) {
    int32_t st = DBHIP_OK;
    for (int a = 0; a < g->world && st == DBHIP_OK; ++a) {          // a: source rank
      for (int b = 0; b < g->world && st == DBHIP_OK; ++b) {        // b: destination rank
        const LoopOp &A = g->ops[a], &B = g->ops[b];
        if (A.kind != B.kind) { set_error("dbhip_comm loopback: ranks %d and %d are in different collectives", a, b); st = DBHIP_ERR_INVALID; break; }
        if (A.kind == 1) {
          if (A.bytes != B.bytes) { set_error("dbhip_comm loopback: all-gather sizes differ"); st = DBHIP_ERR_INVALID; break; }
          if (A.bytes && hipMemcpyAsync(B.recv + (size_t)a * A.bytes, A.send, A.bytes, hipMemcpyDeviceToDevice, s) != hipSuccess) st = DBHIP_ERR_HIP;
        } else {
          const std::vector<XPiece>& xa = *(const std::vector<XPiece>*)A.pieces;
          const std::vector<XPiece>& xb = *(const std::vector<XPiece>*)B.pieces;
          if (xa.size() != xb.size()) { set_error("dbhip_comm loopback: ranks post different numbers of pieces"); st = DBHIP_ERR_INVALID; break; }
          for (size_t e = 0; e < xa.size() && st == DBHIP_OK; ++e) {
            if (xa[e].send_bytes[b] != xb[e].recv_bytes[a]) {
              set_error("dbhip_comm loopback: rank %d sends %zu bytes to rank %d, which expects %zu", a, xa[e].send_bytes[b], b, xb[e].recv_bytes[a]);
              st = DBHIP_ERR_INVALID;
              break;
            }
            if (xa[e].send_bytes[b] &&
                hipMemcpyAsync(xb[e].recv + xb[e].recv_off[a], xa[e].send + xa[e].send_off[b], xa[e].send_bytes[b], hipMemcpyDeviceToDevice, s) != hipSuccess)
              st = DBHIP_ERR_HIP;
          }
        }
      }
    }
    if (st == DBHIP_OK && hipStreamSynchronize(s) != hipSuccess) st = DBHIP_ERR_HIP;
    if (st == DBHIP_ERR_HIP) set_error("dbhip_comm loopback: a device copy of the rendezvous failed");
    g->status = st;
    g->status_msg = st == DBHIP_OK ? "" : dbhip_last_error();
    g->arrived = 0;
    ++g->generation;
    g->cv.notify_all();
    return st;
  }
  const bool in_time = g->cv.wait_for(lk, std::chrono::seconds(loop_timeout_seconds()), [&] { return g->generation != gen || g->failed; });
  if (g->generation != gen) {   // the rendezvous completed (its status is every rank's status)
    if (g->status != DBHIP_OK) set_error("%s", g->status_msg.c_str());
    return g->status;
  }
  if (!in_time && !g->failed) {
    g->failed = true;
    char buf[200];
    snprintf(buf, sizeof(buf), "rank %d waited %d s for %d of %d ranks", c->rank, loop_timeout_seconds(), g->world - g->arrived, g->world);
    g->why = buf;
    g->cv.notify_all();
  }
  --g->arrived;   // (this rank's post is withdrawn: the group is dead, nobody will complete it)
  return loop_failed(g);
}

int32_t alltoallv_group(dbhip_comm* c, const std::vector<XPiece>& xs, hipStream_t s) {
  if (c->loop) { LoopOp op; op.kind = 2; op.pieces = &xs; return loop_collective(c, op, s); }
  if (const int32_t u = comm_usable(c, "all-to-all (variable)")) return u;
  if (!c->comm) {
    for (const XPiece& x : xs)
      if (x.send_bytes[0]) DBHIP_CHECK(hipMemcpyAsync(x.recv + x.recv_off[0], x.send + x.send_off[0], x.send_bytes[0], hipMemcpyDeviceToDevice, s));
    return DBHIP_OK;
  }
  NCCL_CHECK(g_rccl.GroupStart());
  int first = NCCL_SUCCESS;
  const char* what = "";
  for (const XPiece& x : xs) {
    for (int p = 0; p < c->world && first == NCCL_SUCCESS; ++p) {
      int r = NCCL_SUCCESS;
      if (x.send_bytes[p]) r = g_rccl.Send(x.send + x.send_off[p], x.send_bytes[p], NCCL_UINT8, p, c->comm, s);
      if (r != NCCL_SUCCESS) { first = r; what = "ncclSend"; break; }
      if (x.recv_bytes[p]) r = g_rccl.Recv(x.recv + x.recv_off[p], x.recv_bytes[p], NCCL_UINT8, p, c->comm, s);
      if (r != NCCL_SUCCESS) { first = r; what = "ncclRecv"; }
    }
  }
  const int e = g_rccl.GroupEnd();
  if (first != NCCL_SUCCESS) return nccl_fail(first, what);
  if (e != NCCL_SUCCESS) return nccl_fail(e, "ncclGroupEnd");
  return DBHIP_OK;
}

int32_t alltoall_bytes_loop(dbhip_comm* c, const void* send, void* recv, size_t bytes_per_peer, hipStream_t s) {
  std::vector<size_t> off(c->world), len(c->world, bytes_per_peer);
  for (int p = 0; p < c->world; ++p) off[p] = (size_t)p * bytes_per_peer;
  std::vector<XPiece> xs{XPiece{(const uint8_t*)send, (uint8_t*)recv, off.data(), len.data(), off.data(), len.data()}};
  return alltoallv_group(c, xs, s);
}

__global__ __launch_bounds__(256) void topk_globalise_kernel(const uint32_t* __restrict__ idx, int64_t n, uint64_t row_offset, uint32_t* __restrict__ out,
                                                             uint32_t* __restrict__ overflow) {
  bool bad = false;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const uint32_t v = idx[i];
    const uint64_t gid = (uint64_t)v + row_offset;
    bad |= v != 0xFFFFFFFFu && gid > 0xFFFFFFFEULL;   // (0xFFFFFFFF is "no neighbour": a real id must stay below it)
    out[i] = v == 0xFFFFFFFFu ? v : (uint32_t)gid;
  }
  if (__ballot(bad) && lane_id() == 0) atomicOr(overflow, 1u);
}
// [world][nq][k] (rank-major, what the all-gather produces) -> [nq][world * k] (what dbhip_vec_topk_merge takes)
template <typename T>
__global__ __launch_bounds__(256) void topk_regroup_kernel(const T* __restrict__ in, int world, int nq, int k, T* __restrict__ out) {
  const int64_t total = (int64_t)world * nq * k;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int j = (int)(i % k);
    const int64_t t = i / k;
    const int q = (int)(t % nq), r = (int)(t / nq);
    out[((int64_t)q * world + r) * k + j] = in[i];
  }
}

// ---- long strings in the exchange (round 5; flight_scatter_hash.rs:57-120 moves whole blocks, strings included) ----------------------
// views (16 bytes) -> bytes of the long form per row (0 for inline strings)
__global__ __launch_bounds__(256) void xs_long_len_kernel(const uint32_t* __restrict__ views, int64_t n, uint32_t* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const uint32_t len = views[4 * i];
    out[i] = len > 12 ? len : 0u;
  }
}
// the long strings of the scattered rows, copied back to back in row order into `packed` (off[i] = where row i's bytes go), and the view
// rewritten to {len, prefix, buffer 0, offset inside its DESTINATION's piece}. One wave per 64 rows: the lanes copy one row's bytes together.
__global__ __launch_bounds__(256) void xs_pack_kernel(uint32_t* __restrict__ views, int64_t n, const uint64_t* __restrict__ off,
                                                      const void* const* __restrict__ buffers, const int64_t* __restrict__ dest_start, int world,
                                                      uint8_t* __restrict__ packed, uint32_t* __restrict__ err) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t base = wave * 64; base < n; base += nwaves * 64) {
    const int64_t i = base + lane;
    uint32_t len = 0, bufi = 0, boff = 0;
    uint64_t o = 0;
    if (i < n) { len = views[4 * i]; bufi = views[4 * i + 2]; boff = views[4 * i + 3]; o = off[i]; }
    const bool lng = len > 12;
    // the destination whose piece row i belongs to: the last d with dest_start[d] <= i
    uint64_t piece0 = 0;
    if (lng) {
      int lo = 0, hi = world;
      while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (dest_start[mid] <= i) lo = mid; else hi = mid; }
      piece0 = off[dest_start[lo]];
    }
    uint64_t todo = __ballot(lng);
    while (todo) {
      const int r = __ffsll((long long)todo) - 1;
      todo &= todo - 1;
      const uint32_t rl = __shfl(len, r, 64);
      const uint8_t* src = (const uint8_t*)buffers[__shfl(bufi, r, 64)] + __shfl(boff, r, 64);
      uint8_t* dst = packed + __shfl(o, r, 64);
      for (uint32_t b = lane; b < rl; b += 64) dst[b] = src[b];
    }
    if (lng) {
      const uint64_t rel = o - piece0;
      if (rel >> 32) atomicOr(err, 1u);   // a piece of 4 GiB or more: a view's offset is 32 bits
      views[4 * i + 2] = 0;
      views[4 * i + 3] = (uint32_t)rel;
    }
  }
}
// out[p] = off[idx[p]] for a handful of positions (the byte offset at which every destination's piece starts)
__global__ void xs_pick_kernel(const uint64_t* __restrict__ off, const int64_t* __restrict__ idx, int n, uint64_t* __restrict__ out) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p < n) out[p] = off[idx[p]];
}
// received views of source p (rows [row_start[p], row_start[p + 1])): offsets were relative to p's piece -> relative to the packed buffer
__global__ __launch_bounds__(256) void xs_rebase_kernel(uint32_t* __restrict__ views, int64_t n, const int64_t* __restrict__ row_start,
                                                        const uint64_t* __restrict__ byte_start, int world, uint32_t* __restrict__ err) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    if (views[4 * i] <= 12) continue;
    int lo = 0, hi = world;
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (row_start[mid] <= i) lo = mid; else hi = mid; }
    const uint64_t o = (uint64_t)views[4 * i + 3] + byte_start[lo];
    if (o >> 32) atomicOr(err, 1u);
    views[4 * i + 2] = 0;
    views[4 * i + 3] = (uint32_t)o;
  }
}

}  // namespace

// One exchange of a block between all ranks (see dbhip.h): the scattered columns and the counts between begin and finish
struct dbhip_exchange {
  dbhip_comm* c = nullptr;
  int64_t n = 0, recv_total = 0;
  std::vector<dbhip_col> cols;
  std::vector<void*> sdata;
  std::vector<uint8_t*> svalid;
  std::vector<int64_t> send_start, recv_start;   // [world + 1] rows
  void* counts_dev = nullptr;
  // String columns with data buffers: the long bytes of every destination's rows, packed (begin), and how many arrive from whom
  std::vector<int> str_cols;                       // columns that carry long strings
  std::vector<uint8_t*> spacked;                   // per such column: the packed send bytes
  std::vector<std::vector<uint64_t>> sbyte_start;  // per such column: [world + 1] byte offsets of the destinations' pieces in spacked
  std::vector<std::vector<uint64_t>> rbyte_start;  // per such column: [world + 1] byte offsets of the sources' pieces in the output buffer
};

extern "C" {

int32_t dbhip_exchange_destroy(dbhip_exchange* x) {
  if (!x) return DBHIP_OK;
  for (void* p : x->sdata) if (p) (void)dbhip_free(p);
  for (uint8_t* p : x->svalid) if (p) (void)dbhip_free(p);
  for (uint8_t* p : x->spacked) if (p) (void)dbhip_free(p);
  if (x->counts_dev) (void)dbhip_free(x->counts_dev);
  delete x;
  return DBHIP_OK;
}

static int32_t exchange_begin_impl(dbhip_comm* c, const dbhip_col* cols, int32_t ncols, const uint32_t* dest_index, int64_t n, int64_t* out_recv_rows_host,
                                   dbhip_exchange** out_host, void* stream);
// A rank whose exchange fails — bad arguments, an allocation, the scatter — would leave the other ranks of a loopback group in the
// rendezvous of the counts: it aborts the group instead, and they return DBHIP_ERR_INVALID with this rank's message.
int32_t dbhip_exchange_begin(dbhip_comm* c, const dbhip_col* cols, int32_t ncols, const uint32_t* dest_index, int64_t n, int64_t* out_recv_rows_host,
                             dbhip_exchange** out_host, void* stream) {
  // (ANY failure aborts the group, argument errors included: the peers enter the same exchange on their own threads and may already
  // be waiting in the rendezvous of the counts for a rank that has just decided not to come — tests/test_gpu_comm.py holds that case.
  // The price: a group is dead after one bad call; it is per query, like the reference's exchange channels.)
  const int32_t rc = exchange_begin_impl(c, cols, ncols, dest_index, n, out_recv_rows_host, out_host, stream);
  if (rc != DBHIP_OK && c && c->loop) {
    const std::string why = dbhip_last_error();   // (loop_abort must not disturb this rank's own message)
    loop_abort(c, why.c_str());
  }
  return rc;
}
static int32_t exchange_begin_impl(dbhip_comm* c, const dbhip_col* cols, int32_t ncols, const uint32_t* dest_index, int64_t n, int64_t* out_recv_rows_host,
                                   dbhip_exchange** out_host, void* stream) {
  DBHIP_REQUIRE(c && out_recv_rows_host && out_host && ncols >= 0 && n >= 0 && (ncols == 0 || cols), "dbhip_exchange_begin: bad argument");
  for (int k = 0; k < ncols; ++k)
    DBHIP_REQUIRE(!(cols[k].type == DBHIP_T_STRING && cols[k].n_buffers > 0 && !cols[k].buffers), "dbhip_exchange_begin: a String column with n_buffers > 0 needs its buffers");
  dbhip_exchange* x = new (std::nothrow) dbhip_exchange();
  if (!x) return DBHIP_ERR_HIP;
  x->c = c; x->n = n;
  x->cols.assign(cols, cols + ncols);
  x->sdata.assign(ncols, nullptr);
  x->svalid.assign(ncols, nullptr);
  const int W = c->world;
  x->send_start.assign(W + 1, 0);
  x->recv_start.assign(W + 1, 0);
  hipStream_t s = resolve_stream(stream);
  int32_t rc = DBHIP_OK;
  const size_t bm_bytes = (size_t)8 * ((size_t)(n >> 6) + W + 1);
  for (int k = 0; k < ncols && rc == DBHIP_OK; ++k) {
    const int t = cols[k].type;
    const size_t bytes = t == DBHIP_T_BOOL ? bm_bytes : (size_t)(n > 0 ? n : 1) * type_size(t) + 64;
    rc = dbhip_alloc(bytes, &x->sdata[k]);
    if (rc == DBHIP_OK && cols[k].validity) rc = dbhip_alloc(bm_bytes, (void**)&x->svalid[k]);
  }
  bool any_string = false;
  for (int k = 0; k < ncols; ++k) any_string |= cols[k].type == DBHIP_T_STRING;
  if (rc == DBHIP_OK) rc = dbhip_alloc((size_t)(2 * W + 2) * 8, &x->counts_dev);
  // Fixed-width columns only: the rows-per-destination histogram stays on the device, the all-to-all of the counts is
  // posted from there BEFORE any column is scattered, and the host reads what it sends and what it will receive in ONE copy — the peers
  // have this rank's counts while it is still scattering. (Rounds 3-5: histogram -> host -> scatter -> host -> device -> all-to-all -> host.)
  // String columns add their byte counts to the message, known only after the scatter: they keep that order.
  bool counts_done = false;
  if (rc == DBHIP_OK && !any_string) {
    uint64_t* d = (uint64_t*)x->counts_dev;          // [W] sent rows | [1] out-of-range indices | [W] received rows
    std::vector<uint64_t> hc((size_t)2 * W + 1);
    rc = dbhip_scatter_count_internal(dest_index, n, (uint32_t)W, d, s);
    if (rc == DBHIP_OK) rc = alltoall_bytes(c, d, d + W + 1, 8, s);
    if (rc == DBHIP_OK) rc = dbhip_memcpy_d2h(hc.data(), d, hc.size() * 8, stream);
    if (rc == DBHIP_OK && hc[W]) {
      set_error("dbhip_exchange_begin: %llu destination indices are not below the world size %d", (unsigned long long)hc[W], W);
      rc = DBHIP_ERR_INVALID;
    }
    if (rc == DBHIP_OK) {
      for (int p = 0; p < W; ++p) x->recv_start[p + 1] = x->recv_start[p] + (int64_t)hc[(size_t)W + 1 + p];
      x->recv_total = x->recv_start[W];
      rc = dbhip_scatter_columns_counted_internal(cols, ncols, dest_index, n, (uint32_t)W, x->sdata.data(), x->svalid.data(), x->send_start.data(), hc.data(), stream);
      counts_done = true;
    }
  } else if (rc == DBHIP_OK)
    rc = dbhip_scatter_columns(cols, ncols, dest_index, n, (uint32_t)W, x->sdata.data(), x->svalid.data(), x->send_start.data(), stream);
  // String columns with data buffers: every destination's long strings packed back to back, the scattered views re-based onto their piece
  for (int k = 0; k < ncols && rc == DBHIP_OK; ++k) {
    // EVERY String column takes part in the byte-count exchange (the schema decides, so that all ranks agree on the size of the counts
    // message); a shard without rows, or whose values are all inline (no data buffers), simply has no bytes to send
    if (cols[k].type != DBHIP_T_STRING) continue;
    if (cols[k].n_buffers <= 0 || n == 0) {
      x->str_cols.push_back(k);
      x->spacked.push_back(nullptr);
      x->sbyte_start.emplace_back(W + 1, 0);
      x->rbyte_start.emplace_back(W + 1, 0);
      continue;
    }
    uint32_t* lens = nullptr;
    uint64_t *offs = nullptr, *blk = nullptr;
    int64_t* dstart = nullptr;
    uint32_t* err = nullptr;
    uint8_t* packed = nullptr;
    std::vector<uint64_t> bstart(W + 1, 0);
    rc = dbhip_alloc((size_t)(n + 1) * 4, (void**)&lens);
    if (rc == DBHIP_OK) rc = dbhip_alloc((size_t)(n + 1) * 8, (void**)&offs);
    if (rc == DBHIP_OK) rc = dbhip_alloc((size_t)(ceil_div(n + 1, SCAN_TILE) + 2) * 8, (void**)&blk);
    if (rc == DBHIP_OK) rc = dbhip_alloc((size_t)(W + 1) * 16 + 8, (void**)&dstart);
    uint64_t* picked = nullptr;
    if (rc == DBHIP_OK) {
      picked = (uint64_t*)(dstart + W + 1);
      err = (uint32_t*)(picked + W + 1);
      hipLaunchKernelGGL(xs_long_len_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, (const uint32_t*)x->sdata[k], n, lens);
      if (hipMemsetAsync(lens + n, 0, 4, s) != hipSuccess || hipMemsetAsync(err, 0, 8, s) != hipSuccess) rc = DBHIP_ERR_HIP;
    }
    if (rc == DBHIP_OK) rc = dbscan::exclusive_scan_u32(lens, n + 1, blk, offs, s);   // offs[n] = all long bytes
    if (rc == DBHIP_OK) rc = dbhip_memcpy_h2d(dstart, x->send_start.data(), (size_t)(W + 1) * 8, stream);
    uint64_t total = 0;
    if (rc == DBHIP_OK) rc = dbhip_memcpy_d2h(&total, offs + n, 8, stream);
    if (rc == DBHIP_OK) rc = dbhip_alloc((size_t)total + 64, (void**)&packed);
    if (rc == DBHIP_OK) {
      // where every destination's piece starts: the scan at its first row
      hipLaunchKernelGGL(xs_pick_kernel, dim3(ceil_div(W + 1, 64)), dim3(64), 0, s, offs, dstart, W + 1, picked);
      hipLaunchKernelGGL(xs_pack_kernel, dim3(grid_for(ceil_div(n, 64) * 64, 256)), dim3(256), 0, s, (uint32_t*)x->sdata[k], n, offs,
                         (const void* const*)cols[k].buffers, dstart, W, packed, err);
      uint32_t e = 0;
      if (rc == DBHIP_OK) rc = dbhip_memcpy_d2h(bstart.data(), picked, (size_t)(W + 1) * 8, stream);
      if (rc == DBHIP_OK) rc = dbhip_memcpy_d2h(&e, err, 4, stream);
      if (rc == DBHIP_OK && e) { set_error("dbhip_exchange_begin: column %d sends 4 GiB or more of string bytes to one rank (a view's offset is 32 bits)", k); rc = DBHIP_ERR_CAPACITY; }
    }
    for (void* p : {(void*)lens, (void*)offs, (void*)blk, (void*)dstart}) if (p) (void)dbhip_free(p);
    x->str_cols.push_back(k);
    x->spacked.push_back(packed);
    x->sbyte_start.push_back(bstart);
    x->rbyte_start.emplace_back(W + 1, 0);
  }
  if (rc == DBHIP_OK && !counts_done) {
    // what every rank sends to every other — rows, then the long-string bytes of every String column — in ONE small all-to-all, read back
    // once (the receiver sizes its buffers from it)
    const size_t per = 1 + x->str_cols.size();
    std::vector<uint64_t> sc((size_t)W * per), rcv((size_t)W * per);
    for (int p = 0; p < W; ++p) {
      sc[(size_t)p * per] = (uint64_t)(x->send_start[p + 1] - x->send_start[p]);
      for (size_t j = 0; j < x->str_cols.size(); ++j) sc[(size_t)p * per + 1 + j] = x->sbyte_start[j][p + 1] - x->sbyte_start[j][p];
    }
    (void)dbhip_free(x->counts_dev);
    x->counts_dev = nullptr;
    rc = dbhip_alloc((size_t)W * per * 16, &x->counts_dev);
    uint64_t* d = (uint64_t*)x->counts_dev;
    if (rc == DBHIP_OK) rc = dbhip_memcpy_h2d(d, sc.data(), (size_t)W * per * 8, stream);
    if (rc == DBHIP_OK) rc = alltoall_bytes(c, d, d + (size_t)W * per, 8 * per, s);
    if (rc == DBHIP_OK) rc = dbhip_memcpy_d2h(rcv.data(), d + (size_t)W * per, (size_t)W * per * 8, stream);
    for (int p = 0; p < W; ++p) {
      x->recv_start[p + 1] = x->recv_start[p] + (int64_t)rcv[(size_t)p * per];
      for (size_t j = 0; j < x->str_cols.size(); ++j) x->rbyte_start[j][p + 1] = x->rbyte_start[j][p] + rcv[(size_t)p * per + 1 + j];
    }
    x->recv_total = x->recv_start[W];
  }
  if (rc != DBHIP_OK) { (void)dbhip_exchange_destroy(x); return rc; }
  *out_recv_rows_host = x->recv_total;
  *out_host = x;
  return DBHIP_OK;
}

// The two plans that sit on the exchange, as single calls (SURVEY §8e "hash join: shuffle" and "sort"): where each row goes is decided
// by the reference's own rules — dbhip_scatter_indices (flight_scatter_hash.rs: siphash64 of the keys % nodes) / dbhip_sort_bound_partition
// (sort_spill.rs BoundBlockStream + sort_exchange_injector.rs: partition i -> node i % n) — then dbhip_exchange_begin.
int32_t dbhip_shuffle_exchange_begin(dbhip_comm* c, const dbhip_col* keys, int32_t nkeys, const dbhip_col* cols, int32_t ncols, int64_t n,
                                     int64_t* out_recv_rows_host, dbhip_exchange** out_host, void* stream) {
  DBHIP_REQUIRE(c && keys && nkeys >= 1 && n >= 0, "dbhip_shuffle_exchange_begin: bad argument");
  uint32_t* index = nullptr;
  uint64_t* counts = nullptr;
  int32_t rc = dbhip_alloc((size_t)(n > 0 ? n : 1) * 4, (void**)&index);
  if (rc == DBHIP_OK) rc = dbhip_alloc((size_t)c->world * 8, (void**)&counts);
  if (rc == DBHIP_OK) rc = dbhip_scatter_indices(keys, nkeys, n, (uint32_t)c->world, 0, index, counts, stream);
  if (rc == DBHIP_OK) rc = dbhip_exchange_begin(c, cols, ncols, index, n, out_recv_rows_host, out_host, stream);
  if (index) (void)dbhip_free(index);      // (dbhip_free drains the device: the scatter that read `index` has finished)
  if (counts) (void)dbhip_free(counts);
  return rc;
}

namespace {
__global__ __launch_bounds__(256) void part_to_rank_kernel(uint32_t* part, int64_t n, uint32_t world) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) part[i] %= world;
}
}  // namespace

int32_t dbhip_sort_exchange_begin(dbhip_comm* c, const dbhip_col* keys, const dbhip_col* bounds, const uint8_t* desc_host,
                                  const uint8_t* nulls_first_host, int32_t nkeys, int64_t nbounds, const dbhip_col* cols, int32_t ncols,
                                  int64_t n, int64_t* out_recv_rows_host, dbhip_exchange** out_host, void* stream) {
  DBHIP_REQUIRE(c && keys && nkeys >= 1 && n >= 0 && nbounds >= 0 && (nbounds == 0 || bounds), "dbhip_sort_exchange_begin: bad argument");
  uint32_t* part = nullptr;
  uint64_t* counts = nullptr;
  int32_t rc = dbhip_alloc((size_t)(n > 0 ? n : 1) * 4, (void**)&part);
  if (rc == DBHIP_OK) rc = dbhip_alloc((size_t)(nbounds + 1) * 8, (void**)&counts);
  if (rc == DBHIP_OK) rc = dbhip_sort_bound_partition(keys, bounds, desc_host, nulls_first_host, nkeys, n, nbounds, part, counts, stream);
  if (rc == DBHIP_OK && n > 0 && nbounds + 1 > c->world)   // more partitions than nodes: SortBoundScatter sends partition i to node i % n
    hipLaunchKernelGGL(part_to_rank_kernel, dim3(grid_for(n, 256)), dim3(256), 0, resolve_stream(stream), part, n, (uint32_t)c->world);
  if (rc == DBHIP_OK) rc = dbhip_exchange_begin(c, cols, ncols, part, n, out_recv_rows_host, out_host, stream);
  if (part) (void)dbhip_free(part);
  if (counts) (void)dbhip_free(counts);
  return rc;
}

int32_t dbhip_exchange_string_bytes(dbhip_exchange* x, int64_t* out_bytes_host) {
  DBHIP_REQUIRE(x && (out_bytes_host || x->cols.empty()), "dbhip_exchange_string_bytes: NULL argument");
  for (size_t k = 0; k < x->cols.size(); ++k) out_bytes_host[k] = 0;
  for (size_t j = 0; j < x->str_cols.size(); ++j) out_bytes_host[x->str_cols[j]] = (int64_t)x->rbyte_start[j][x->c->world];
  return DBHIP_OK;
}

int32_t dbhip_exchange_finish(dbhip_exchange* x, void* const* out_data_host, uint8_t* const* out_validity_host, int64_t* out_src_starts_host,
                              void* stream) {
  return dbhip_exchange_finish_strings(x, out_data_host, out_validity_host, nullptr, out_src_starts_host, stream);
}

int32_t dbhip_exchange_finish_strings(dbhip_exchange* x, void* const* out_data_host, uint8_t* const* out_validity_host,
                                      uint8_t* const* out_string_bytes_host, int64_t* out_src_starts_host, void* stream) {
  DBHIP_REQUIRE(x && (x->cols.empty() || (out_data_host && out_validity_host)), "dbhip_exchange_finish: bad argument");
  for (size_t j = 0; j < x->str_cols.size(); ++j)
    if (x->rbyte_start[j][x->c->world] > 0 && !(out_string_bytes_host && out_string_bytes_host[x->str_cols[j]])) {
      set_error("dbhip_exchange_finish: column %d receives %llu bytes of long strings: pass a buffer for them (dbhip_exchange_string_bytes, "
                "dbhip_exchange_finish_strings)", x->str_cols[j], (unsigned long long)x->rbyte_start[j][x->c->world]);
      return DBHIP_ERR_INVALID;
    }
  dbhip_comm* c = x->c;
  const int W = c->world, ncols = (int)x->cols.size();
  hipStream_t s = resolve_stream(stream);
  if (out_src_starts_host) for (int p = 0; p <= W; ++p) out_src_starts_host[p] = x->recv_start[p];
  // byte offsets / sizes of every piece. Value buffers: rows x element size. Bitmaps (Boolean values, validities): every destination's
  // piece is a stand-alone Bitmap on a 64-bit word of its own (dbhip_scatter_columns); the received pieces land word-aligned in a
  // temporary and are concatenated bit by bit (dbhip_concat_columns)
  std::vector<size_t> so_b(W), sb_b(W), ro_b(W), rb_b(W);
  size_t rwords = 0;
  for (int p = 0; p < W; ++p) {
    so_b[p] = (size_t)8 * ((size_t)(x->send_start[p] >> 6) + p);
    sb_b[p] = (size_t)8 * (size_t)((x->send_start[p + 1] - x->send_start[p] + 63) >> 6);
    ro_b[p] = rwords * 8;
    rb_b[p] = (size_t)8 * (size_t)((x->recv_start[p + 1] - x->recv_start[p] + 63) >> 6);
    rwords += rb_b[p] / 8;
  }
  std::vector<std::vector<size_t>> offs;   // keeps the per-column offset arrays alive until the group has been issued
  offs.reserve((size_t)ncols * 4);
  std::vector<XPiece> xs;
  std::vector<uint8_t*> tmp_bits;           // per bitmap piece set: the word-aligned receive image
  struct BitJob { uint8_t* tmp; uint8_t* out; };
  std::vector<BitJob> jobs;
  int32_t rc = DBHIP_OK;
  auto add_bits = [&](const uint8_t* send, uint8_t* out) -> int32_t {
    DBHIP_REQUIRE(out && ((uintptr_t)out & 7) == 0, "dbhip_exchange_finish: Bitmap outputs must be non-NULL and 8-byte aligned");
    void* t = nullptr;
    int32_t r = dbhip_alloc(rwords * 8 + 64, &t);
    if (r) return r;
    tmp_bits.push_back((uint8_t*)t);
    xs.push_back(XPiece{send, (uint8_t*)t, so_b.data(), sb_b.data(), ro_b.data(), rb_b.data()});
    jobs.push_back(BitJob{(uint8_t*)t, out});
    return DBHIP_OK;
  };
  for (int k = 0; k < ncols && rc == DBHIP_OK; ++k) {
    const int t = x->cols[k].type;
    if (t == DBHIP_T_BOOL) rc = add_bits((const uint8_t*)x->sdata[k], (uint8_t*)out_data_host[k]);
    else {
      if (x->recv_total > 0 && !out_data_host[k]) { set_error("dbhip_exchange_finish: NULL output for column %d", k); rc = DBHIP_ERR_INVALID; break; }
      const size_t es = (size_t)type_size(t);
      offs.emplace_back(W); offs.emplace_back(W); offs.emplace_back(W); offs.emplace_back(W);
      std::vector<size_t>&so = offs[offs.size() - 4], &sb = offs[offs.size() - 3], &ro = offs[offs.size() - 2], &rb = offs[offs.size() - 1];
      for (int p = 0; p < W; ++p) {
        so[p] = (size_t)x->send_start[p] * es; sb[p] = (size_t)(x->send_start[p + 1] - x->send_start[p]) * es;
        ro[p] = (size_t)x->recv_start[p] * es; rb[p] = (size_t)(x->recv_start[p + 1] - x->recv_start[p]) * es;
      }
      xs.push_back(XPiece{(const uint8_t*)x->sdata[k], (uint8_t*)out_data_host[k], so.data(), sb.data(), ro.data(), rb.data()});
    }
    if (rc == DBHIP_OK && x->svalid[k]) rc = add_bits(x->svalid[k], out_validity_host[k]);
  }
  // the packed long-string bytes of every String column: one more piece per column in the same group
  std::vector<std::vector<size_t>> soffs;
  soffs.reserve(x->str_cols.size() * 4);
  for (size_t j = 0; j < x->str_cols.size() && rc == DBHIP_OK; ++j) {
    // (posted for EVERY String column, also when this rank neither sends nor receives long bytes for it: the loopback rendezvous
    // needs the same number of pieces from every rank, and another rank may well move bytes of this column; zero-byte transfers
    // are skipped by both transports and their pointers never dereferenced)
    soffs.emplace_back(W); soffs.emplace_back(W); soffs.emplace_back(W); soffs.emplace_back(W);
    std::vector<size_t>&so = soffs[soffs.size() - 4], &sb = soffs[soffs.size() - 3], &ro = soffs[soffs.size() - 2], &rb = soffs[soffs.size() - 1];
    for (int p = 0; p < W; ++p) {
      so[p] = (size_t)x->sbyte_start[j][p]; sb[p] = (size_t)(x->sbyte_start[j][p + 1] - x->sbyte_start[j][p]);
      ro[p] = (size_t)x->rbyte_start[j][p]; rb[p] = (size_t)(x->rbyte_start[j][p + 1] - x->rbyte_start[j][p]);
    }
    uint8_t* out = out_string_bytes_host ? out_string_bytes_host[x->str_cols[j]] : nullptr;
    xs.push_back(XPiece{x->spacked[j], out, so.data(), sb.data(), ro.data(), rb.data()});
  }
  if (rc == DBHIP_OK) rc = alltoallv_group(c, xs, s);   // ONE group: every column, every peer
  // received views: offsets inside the sender's piece -> inside the packed output buffer (buffer 0 of the received column)
  for (size_t j = 0; j < x->str_cols.size() && rc == DBHIP_OK; ++j) {
    if (x->rbyte_start[j][W] == 0 || x->recv_total == 0) continue;
    int64_t* rs = nullptr;
    rc = dbhip_alloc((size_t)(W + 1) * 16 + 8, (void**)&rs);
    if (rc) break;
    uint64_t* bs = (uint64_t*)(rs + W + 1);
    uint32_t* err = (uint32_t*)(bs + W + 1);
    rc = dbhip_memcpy_h2d(rs, x->recv_start.data(), (size_t)(W + 1) * 8, stream);
    if (rc == DBHIP_OK) rc = dbhip_memcpy_h2d(bs, x->rbyte_start[j].data(), (size_t)(W + 1) * 8, stream);
    if (rc == DBHIP_OK && hipMemsetAsync(err, 0, 4, s) != hipSuccess) rc = DBHIP_ERR_HIP;
    if (rc == DBHIP_OK)
      hipLaunchKernelGGL(xs_rebase_kernel, dim3(grid_for(x->recv_total, 256)), dim3(256), 0, s, (uint32_t*)out_data_host[x->str_cols[j]], x->recv_total,
                         rs, bs, W, err);
    uint32_t e = 0;
    if (rc == DBHIP_OK) rc = dbhip_memcpy_d2h(&e, err, 4, stream);
    (void)dbhip_free(rs);
    if (rc == DBHIP_OK && e) { set_error("dbhip_exchange_finish: column %d receives 4 GiB or more of string bytes (a view's offset is 32 bits)", x->str_cols[j]); rc = DBHIP_ERR_CAPACITY; }
  }
  for (size_t j = 0; j < jobs.size() && rc == DBHIP_OK; ++j) {
    std::vector<dbhip_col> pc(W);
    std::vector<int64_t> rows(W);
    for (int p = 0; p < W; ++p) {
      memset(&pc[p], 0, sizeof(dbhip_col));
      pc[p].type = DBHIP_T_BOOL;
      pc[p].data = jobs[j].tmp + ro_b[p];
      rows[p] = x->recv_start[p + 1] - x->recv_start[p];
    }
    if (x->recv_total > 0) rc = dbhip_concat_columns(pc.data(), rows.data(), nullptr, W, jobs[j].out, nullptr, nullptr, nullptr, stream);
  }
  if (rc == DBHIP_OK && hipStreamSynchronize(s) != hipSuccess) rc = DBHIP_ERR_HIP;   // the temporaries go away below
  for (uint8_t* t : tmp_bits) (void)dbhip_free(t);
  return rc;
}

int32_t dbhip_vec_topk_allgather(dbhip_comm* c, const uint32_t* idx_dev, const float* dist_dev, int32_t nq, int32_t k, uint64_t row_offset,
                                 uint32_t* out_idx_dev, float* out_dist_dev, void* stream) {
  DBHIP_REQUIRE(c && nq >= 0 && k >= 1, "dbhip_vec_topk_allgather: bad argument");
  if (nq == 0) return DBHIP_OK;
  DBHIP_REQUIRE(idx_dev && dist_dev && out_idx_dev && out_dist_dev, "dbhip_vec_topk_allgather: NULL argument");
  DBHIP_REQUIRE(row_offset < 0xFFFFFFFFULL, "dbhip_vec_topk_allgather: the shard's row offset does not fit the u32 id space");
  const int W = c->world;
  const int64_t per = (int64_t)nq * k;
  hipStream_t s = resolve_stream(stream);
  // staging: [mine: ids | dists] [gathered ids: W x per] [gathered dists] [regrouped ids] [regrouped dists]
  uint8_t* ws = (uint8_t*)scratch((size_t)per * 8 + (size_t)W * per * 16 + 512, 18, s);
  if (!ws) return DBHIP_ERR_HIP;
  uint32_t* overflow = (uint32_t*)(ws + (((size_t)per * 8 + (size_t)W * per * 16 + 255) & ~(size_t)255));
  DBHIP_CHECK(hipMemsetAsync(overflow, 0, 4, s));
  uint32_t* my_i = (uint32_t*)ws;
  uint32_t* all_i = my_i + per;
  float* all_d = (float*)(all_i + (size_t)W * per);
  uint32_t* grp_i = (uint32_t*)(all_d + (size_t)W * per);
  float* grp_d = (float*)(grp_i + (size_t)W * per);
  hipLaunchKernelGGL(topk_globalise_kernel, dim3(grid_for(per, 256)), dim3(256), 0, s, idx_dev, per, row_offset, my_i, overflow);
  DBHIP_LAUNCH_CHECK();
  if (c->loop) {
    LoopOp a; a.kind = 1; a.send = (const uint8_t*)my_i; a.recv = (uint8_t*)all_i; a.bytes = (size_t)per * 4;
    int32_t rc = loop_collective(c, a, s);
    if (rc) return rc;
    LoopOp b; b.kind = 1; b.send = (const uint8_t*)dist_dev; b.recv = (uint8_t*)all_d; b.bytes = (size_t)per * 4;
    if ((rc = loop_collective(c, b, s))) return rc;
  } else if (const int32_t u = comm_usable(c, "dbhip_vec_topk_allgather")) {
    return u;
  } else if (!c->comm) {
    DBHIP_CHECK(hipMemcpyAsync(all_i, my_i, (size_t)per * 4, hipMemcpyDeviceToDevice, s));
    DBHIP_CHECK(hipMemcpyAsync(all_d, dist_dev, (size_t)per * 4, hipMemcpyDeviceToDevice, s));
  } else {   // both all-gathers in one group: one launch on the wire
    NCCL_CHECK(g_rccl.GroupStart());
    const int r1 = g_rccl.AllGather(my_i, all_i, (size_t)per * 4, NCCL_UINT8, c->comm, s);
    const int r2 = r1 == NCCL_SUCCESS ? g_rccl.AllGather(dist_dev, all_d, (size_t)per * 4, NCCL_UINT8, c->comm, s) : NCCL_SUCCESS;
    const int e = g_rccl.GroupEnd();
    if (r1 != NCCL_SUCCESS) return nccl_fail(r1, "ncclAllGather");
    if (r2 != NCCL_SUCCESS) return nccl_fail(r2, "ncclAllGather");
    if (e != NCCL_SUCCESS) return nccl_fail(e, "ncclGroupEnd");
  }
  hipLaunchKernelGGL(topk_regroup_kernel<uint32_t>, dim3(grid_for((int64_t)W * per, 256)), dim3(256), 0, s, all_i, W, nq, k, grp_i);
  hipLaunchKernelGGL(topk_regroup_kernel<float>, dim3(grid_for((int64_t)W * per, 256)), dim3(256), 0, s, all_d, W, nq, k, grp_d);
  DBHIP_LAUNCH_CHECK();
  int32_t rc = dbhip_vec_topk_merge(grp_d, grp_i, (int64_t)W * k, nq, k, out_idx_dev, out_dist_dev, stream);
  if (rc) return rc;
  uint32_t over = 0;
  DBHIP_CHECK(hipMemcpyAsync(&over, overflow, 4, hipMemcpyDeviceToHost, s));
  DBHIP_CHECK(hipStreamSynchronize(s));   // scratch is reused by the next call
  if (over) {   // (found after the collectives so that every rank takes part in them whatever its own ids are)
    set_error("dbhip_vec_topk_allgather: a row id of this shard + its row offset %llu does not fit the u32 id space", (unsigned long long)row_offset);
    return DBHIP_ERR_INVALID;
  }
  return DBHIP_OK;
}

int32_t dbhip_comm_unique_id(uint8_t* out_id128_host) {
  DBHIP_REQUIRE(out_id128_host, "dbhip_comm_unique_id: NULL argument");
  int32_t rc = load_rccl();
  if (rc) return rc;
  ncclUniqueId id;
  NCCL_CHECK(g_rccl.GetUniqueId(&id));
  memcpy(out_id128_host, id.internal, 128);
  return DBHIP_OK;
}

int32_t dbhip_comm_create(int32_t rank, int32_t world, const uint8_t* id128_host, dbhip_comm** out_host) {
  DBHIP_REQUIRE(out_host && world >= 1 && rank >= 0 && rank < world && (world == 1 || id128_host), "dbhip_comm_create: bad argument");
  dbhip_comm* c = new (std::nothrow) dbhip_comm();
  if (!c) return DBHIP_ERR_HIP;
  c->rank = rank; c->world = world;
  if (world > 1 || id128_host) {   // (a world of one without an id stays a local object: nothing to load)
    int32_t rc = load_rccl();
    if (rc) { delete c; return rc; }
    ncclUniqueId id;
    memcpy(id.internal, id128_host, 128);
    const int r = g_rccl.CommInitRank(&c->comm, world, id, rank);   // on the calling thread's current device
    if (r != NCCL_SUCCESS) { delete c; return nccl_fail(r, "ncclCommInitRank"); }
  }
  *out_host = c;
  return DBHIP_OK;
}

static std::mutex g_loop_mu;
static std::map<uint64_t, LoopGroup*> g_loops;
int32_t dbhip_comm_create_loopback(uint64_t group_id, int32_t rank, int32_t world, dbhip_comm** out_host) {
  DBHIP_REQUIRE(out_host && world >= 1 && world <= 64 && rank >= 0 && rank < world, "dbhip_comm_create_loopback: bad argument");
  dbhip_comm* c = new (std::nothrow) dbhip_comm();
  if (!c) return DBHIP_ERR_HIP;
  std::lock_guard<std::mutex> lk(g_loop_mu);
  LoopGroup*& g = g_loops[group_id];
  if (!g) { g = new (std::nothrow) LoopGroup(); if (!g) { delete c; return DBHIP_ERR_HIP; } g->world = world; g->ops.resize(world); }
  if (g->world != world) { delete c; set_error("dbhip_comm_create_loopback: group %llu exists with another world size", (unsigned long long)group_id); return DBHIP_ERR_INVALID; }
  ++g->refs;
  c->loop = g; c->rank = rank; c->world = world;
  *out_host = c;
  return DBHIP_OK;
}

int32_t dbhip_comm_destroy(dbhip_comm* c) {
  if (!c) return DBHIP_OK;
  if (c->loop) {
    std::lock_guard<std::mutex> lk(g_loop_mu);
    if (--c->loop->refs == 0) {
      for (auto it = g_loops.begin(); it != g_loops.end(); ++it) if (it->second == c->loop) { g_loops.erase(it); break; }
      delete c->loop;
    }
  }
  if (c->comm) (void)g_rccl.CommDestroy(c->comm);
  if (c->send) (void)dbhip_free(c->send);
  if (c->recv) (void)dbhip_free(c->recv);
  delete c;
  return DBHIP_OK;
}

int32_t dbhip_comm_abort(dbhip_comm* c) {
  DBHIP_REQUIRE(c, "dbhip_comm_abort: NULL argument");
  if (c->loop) { loop_abort(c, "dbhip_comm_abort"); return DBHIP_OK; }
  if (!c->comm) { c->aborted = c->world > 1; return DBHIP_OK; }   // (a world of one has nobody to tell)
  c->aborted = true;   // whatever happens below: no later collective of this handle runs
  if (!g_rccl.CommAbort) {
    set_error("dbhip_comm_abort: this librccl has no ncclCommAbort; the communicator is marked aborted on this rank only — peers blocked "
              "in a collective stay blocked");
    return DBHIP_ERR_UNSUPPORTED;
  }
  (void)g_rccl.CommAbort(c->comm);
  c->comm = nullptr;
  return DBHIP_OK;
}

int32_t dbhip_comm_allgather(dbhip_comm* c, const void* send_dev, void* recv_dev, int64_t bytes_per_rank, void* stream) {
  DBHIP_REQUIRE(c && send_dev && recv_dev && bytes_per_rank >= 0, "dbhip_comm_allgather: bad argument");
  hipStream_t s = resolve_stream(stream);
  if (bytes_per_rank == 0) return DBHIP_OK;
  if (c->loop) { LoopOp a; a.kind = 1; a.send = (const uint8_t*)send_dev; a.recv = (uint8_t*)recv_dev; a.bytes = (size_t)bytes_per_rank; return loop_collective(c, a, s); }
  if (const int32_t u = comm_usable(c, "dbhip_comm_allgather")) return u;
  if (!c->comm) {
    DBHIP_CHECK(hipMemcpyAsync((uint8_t*)recv_dev + (size_t)c->rank * bytes_per_rank, send_dev, (size_t)bytes_per_rank, hipMemcpyDeviceToDevice, s));
    return DBHIP_OK;
  }
  NCCL_CHECK(g_rccl.AllGather(send_dev, recv_dev, (size_t)bytes_per_rank, NCCL_UINT8, c->comm, s));
  return DBHIP_OK;
}

int32_t dbhip_comm_alltoall(dbhip_comm* c, const void* send_dev, void* recv_dev, int64_t bytes_per_peer, void* stream) {
  DBHIP_REQUIRE(c && send_dev && recv_dev && bytes_per_peer >= 0, "dbhip_comm_alltoall: bad argument");
  if (bytes_per_peer == 0) return DBHIP_OK;
  return alltoall_bytes(c, send_dev, recv_dev, (size_t)bytes_per_peer, resolve_stream(stream));
}

int32_t dbhip_comm_allreduce_sum_u64(dbhip_comm* c, const uint64_t* send_dev, uint64_t* recv_dev, int64_t count, void* stream) {
  DBHIP_REQUIRE(c && send_dev && recv_dev && count >= 0, "dbhip_comm_allreduce_sum_u64: bad argument");
  hipStream_t s = resolve_stream(stream);
  if (count == 0) return DBHIP_OK;
  if (c->loop) { set_error("dbhip_comm_allreduce_sum_u64: not part of the in-process loopback world"); return DBHIP_ERR_UNSUPPORTED; }
  if (const int32_t u = comm_usable(c, "dbhip_comm_allreduce_sum_u64")) return u;
  if (!c->comm) {
    if (send_dev != recv_dev) DBHIP_CHECK(hipMemcpyAsync(recv_dev, send_dev, (size_t)count * 8, hipMemcpyDeviceToDevice, s));
    return DBHIP_OK;
  }
  NCCL_CHECK(g_rccl.AllReduce(send_dev, recv_dev, (size_t)count, NCCL_UINT64, NCCL_SUM, c->comm, s));
  return DBHIP_OK;
}

int32_t dbhip_groupby_exchange_allgather(dbhip_groupby* g, dbhip_comm* c, int64_t max_rows, void* stream) {
  DBHIP_REQUIRE(g && c && max_rows >= 1, "dbhip_groupby_exchange_allgather: bad argument");
  int64_t row_bytes = 0;
  int32_t rc = dbhip_groupby_row_bytes(g, &row_bytes);
  if (rc) return rc;
  const size_t block = (size_t)(max_rows + 1) * (size_t)row_bytes;
  if ((rc = ensure_staging(c, block * (size_t)c->world))) return rc;
  if ((rc = dbhip_groupby_flush_block(g, c->send, max_rows, stream))) return rc;       // header + rows, no host sync
  if ((rc = dbhip_comm_allgather(c, c->send, c->recv, (int64_t)block, stream))) return rc;
  // the other ranks' blocks merge into this table (its own states never left it); an overflowed block anywhere is reported
  // (DBHIP_ERR_CAPACITY) before the table is touched — every rank sees the same headers and takes the same decision
  return dbhip_groupby_merge_blocks(g, c->recv, c->world, max_rows, c->rank, stream);
}

int32_t dbhip_groupby_exchange_alltoall(dbhip_groupby* g, dbhip_comm* c, int64_t max_rows, void* stream) {
  DBHIP_REQUIRE(g && c && max_rows >= 1, "dbhip_groupby_exchange_alltoall: bad argument");
  int64_t row_bytes = 0;
  int32_t rc = dbhip_groupby_row_bytes(g, &row_bytes);
  if (rc) return rc;
  const size_t block = (size_t)(max_rows + 1) * (size_t)row_bytes;
  if ((rc = ensure_staging(c, block * (size_t)c->world))) return rc;
  if ((rc = dbhip_groupby_partition_blocks(g, c->world, c->send, max_rows, stream))) return rc;   // hash % world, on the device
  if ((rc = dbhip_comm_alltoall(c, c->send, c->recv, (int64_t)block, stream))) return rc;
  return dbhip_groupby_replace_with_blocks(g, c->recv, c->world, max_rows, stream);
}

}  // extern "C"

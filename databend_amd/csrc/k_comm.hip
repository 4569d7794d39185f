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

#include <dlfcn.h>
#include <string.h>

#include <new>

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

struct dbhip_comm {
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1;
  void* send = nullptr;      // exchange staging (blocks), grown on demand
  void* recv = nullptr;
  size_t cap = 0;
};

namespace {

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
int32_t alltoall_bytes(dbhip_comm* c, const void* send, void* recv, size_t bytes_per_peer, hipStream_t s) {
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

}  // namespace

extern "C" {

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

int32_t dbhip_comm_destroy(dbhip_comm* c) {
  if (!c) return DBHIP_OK;
  if (c->comm) (void)g_rccl.CommDestroy(c->comm);
  if (c->send) (void)dbhip_free(c->send);
  if (c->recv) (void)dbhip_free(c->recv);
  delete c;
  return DBHIP_OK;
}

int32_t dbhip_comm_allgather(dbhip_comm* c, const void* send_dev, void* recv_dev, int64_t bytes_per_rank, void* stream) {
  DBHIP_REQUIRE(c && send_dev && recv_dev && bytes_per_rank >= 0, "dbhip_comm_allgather: bad argument");
  hipStream_t s = resolve_stream(stream);
  if (bytes_per_rank == 0) return DBHIP_OK;
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

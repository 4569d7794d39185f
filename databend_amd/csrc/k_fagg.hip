// k_fagg.hip — generic fused TransformFilter -> BlockOperator::Map -> TransformPartialAggregate for tables with a
// handful of groups (SURVEY §8 a1/a6/a10/a11 + §8f-2): ONE pass over the unfiltered input columns.
//
// Reference pipeline (three processors, one materialised column per call node between them):
//   TransformFilter            filter/filter_executor.rs:81-118 -> selection -> DataBlock::take of every column
//   CompoundBlockOperator      sql/src/evaluator/block_operator.rs:42-85 (Evaluator::run per expression)
//   TransformPartialAggregate  aggregate_hashtable.rs:168-292 (hash -> probe -> accumulate_keys)
// Here the binding hands over the expression program (filter root + one root per aggregate argument, dev_expr.h) and
// the kernel, per chunk of 2 x 64 rows per wave:
//   loads every input column once (coalesced), interprets the program over the LDS register file,
//   resolves each passing row's key words to one of <= 8 group slots through a tiny per-workgroup LDS key table that
//     every wave caches in scalar registers (the device analogue of a cache-resident partial AggregateHashTable),
//   and accumulates the argument values into PER-LANE register accumulators (no atomics, no cross-lane traffic in
//     the loop); one wave reduction per touched slot at the very end -> <= 8 partial rows per wave, merged into the
//     HBM table by the row path exactly like partial payloads in TransformFinalAggregate.
// This is the query-specific k_q1.hip generalised: any <= 4 key words, <= 8 aggregates (count / sum incl. exact
// Decimal128 / min / max, nullable arguments), any expression program dev_expr.h interprets. A workgroup that meets a
// 9th distinct key gives up (DBHIP_ERR_CAPACITY, nothing merged): the caller keeps the operator-at-a-time kernels.
#include "fagg_device.h"
#include "runtime.h"

#include <dirent.h>
#include <dlfcn.h>
#include <errno.h>
#include <sys/stat.h>
#include <fcntl.h>
#include <signal.h>
#include <spawn.h>
#include <sys/wait.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <deque>
#include <map>
#include <mutex>
#include <string>

#include <stdlib.h>
#include <string.h>
#include <time.h>

using namespace dbhip;

int32_t dbhip_groupby_merge_rows_dev_internal(dbhip_groupby* g, const uint64_t* rows, int64_t n_max, const uint64_t* n_dev,
                                              const uint64_t* abort_dev, hipStream_t s);
int32_t dbhip_groupby_merge_rows_internal(dbhip_groupby* g, const uint64_t* rows, int64_t n, hipStream_t s);
int64_t dbhip_groupby_capacity_internal(dbhip_groupby* g);
int32_t dbhip_groupby_reserve_merge_internal(dbhip_groupby* g, int64_t n);
int64_t dbhip_groupby_count_internal(dbhip_groupby* g);
const GbLayout* dbhip_groupby_layout_internal(dbhip_groupby* g);
int32_t dbhip_groupby_merge_rows_deferred_internal(dbhip_groupby* g, const uint64_t* rows, int64_t n_max, const uint64_t* n_dev,
                                                   const uint64_t* abort_dev, hipStream_t s);
int32_t dbhip_groupby_ensure_room_internal(dbhip_groupby* g, int64_t extra, hipStream_t s);
uint64_t* dbhip_groupby_ctrl_internal(dbhip_groupby* g);
void dbhip_groupby_set_count_internal(dbhip_groupby* g, int64_t count);
void** dbhip_groupby_pipe_slot_internal(dbhip_groupby* g);

namespace {

// ---------------------------------------------------------------------------------------------------------------------
// run-time specialisation (see fagg_device.h): the kernel's own source, compiled by hiprtc — in a helper process, jitc.cpp —
// with this query's program and layout as a constexpr. Sources of the device headers are embedded at build time
// (build/jit_embed.inc, Makefile).
// ---------------------------------------------------------------------------------------------------------------------
#include "build/jit_embed.inc"   // kJitHdrName[], kJitHdrSrc[], kJitHdrCount

struct JitOut {
  std::string s;
  void num(long long v) { s += std::to_string(v); s += ","; }
  void u64(uint64_t v) { char b[40]; snprintf(b, sizeof(b), "0x%llxULL,", (unsigned long long)v); s += b; }
  void null() { s += "nullptr,"; }
  void open() { s += "{"; }
  void close() { s += "},"; }
};

void jit_ins(JitOut& o, const ExIns& I) {
  o.open();
  o.num(I.op); o.num(I.dst); o.num(I.a); o.num(I.b); o.num(I.c); o.num(I.type);
  o.num(I.acls); o.num(I.bcls); o.num(I.ocls); o.num(I.norm_sh); o.num(I.norm_signed); o.num(I.norm_f32);
  o.num(I.a_wide); o.num(I.b_wide); o.num(I.o_wide); o.num(I.a_dec); o.num(I.b_dec); o.num(I.dec_idx); o.num(I.dep); o.num(0);
  o.u64(I.imm); o.u64(I.imm_hi);
  o.close();
}
void jit_dec(JitOut& o, const DecOp& D) {
  o.open();
  o.num(D.op); o.num(D.a_from_scale); o.num(D.a_to_scale); o.num(D.a_to_precision); o.num(D.a_check);
  o.num(D.b_from_scale); o.num(D.b_to_scale); o.num(D.b_to_precision); o.num(D.b_check);
  o.num(D.t_is_128); o.num(D.ret_precision); o.num(D.ret_scale); o.num(D.overflow); o.num(D.scale_mul); o.num(D.trivial);
  o.close();
}
// `static constexpr FaArgs kM = {...};` in declaration order (ExProg, dev_expr.h; FaArgs, fagg_device.h); every pointer,
// offset and count that differs between calls of the same query shape is null / 0 here and read from the kernel argument
// rows per lane of the specialised kernel: >= 256 bytes per lane in flight (all loads of a chunk are issued up front)
int jit_rows(const FaArgs& A) {
  int bytes = 0;
  for (int c = 0; c < A.P.n_inputs; ++c) {
    switch (A.P.in_type[c]) {
      case LK_8: bytes += 8; break;
      case LK_16: bytes += 16; break;
      case LK_S4: case LK_U4: case LK_F4: bytes += 4; break;
      case LK_S2: case LK_U2: bytes += 2; break;
      default: bytes += 1; break;
    }
  }
  for (int k = 0; k < A.nkeys; ++k) bytes += type_size(A.key[k].type) > 0 ? type_size(A.key[k].type) : 1;
  // 64-byte rows and wider: THREE rows per lane (round 5). With 4 (a 256-row chunk: every column of a wave's chunk starts on a multiple of
  // 1 / 2 / 4 KiB) the Q1 kernel ran 6.45 .. 7.6 ms from process to process on one box — the seven column streams fall on the same
  // HBM channels or not, depending on where the allocator put the columns — against 6.30 .. 6.50 ms with 3 (192-row chunks), every
  // run; 2 measured 6.8 ms, 1 8.7 ms (profiles/r05_fagg_rows_sweep.txt).
  // a program that divides (a rounding decimal multiply, a decimal divide) is ALU bound, not load bound: with the 16 row slots its narrow
  // rows would get, the kernel needs 310 VGPRs (one wave per SIMD) and 3 s of hiprtc; 3 slots: 10.4 ms instead of 15.2 ms on the rescaling
  // Q1 variant, 1.1 s cold PREPARE (r05 sweep: 2 -> 10.9 ms / 0.85 s, 3 -> 10.4 / 1.14, 4 -> 10.3 / 1.42, 6 -> 14.4 / 2.2, 8 -> 15.2 / 3.0)
  for (int i = 0; i < A.P.n_ins; ++i)
    if (A.P.ins[i].op == EX_DEC && dec_op_needs_division(A.P.dec[A.P.ins[i].dec_idx])) return 3;
  return bytes >= 64 ? 3 : (bytes >= 32 ? 8 : 16);
}
std::string jit_meta(const FaArgs& A, bool multi = false) {
  JitOut o;
  o.s = std::string(multi ? "#define FA_MULTI 1\n" : "") + "#define FA_META_ROWS " + std::to_string(jit_rows(A)) + "\nstatic constexpr FaArgs kM = {";
  const ExProg& P = A.P;
  o.open();                                                           // ExProg
  o.open(); for (int i = 0; i < EX_MAX_INS; ++i) jit_ins(o, P.ins[i]); o.close();
  o.open(); for (int i = 0; i < EX_MAX_DEC; ++i) jit_dec(o, P.dec[i]); o.close();
  o.open(); for (int i = 0; i < EX_MAX_INPUTS; ++i) o.null(); o.close();   // in_data
  o.open(); for (int i = 0; i < EX_MAX_INPUTS; ++i) o.null(); o.close();   // in_valid
  o.open(); for (int i = 0; i < EX_MAX_INPUTS; ++i) o.num(0); o.close();   // in_voff
  o.open(); for (int i = 0; i < EX_MAX_INPUTS; ++i) o.num(P.in_type[i]); o.close();
  o.open(); for (int i = 0; i < EX_MAX_INPUTS; ++i) o.num(P.in_scalar[i]); o.close();
  o.open(); for (int i = 0; i < EX_MAX_INPUTS; ++i) o.num(P.in_slot[i]); o.close();
  o.open(); for (int i = 0; i < EX_MAX_INPUTS; ++i) o.num(P.in_wide_ord[i]); o.close();
  o.open(); for (int i = 0; i < EX_MAX_INPUTS; ++i) o.num(P.in_has_valid[i]); o.close();
  o.num(P.n_ins); o.num(P.n_inputs); o.num(P.n_slots); o.num(P.n_filter_ins); o.num(P.filter_slot); o.num(P.filter_dep);
  o.null(); o.null();
  o.close();
  o.open();                                                           // key[FA_KW]
  for (int k = 0; k < FA_KW; ++k) { o.open(); o.null(); o.null(); o.num(0); o.null(); o.num(A.key[k].type); o.num(A.key[k].is_scalar); o.close(); }
  o.close();
  o.open(); for (int k = 0; k < FA_KW; ++k) o.num(A.key_type[k]); o.close();
  o.open(); for (int k = 0; k < FA_KW; ++k) o.num(A.key_off[k]); o.close();
  o.open(); for (int k = 0; k < FA_KW; ++k) o.num(A.key_words[k]); o.close();
  o.open(); for (int k = 0; k < FA_KW; ++k) o.num(A.key_has_valid[k]); o.close();
  o.num(A.has_filter);
  o.num(A.nkeys); o.num(A.nkey_words); o.num(A.validity_word); o.num(A.hash_word); o.num(A.W);
  o.num(A.naggs); o.num(A.nwords); o.num(A.state_off);
  o.open(); for (int w = 0; w < FA_MAXW; ++w) o.num(A.wm[w]); o.close();
  o.null(); o.num(0); o.num(0); o.null(); o.null();
  o.null(); o.num(0); o.num(0);   // blocks, wgs_per_block, _pad
  o.s += "};\n";
  return o.s;
}

const char* const kJitPrelude =
    "#define DBHIP_JIT 1\n"
    "typedef unsigned long uint64_t; typedef long int64_t; typedef unsigned int uint32_t; typedef int int32_t;\n"
    "typedef unsigned short uint16_t; typedef short int16_t; typedef unsigned char uint8_t; typedef signed char int8_t;\n"
    "typedef unsigned long size_t; typedef unsigned long uintptr_t;\n"
    "#define INT64_MAX 9223372036854775807L\n#define INT64_MIN (-9223372036854775807L - 1)\n"
    "#define INT32_MAX 2147483647\n#define INT32_MIN (-2147483647 - 1)\n#define UINT64_MAX 18446744073709551615UL\n"
    "#define UINT32_MAX 4294967295U\n";

std::string jit_tail(int slots, bool general, int nw) {
  char tail[256];
  snprintf(tail, sizeof(tail), "extern \"C\" __global__ __launch_bounds__(256) void fagg_jit(FaArgs A) { fagg_body<%d, %s, %d>(A); }\n", slots,
           general ? "true" : "false", nw);
  return tail;
}
// (meta, variant) -> code object through the out-of-process compiler driver `dbhip_jitc` (jitc.cpp: why it is a process of
// its own) next to libdbhip.so: the sources go to a fresh temporary directory, the helper gets `deadline_s` seconds. false + log
// on any failure (no helper, compile error, deadline). Needs no device.
std::string jit_helper_path() {
  Dl_info info;
  if (!dladdr((const void*)&jit_helper_path, &info) || !info.dli_fname) return "";
  std::string p = info.dli_fname;
  const size_t slash = p.rfind('/');
  return (slash == std::string::npos ? std::string(".") : p.substr(0, slash)) + "/dbhip_jitc";
}
bool write_file(const std::string& path, const char* data, size_t n) {
  FILE* f = fopen(path.c_str(), "wb");
  if (!f) return false;
  const bool ok = fwrite(data, 1, n, f) == n;
  fclose(f);
  return ok;
}
// gfx target of the code object: the device the library drives (hipGetDeviceProperties().gcnArchName up to the first ':'),
// DBHIP_JIT_ARCH to override, gfx950 when no device is visible (offline compile checks)
std::string jit_arch() {
  if (const char* e = getenv("DBHIP_JIT_ARCH")) return e;
  int dev = 0;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.gcnArchName[0]) {
    std::string a = prop.gcnArchName;
    const size_t c = a.find(':');
    return c == std::string::npos ? a : a.substr(0, c);
  }
  return "gfx950";
}

uint64_t fnv1a(const void* data, size_t n, uint64_t h) {
  const unsigned char* p = (const unsigned char*)data;
  for (size_t i = 0; i < n; ++i) { h ^= p[i]; h *= 0x100000001b3ULL; }
  return h;
}
// On-disk cache of code objects, keyed by everything that determines the code: generated metadata + variant, the embedded
// device headers, the compiler flags and the target. DBHIP_JIT_CACHE_DIR names the directory ("" / "off" disables); default
// $XDG_CACHE_HOME/dbhip or ~/.cache/dbhip. A new process finds the kernels of the query shapes an earlier one PREPAREd and
// pays a file read (tens of microseconds) instead of a 0.5 s compile.
std::string jit_cache_dir() {
  if (const char* e = getenv("DBHIP_JIT_CACHE_DIR")) return (!*e || !strcmp(e, "off")) ? std::string() : std::string(e);
  if (const char* x = getenv("XDG_CACHE_HOME")) if (*x) return std::string(x) + "/dbhip";
  if (const char* h = getenv("HOME")) if (*h) return std::string(h) + "/.cache/dbhip";
  return std::string();
}
std::string jit_cache_path(const std::string& meta, const std::string& tail, const std::string& arch, const std::string& defs) {
  const std::string dir = jit_cache_dir();
  if (dir.empty()) return dir;
  uint64_t h1 = 0xcbf29ce484222325ULL, h2 = 0x84222325cbf29ce4ULL;
  auto mix = [&](const void* d, size_t n) { h1 = fnv1a(d, n, h1); h2 = fnv1a(d, n, h2 ^ 0x9e3779b97f4a7c15ULL); };
  mix(meta.data(), meta.size()); mix(tail.data(), tail.size()); mix(arch.data(), arch.size()); mix(defs.data(), defs.size());
  for (int i = 0; i < kJitHdrCount; ++i) mix(kJitHdrSrc[i], strlen(kJitHdrSrc[i]));
  char name[64];
  snprintf(name, sizeof(name), "/fagg_%016llx%016llx.co", (unsigned long long)h1, (unsigned long long)h2);
  return dir + name;
}
bool read_file(const std::string& path, std::vector<char>* out) {
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) return false;
  char buf[65536];
  size_t n;
  while ((n = fread(buf, 1, sizeof(buf), f)) > 0) out->insert(out->end(), buf, buf + n);
  fclose(f);
  return !out->empty();
}
// A cached code object is GPU code that will run inside the host's process: it is only loaded from a directory and a file that
// belong to this user and that nobody else can write (a shared or world-writable cache directory would let another user plant a
// kernel), and only if it looks like a complete code object (ELF magic; the helper publishes by rename, so a short file is a
// crashed writer). Anything else is a cache MISS: the file is removed where that is allowed and the shape is compiled again.
bool jit_cache_trusted(const std::string& dir, int fd) {
  struct stat ds, fs;
  if (stat(dir.c_str(), &ds) != 0 || !S_ISDIR(ds.st_mode) || ds.st_uid != geteuid() || (ds.st_mode & (S_IWGRP | S_IWOTH))) return false;
  if (fstat(fd, &fs) != 0 || !S_ISREG(fs.st_mode) || fs.st_uid != geteuid() || (fs.st_mode & (S_IWGRP | S_IWOTH))) return false;
  return true;
}
bool read_cached_code(const std::string& path, std::vector<char>* out) {
  const int fd = open(path.c_str(), O_RDONLY | O_NOFOLLOW | O_CLOEXEC);
  if (fd < 0) return false;
  const size_t slash = path.rfind('/');
  bool ok = jit_cache_trusted(slash == std::string::npos ? std::string(".") : path.substr(0, slash), fd);
  if (ok) {
    char buf[65536];
    ssize_t n;
    while ((n = read(fd, buf, sizeof(buf))) > 0) out->insert(out->end(), buf, buf + n);
    ok = out->size() > 64 && !memcmp(out->data(), "\177ELF", 4);
  }
  close(fd);
  if (!ok) { out->clear(); (void)unlink(path.c_str()); }   // miss (untrusted, truncated or foreign): compile again
  return ok;
}
// build_XXXXXX directories of helpers that were killed before they could clean up (older than 15 minutes)
void sweep_stale_builds(const std::string& dir) {
  DIR* d = opendir(dir.c_str());
  if (!d) return;
  const time_t now = time(nullptr);
  while (struct dirent* e = readdir(d)) {
    if (strncmp(e->d_name, "build_", 6) != 0) continue;
    const std::string sub = dir + "/" + e->d_name;
    struct stat st;
    if (lstat(sub.c_str(), &st) != 0 || !S_ISDIR(st.st_mode) || st.st_uid != geteuid() || now - st.st_mtime < 15 * 60) continue;
    if (DIR* b = opendir(sub.c_str())) {
      while (struct dirent* f = readdir(b)) if (strcmp(f->d_name, ".") && strcmp(f->d_name, "..")) (void)unlink((sub + "/" + f->d_name).c_str());
      closedir(b);
    }
    (void)rmdir(sub.c_str());
  }
  closedir(d);
}
void mkdir_p(const std::string& dir) {
  for (size_t i = 1; i <= dir.size(); ++i)
    if (i == dir.size() || dir[i] == '/') (void)mkdir(dir.substr(0, i).c_str(), 0700);
}

// detached = true: start the compile and return at once (false, *log = "pending"); the helper publishes the code object into the
// on-disk cache when it is done, where the next lookup finds it. Needs the cache directory (the temporary directory with the
// sources lives inside it, so that the helper's rename stays on one file system).
bool jit_compile(const std::string& meta, const std::string& tail, std::vector<char>* code, std::string* log, bool detached = false) {
  const std::string arch = jit_arch();
  const std::string defs = exp_env("DBHIP_FAGG_JIT_DEFS") ? exp_env("DBHIP_FAGG_JIT_DEFS") : "";
  const std::string cached = jit_cache_path(meta, tail, arch, defs);
  if (!cached.empty() && read_cached_code(cached, code)) return true;   // compiled by an earlier process (or an earlier table)
  code->clear();
  const std::string helper = jit_helper_path();
  if (helper.empty() || access(helper.c_str(), X_OK) != 0) { *log = "dbhip_jitc not found next to libdbhip.so (" + helper + ")"; return false; }
  if (detached && cached.empty()) { *log = "no on-disk cache directory (DBHIP_JIT_CACHE_DIR): nothing to publish a background compile into"; return false; }
  if (detached) {
    mkdir_p(jit_cache_dir());
    static bool swept = false;
    if (!swept) { swept = true; sweep_stale_builds(jit_cache_dir()); }
  }
  std::string tmpl_s = detached ? jit_cache_dir() + "/build_XXXXXX" : std::string("/tmp/dbhip_jit_XXXXXX");
  if (!mkdtemp(&tmpl_s[0])) { *log = "mkdtemp failed"; return false; }
  const std::string dir = tmpl_s;
  auto cleanup = [&] {
    for (int i = 0; i < kJitHdrCount; ++i) unlink((dir + "/" + kJitHdrName[i]).c_str());
    unlink((dir + "/fagg_meta.inc").c_str()); unlink((dir + "/main.hip").c_str()); unlink((dir + "/out.co").c_str()); unlink((dir + "/log.txt").c_str());
    rmdir(dir.c_str());
  };
  bool ok = true;
  for (int i = 0; i < kJitHdrCount && ok; ++i) ok = write_file(dir + "/" + kJitHdrName[i], kJitHdrSrc[i], strlen(kJitHdrSrc[i]));
  const std::string src = std::string(kJitPrelude) + "#include \"dbhip.h\"\n#include \"fagg_device.h\"\n" + tail;
  ok = ok && write_file(dir + "/fagg_meta.inc", meta.data(), meta.size()) && write_file(dir + "/main.hip", src.data(), src.size());
  if (!ok) { cleanup(); *log = "cannot write the sources to " + dir; return false; }
  std::vector<std::string> args = {helper};
  if (detached) { args.push_back("--detach"); args.push_back("--publish"); args.push_back(cached); args.push_back("--rmdir"); args.push_back(dir); }
  for (const std::string& a : {dir + "/main.hip", dir + "/out.co", "-I" + dir, "--offload-arch=" + arch, std::string("-O3"), std::string("-std=c++17"),
                               std::string("-mllvm"), std::string("-pragma-unroll-threshold=4000000")}) args.push_back(a);
  if (const char* e = exp_env("DBHIP_FAGG_JIT_DEFS")) {   // experiment knobs, e.g. "-DFA_JIT_ROWS=4 -DFA_JIT_GLOBAL" (read per compile)
    std::string t;
    for (const char* c = e;; ++c) {
      if (*c == ' ' || *c == 0) { if (!t.empty()) args.push_back(t); t.clear(); if (!*c) break; }
      else t.push_back(*c);
    }
  }
  std::vector<char*> argv;
  for (std::string& a : args) argv.push_back(&a[0]);
  argv.push_back(nullptr);
  const std::string logpath = dir + "/log.txt";
  posix_spawn_file_actions_t fa;
  posix_spawn_file_actions_init(&fa);
  posix_spawn_file_actions_addopen(&fa, 2, logpath.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0600);
  posix_spawn_file_actions_addopen(&fa, 1, "/dev/null", O_WRONLY, 0);
  pid_t pid = 0;
  const int sp = posix_spawn(&pid, helper.c_str(), &fa, nullptr, argv.data(), ::environ);
  posix_spawn_file_actions_destroy(&fa);
  if (sp != 0) { cleanup(); *log = "posix_spawn(dbhip_jitc) failed"; return false; }
  if (detached) {   // the helper forks and its parent returns at once; the grandchild compiles, publishes and removes `dir`
    int st = 0;
    while (waitpid(pid, &st, 0) < 0 && errno == EINTR) {}
    *log = "pending";
    return false;
  }
  const int deadline_ms = 60 * 1000;
  int status = 0, waited = 0;
  bool done = false, reaped_elsewhere = false;
  while (waited < deadline_ms) {
    const pid_t w = waitpid(pid, &status, WNOHANG);
    if (w == pid) { done = true; break; }
    if (w < 0 && errno == EINTR) continue;
    if (w < 0) { reaped_elsewhere = errno == ECHILD; break; }   // ECHILD: the host ignores SIGCHLD, the child is reaped for us
    usleep(10 * 1000);
    waited += 10;
  }
  if (reaped_elsewhere) {
    // no status to read (and no pid to signal: it may have been reused): wait for out.co to appear complete instead
    for (; waited < deadline_ms && access((dir + "/out.co").c_str(), R_OK) != 0; waited += 10) usleep(10 * 1000);
    usleep(50 * 1000);
    status = 0;
    done = access((dir + "/out.co").c_str(), R_OK) == 0;
  }
  if (!done) {
    if (!reaped_elsewhere) {
      kill(pid, SIGKILL);
      while (waitpid(pid, &status, 0) < 0 && errno == EINTR) {}
    }
    cleanup();
    *log = "dbhip_jitc did not finish within 60 s: killed";
    return false;
  }
  {
    FILE* f = fopen(logpath.c_str(), "rb");
    if (f) { char buf[4096]; size_t n; while ((n = fread(buf, 1, sizeof(buf), f)) > 0) log->append(buf, n); fclose(f); }
  }
  ok = WIFEXITED(status) && WEXITSTATUS(status) == 0;
  if (ok) {
    FILE* f = fopen((dir + "/out.co").c_str(), "rb");
    ok = f != nullptr;
    if (f) { char buf[65536]; size_t n; while ((n = fread(buf, 1, sizeof(buf), f)) > 0) code->insert(code->end(), buf, buf + n); fclose(f); }
    ok = ok && !code->empty();
  }
  if (ok && !cached.empty()) {   // publish atomically: a concurrent reader sees the whole file or none
    mkdir_p(jit_cache_dir());
    const std::string tmp = cached + "." + std::to_string((long long)getpid()) + ".tmp";
    if (write_file(tmp, code->data(), code->size())) { if (rename(tmp.c_str(), cached.c_str()) != 0) unlink(tmp.c_str()); }
  }
  cleanup();
  return ok;
}

// One entry per (metadata, variant). The hiprtc compile (~0.5-1 s) happens in dbhip_groupby_prepare_program — the PREPARE of
// a pipeline — on the calling thread; add_block_program uses a specialised kernel only when it finds one in the cache and
// interprets otherwise, so a query never waits for a compiler (env DBHIP_FAGG_JIT=sync: compile on first use instead).
// (r02j3: compiling on a detached background thread while the main thread kept launching hung inside the ROCm compiler
// library on the GPU box — no thread of this library calls hiprtc concurrently with anything else any more.)
struct JitEntry {
  enum State { ABSENT = 0, PENDING, READY, FAILED } state = ABSENT;
  hipModule_t mod = nullptr;
  hipFunction_t fn = nullptr;
  std::chrono::steady_clock::time_point checked{}, started{};
};
std::mutex g_jit_mu;   // held across the compile: one hiprtc call at a time
std::map<std::string, JitEntry> g_jit_cache;   // key: the generated source (meta + variant)

int jit_mode() {   // 0 = off, 1 = prepared kernels only (default), 2 = compile on first use
  static const int m = [] {
    const char* e = getenv("DBHIP_FAGG_JIT");
    if (!e) return 1;
    if (!strcmp(e, "sync")) return 2;
    return atoi(e) == 0 ? 0 : 1;
  }();
  return m;
}

enum { JIT_LOOKUP = 0, JIT_COMPILE = 1, JIT_BACKGROUND = 2 };
// -> the specialised kernel of (shape, variant), or nullptr. how = JIT_LOOKUP: only what the in-process cache or the on-disk
// cache holds; JIT_COMPILE: compile now if needed (PREPARE; blocks ~0.5 s); JIT_BACKGROUND: if it is nowhere yet, start the
// compile in a detached helper and return nullptr now — *pending says so, a later call finds the result on disk.
// the binary image of a query shape: FaArgs with everything that varies between calls of one shape zeroed (the fields jit_meta prints
// as null / 0)
void fa_shape(const FaArgs& A, FaArgs& K) {
  K = A;
  for (int i = 0; i < EX_MAX_INPUTS; ++i) { K.P.in_data[i] = nullptr; K.P.in_valid[i] = nullptr; K.P.in_voff[i] = 0; }
  K.P.err_words = nullptr; K.P.err_count = nullptr;
  for (int k = 0; k < FA_KW; ++k) { K.key[k].data = nullptr; K.key[k].validity = nullptr; K.key[k].voff = 0; K.key[k].buffers = nullptr; }
  K.filter_bits = nullptr; K.filter_off = 0; K.n = 0; K.partial_rows = nullptr; K.ctrl = nullptr; K.blocks = nullptr; K.wgs_per_block = 0;
}
hipFunction_t jit_kernel(const FaArgs& A, int slots, bool general, int nw, int how, bool* pending = nullptr, bool multi = false) {
  if (pending) *pending = false;
  const int mode = jit_mode();
  if (mode == 0) return nullptr;
  if (mode == 2) how = JIT_COMPILE;
  // in-process key: the binary image of the query shape (FaArgs with everything that varies between calls of one shape
  // zeroed — the fields jit_meta prints as null / 0), the variant, the experiment knobs and the device the module is loaded
  // on. (Generating the metadata TEXT per call to look the kernel up cost 0.25 ms of host time per launch.)
  FaArgs K;
  fa_shape(A, K);
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::string key((const char*)&K, sizeof(K));
  key += "|" + std::to_string(slots) + "|" + std::to_string((int)general) + "|" + std::to_string(nw) + "|" + std::to_string(dev) + "|" + (multi ? "m|" : "|");
  key += exp_env("DBHIP_FAGG_JIT_DEFS") ? exp_env("DBHIP_FAGG_JIT_DEFS") : "";
  const bool trace = getenv("DBHIP_TRACE") != nullptr;
  std::lock_guard<std::mutex> lock(g_jit_mu);
  auto it = g_jit_cache.find(key);
  const auto now = std::chrono::steady_clock::now();
  if (it != g_jit_cache.end()) {
    JitEntry& e = it->second;
    if (e.state == JitEntry::READY) return e.fn;
    if (e.state == JitEntry::FAILED) return nullptr;
    // ABSENT (looked up, never compiled) / PENDING (a background compile runs): look at the disk again, not more often than
    // every 20 ms; a blocking request compiles now whatever the state
    const bool will_spawn = how == JIT_BACKGROUND && e.state == JitEntry::ABSENT;
    if (how != JIT_COMPILE && !will_spawn && now - e.checked < std::chrono::milliseconds(20)) { if (pending) *pending = e.state == JitEntry::PENDING; return nullptr; }
  }
  JitEntry& e = g_jit_cache[key];
  e.checked = now;
  const std::string meta = jit_meta(A, multi), tail = jit_tail(slots, general, nw);
  std::vector<char> code;
  std::string log;
  bool have = false;
  {
    const std::string cached = jit_cache_path(meta, tail, jit_arch(), exp_env("DBHIP_FAGG_JIT_DEFS") ? exp_env("DBHIP_FAGG_JIT_DEFS") : "");
    have = !cached.empty() && read_cached_code(cached, &code);
  }
  const bool from_disk = have;
  if (!have && how == JIT_COMPILE) {
    have = jit_compile(meta, tail, &code, &log);
    if (!have) {
      e.state = JitEntry::FAILED;
      if (trace) fprintf(stderr, "[dbhip] fagg jit: hiprtc failed; the interpreting kernel is used.\n%s\n", log.c_str());
      return nullptr;
    }
  }
  if (!have && how == JIT_BACKGROUND && e.state != JitEntry::PENDING) {
    code.clear();
    (void)jit_compile(meta, tail, &code, &log, true);
    if (log == "pending") { e.state = JitEntry::PENDING; e.started = now; if (trace) fprintf(stderr, "[dbhip] fagg jit: compiling in the background\n"); }
    else { e.state = JitEntry::FAILED; if (trace) fprintf(stderr, "[dbhip] fagg jit: no background compile: %s\n", log.c_str()); }
  }
  if (!have) {
    if (e.state == JitEntry::PENDING && now - e.started > std::chrono::seconds(120)) e.state = JitEntry::FAILED;   // the helper died
    if (pending) *pending = e.state == JitEntry::PENDING;
    return nullptr;
  }
  if (const char* dump = exp_env("DBHIP_FAGG_JIT_DUMP")) {   // code object (and the generated metadata) for offline disassembly
    static int seq = 0;
    char path[512];
    snprintf(path, sizeof(path), "%s.%d.co", dump, seq);
    if (FILE* f = fopen(path, "wb")) { fwrite(code.data(), 1, code.size(), f); fclose(f); }
    snprintf(path, sizeof(path), "%s.%d.meta", dump, seq++);
    if (FILE* f = fopen(path, "wb")) { fwrite(meta.data(), 1, meta.size(), f); fwrite(tail.data(), 1, tail.size(), f); fclose(f); }
  }
  if (hipModuleLoadData(&e.mod, code.data()) == hipSuccess && hipModuleGetFunction(&e.fn, e.mod, "fagg_jit") == hipSuccess) {
    e.state = JitEntry::READY;
    if (trace) fprintf(stderr, "[dbhip] fagg jit: specialised kernel ready (%zu bytes of code)\n", code.size());
    return e.fn;
  }
  if (from_disk) {
    // a cached file that does not load (another compiler's, corrupt): a cache miss, not a verdict on the shape — drop it so
    // that the next PREPARE / background compile writes a fresh one
    const std::string cached = jit_cache_path(meta, tail, jit_arch(), exp_env("DBHIP_FAGG_JIT_DEFS") ? exp_env("DBHIP_FAGG_JIT_DEFS") : "");
    if (!cached.empty()) (void)unlink(cached.c_str());
    e.state = JitEntry::ABSENT;
    if (trace) fprintf(stderr, "[dbhip] fagg jit: the cached code object did not load; removed, the shape will be compiled again\n");
    return nullptr;
  }
  e.state = JitEntry::FAILED;
  if (trace) fprintf(stderr, "[dbhip] fagg jit: the code object did not load; the interpreting kernel is used\n");
  return nullptr;
}

bool fa_arg_type_ok(int t) { return type_class(t) >= 0 || t == DBHIP_T_BOOL || t == DBHIP_T_DEC128; }

}  // namespace

// Can this table's layout go through the fused kernel at all? (<= 4 key words incl. validity, <= 8 aggregates, <= 12
// state words laid out back to back)
bool dbhip_fagg_layout_ok_internal(const GbLayout& L) {
  if (L.nkeys > FA_KW || L.nkey_words > FA_KW || L.naggs > FA_MAXA || L.naggs < 1) return false;
  for (int k = 0; k < L.nkeys; ++k)
    if (L.key_type[k] == DBHIP_T_DEC256) return false;   // four-word keys: row path
  int words = 0;
  for (int a = 0; a < L.naggs; ++a) {
    if (L.agg_off[a] != L.agg_off[0] + words) return false;
    if ((L.agg_kind[a] == DBHIP_AGG_MIN || L.agg_kind[a] == DBHIP_AGG_MAX) && L.agg_words[a] >= 3) return false;   // Decimal128 / Decimal256 / String min / max: row path
    if (L.agg_type[a] == DBHIP_T_DEC256 && L.agg_kind[a] != DBHIP_AGG_COUNT) return false;                           // Decimal256 sum: row path
    words += L.agg_words[a];
  }
  return words <= FA_MAXW && L.hash_word == L.nkey_words && L.agg_off[0] == L.hash_word + 1;
}

// launches by kind since the library was loaded (tests and benches tell which kernel a call went through)
static std::atomic<uint64_t> g_fa_jit_launches{0}, g_fa_interp_launches{0}, g_fa_pending_refusals{0}, g_fa_multi_blocks{0};
extern "C" int32_t dbhip_fagg_stats(uint64_t* out3_host) {
  DBHIP_REQUIRE(out3_host, "dbhip_fagg_stats: NULL argument");
  out3_host[0] = g_fa_jit_launches.load(); out3_host[1] = g_fa_interp_launches.load(); out3_host[2] = g_fa_pending_refusals.load();
  return DBHIP_OK;
}

static thread_local bool t_prepare_only = false;
static thread_local bool t_jit_only = false;      // dbhip_fagg_add_columns_internal: never fall back to the interpreting kernel
static thread_local bool t_jit_pending = false;   // ... and whether the refusal was "the compile is still running"
static thread_local bool t_jit_may_compile = true;   // ... and whether a missing kernel may be compiled in the background

// Host side of the fused launch: compiles the program (roots = filter + one per aggregate argument), derives the per-word
// metadata of the layout's states and copies the key columns. Touches no device: the offline compile check
// (dbhip_jit_offline, tools/jit_offline.py) goes through the same function.
static int32_t fa_build_args(const GbLayout& L, const dbhip_col* keys, const dbhip_agg_program* prog, FaArgs& A, bool* out_general,
                             int* out_nwords, bool* out_may_raise) {
  // ---- roots: filter + one per aggregate argument ----
  ExRoot roots[EX_MAX_ROOTS + 1];
  int n_roots = 0, filter_root = -1;
  int root_of_agg[FA_MAXA];
  if (prog->filter_reg >= 0) { roots[n_roots].reg = prog->filter_reg; filter_root = n_roots++; }
  for (int a = 0; a < L.naggs; ++a) {
    root_of_agg[a] = -1;
    const int32_t r = prog->arg_regs[a];
    if (r == DBHIP_ARG_NONE) {
      if (L.agg_kind[a] != DBHIP_AGG_COUNT) { set_error("dbhip_groupby_add_block_program: aggregate %d needs an argument", a); return DBHIP_ERR_INVALID; }
      continue;
    }
    // the same register may feed several aggregates: one root each (the compiler maps equal registers to equal slots)
    if (n_roots >= EX_MAX_ROOTS) { set_error("dbhip_groupby_add_block_program: more than %d program results", EX_MAX_ROOTS); return DBHIP_ERR_UNSUPPORTED; }
    roots[n_roots].reg = r;
    root_of_agg[a] = n_roots++;
  }
  bool may_raise = false, any_nullable = false;
  if (n_roots == 0) {  // count(*) only: a harmless root keeps the compiler's contract (an input column, if any)
    set_error("dbhip_groupby_add_block_program: nothing to evaluate (count(*) only): use dbhip_groupby_add_block_filtered");
    return DBHIP_ERR_UNSUPPORTED;
  }
  int32_t rc = dbhip_expr_compile_internal(prog->prog, prog->n_ins, prog->inputs, prog->n_inputs, roots, n_roots, filter_root, &A.P,
                                           &may_raise, &any_nullable);
  if (rc) return rc;
  // ---- aggregates <-> roots -> packed per-word metadata ----
  A.naggs = L.naggs;
  int nwords = 0;
  bool general = false;
  for (int a = 0; a < L.naggs; ++a) {
    int slot = 0, wide = 0, sgn = 0, enc_type = L.agg_type[a];
    uint32_t dep = 0;
    const int fw = L.agg_flag[a];
    const int base_words = L.agg_words[a] - (fw ? 1 : 0);
    if (root_of_agg[a] >= 0) {
      const ExRoot& R = roots[root_of_agg[a]];
      if (L.agg_kind[a] != DBHIP_AGG_COUNT) {
        if (R.type != L.agg_type[a] || !fa_arg_type_ok(R.type)) {
          set_error("dbhip_groupby_add_block_program: aggregate %d expects an argument of type %d, the program yields type %d", a, L.agg_type[a], R.type);
          return DBHIP_ERR_INVALID;
        }
      }
      slot = R.slot; wide = R.wide; dep = R.dep;
      sgn = R.type == DBHIP_T_DEC128 || type_class(R.type) == CLS_SIGNED;
      if (R.type == DBHIP_T_F32) enc_type = DBHIP_T_F64;  // the register image of an f32 is its f64 value: same order-preserving key
    }
    auto put = [&](int op, int comp, int carry_in) {
      A.wm[nwords++] = (uint32_t)op | ((uint32_t)comp << 3) | ((uint32_t)slot << 6) | ((uint32_t)wide << 12) | ((uint32_t)sgn << 13) |
                       ((uint32_t)carry_in << 14) | (1u << 15) | ((dep & 255u) << 16) | (((uint32_t)enc_type & 31u) << 24);
    };
    switch (L.agg_kind[a]) {
      case DBHIP_AGG_COUNT: put(W_ADD1, C_FLAG, 0); break;
      case DBHIP_AGG_SUM:
        if (base_words == 3) { put(W_ADD3, C_LO, 0); put(W_CONT, C_HI, 1); put(W_CONT, C_EXT, 1); }
        else if (L.agg_type[a] == DBHIP_T_F32 || L.agg_type[a] == DBHIP_T_F64) { put(W_FADD, C_LO, 0); general = true; }
        else put(W_ADD1, C_LO, 0);
        if (fw) put(W_OR, C_FLAG, 0);
        break;
      case DBHIP_AGG_MIN: case DBHIP_AGG_MAX:
        if (L.agg_words[a] >= 3) { set_error("fused aggregation: min / max over Decimal128 / Decimal256 / String stays on the row path"); return DBHIP_ERR_UNSUPPORTED; }
        if (L.agg_kind[a] == DBHIP_AGG_MIN) put(W_MIN, C_ENC, 0); else put(W_MAX, C_ENC, 0);
        put(W_CONT, C_FLAG, 0); general = true; break;
    }
  }
  A.nwords = nwords;
  A.state_off = L.agg_off[0];
  // ---- keys ----
  A.nkeys = L.nkeys; A.nkey_words = L.nkey_words; A.validity_word = L.validity_word; A.hash_word = L.hash_word; A.W = L.W;
  for (int k = 0; k < L.nkeys; ++k) {
    if (keys[k].type != L.key_type[k]) { set_error("dbhip_groupby_add_block_program: key %d has type %d, table expects %d", k, keys[k].type, L.key_type[k]); return DBHIP_ERR_INVALID; }
    if (keys[k].validity && !L.key_nullable[k]) { set_error("dbhip_groupby_add_block_program: key %d carries validity but was declared NOT NULL", k); return DBHIP_ERR_INVALID; }
    GbCol& c = A.key[k];
    c.data = keys[k].data; c.validity = keys[k].validity; c.voff = keys[k].validity_offset; c.buffers = keys[k].buffers;
    c.type = keys[k].type; c.is_scalar = keys[k].is_scalar;
    A.key_type[k] = L.key_type[k]; A.key_off[k] = L.key_off[k]; A.key_words[k] = L.key_words[k];
    A.key_has_valid[k] = keys[k].validity != nullptr;
  }
  *out_general = general; *out_nwords = nwords; *out_may_raise = may_raise;
  return DBHIP_OK;
}


// One launch of the fused kernel over a block: the specialised kernel of (shape, SLOTS) when the in-process / on-disk cache has it —
// an un-PREPAREd program is interpreted now and compiled in the background for the calls to come — else the interpreting kernel.
// DBHIP_ERR_UNSUPPORTED (t_jit_only, plain add_block's use of this kernel: only the specialised form beats the LDS path).
static int32_t fa_launch(const FaArgs& A, int slots, bool general, int nwords, int grid, size_t lds, hipStream_t s) {
  bool jit_pending = false;
  hipFunction_t jf = jit_kernel(A, slots, general, nwords, (t_jit_only && !t_jit_may_compile) ? JIT_LOOKUP : JIT_BACKGROUND, &jit_pending);
  if (!jf && t_jit_only) {
    t_jit_pending = jit_pending;
    if (jit_pending) ++g_fa_pending_refusals;
    set_error("dbhip_fagg: the specialised kernel of this shape is %s", jit_pending ? "being compiled in the background" : "not available");
    return DBHIP_ERR_UNSUPPORTED;
  }
  if (jf) {
    size_t asz = sizeof(A);
    void* extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, (void*)&A, HIP_LAUNCH_PARAM_BUFFER_SIZE, &asz, HIP_LAUNCH_PARAM_END};
    // (no dynamic LDS: the register file is in VGPRs, and since round 6 the end of the kernel folds in registers too)
    DBHIP_CHECK(hipModuleLaunchKernel(jf, grid, 1, 1, 256, 1, 1, 0, s, nullptr, extra));
    ++g_fa_jit_launches;
    return DBHIP_OK;
  }
  ++g_fa_interp_launches;
#define FA_LAUNCH(SL, GEN)                                                                                     \
  do {                                                                                                         \
    if (nwords <= 4) hipLaunchKernelGGL((fagg_kernel<SL, GEN, 4>), dim3(grid), dim3(256), lds, s, A);           \
    else hipLaunchKernelGGL((fagg_kernel<SL, GEN, FA_MAXW>), dim3(grid), dim3(256), lds, s, A);                 \
  } while (0)
  // (interpreted) a program with a rounding decimal multiply / divide: the instantiations that carry the long divisions
  bool has_div = false;
  for (int i = 0; i < A.P.n_ins; ++i) has_div |= A.P.ins[i].op == EX_DEC && dec_op_needs_division(A.P.dec[A.P.ins[i].dec_idx]);
  if (has_div && slots == 4) hipLaunchKernelGGL((fagg_kernel<4, true, FA_MAXW, true>), dim3(grid), dim3(256), lds, s, A);
  else if (has_div) hipLaunchKernelGGL((fagg_kernel<8, true, FA_MAXW, true>), dim3(grid), dim3(256), lds, s, A);
  else if (slots == 4 && !general) FA_LAUNCH(4, false);
  else if (slots == 4) FA_LAUNCH(4, true);
  else if (!general) FA_LAUNCH(8, false);
  else FA_LAUNCH(8, true);
#undef FA_LAUNCH
  return DBHIP_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// PIPELINED mode (dbhip_groupby_set_pipelined / dbhip_groupby_checkpoint; round 6).
// The reference's pipeline hands TransformPartialAggregate one DataBlock of <= 65,536 rows at a time (max_block_size,
// settings_default.rs:142-148; transform_aggregate_partial.rs:262-270). At that size the fused kernel is a few microseconds of
// work, and a call that reads the kernel's control block back and waits for the merge is all round trip. In pipelined mode a block
// costs ONE kernel launch and no synchronisation:
//   * the partial rows of successive blocks are APPENDED to a buffer the table owns (cursor, give-up flags and the row-error count
//     live on the device and are sticky);
//   * when the buffer could overflow (the host keeps an upper bound: workgroups x 8 slots per block) the merge of the
//     whole window is queued behind the kernels — seal (row errors -> flag), probe / accumulate / retry with the row count and the
//     abort flags still on the device, commit (cursor back to 0, the window's blocks counted as committed unless a flag is up);
//     the commit kernel writes the table's group count into mapped host memory, from which the host bounds the table's fill without
//     waiting (it synchronises only when that bound says the table might have to grow);
//   * dbhip_groupby_checkpoint drains the stream and reports: DBHIP_OK (every block merged), or the error a synchronous call would
//     have returned plus the number of blocks that WERE merged — windows commit in order and a window with a raised flag, and every
//     window behind it, merges nothing, so the blocks from that index on can be handed to the operator-at-a-time path exactly like a
//     block that the synchronous call gave back. A "more than 4 groups in one workgroup" give-up of the 4-slot kernel is replayed
//     here with the 8-slot kernel first (the library keeps the launch arguments of unconfirmed blocks).
// The caller keeps the blocks' buffers alive until the checkpoint (they are inputs of queued kernels), which is also what makes the
// replay possible. One stream per pipelined table (the reference's partial tables are per pipeline thread as well).
// ---------------------------------------------------------------------------------------------------------------------
namespace {
constexpr int FA_PIPE_WINDOW = 2048;  // blocks per window at most (one merge; what one raised flag gives back). r06: 128 = one merge chain (five
                                      // dependent small kernels, ~30 us) behind EVERY 128-block launch; a window also closes when its launches could
                                      // fill the row buffer (16 launches of the full grid)
constexpr int FA_PIPE_BATCH = 512;    // blocks per multi-block launch (r06 sweeps at 65,536-row blocks, one thread: 32 -> 25, 64 -> 33, 128 -> 40-43 G rows/s;
                                      // a launch costs ~40 us whatever it holds: 8 Mi rows per launch stream at 54 G rows/s, 32 Mi at 70+)
constexpr int FA_PIPE_BATCH_MAX = 512; // (DBHIP_FAGG_PIPE_BATCH sweeps 2..512: experiments build)
constexpr size_t FA_PIPE_COPY_BYTES = 16384;    // a pinned -> device hipMemcpyAsync above 16 KB blocks the calling thread on this stack (fagg_device.h,
                                                // tools/probes/h2d_small_copy.hip): a launch's packed block table is uploaded in pieces of this size
constexpr size_t FA_PIPE_TABLE_BYTES = 65536;   // one launch's packed block table (512 blocks of 16 words; wide shapes get fewer blocks per launch)
constexpr int FA_PIPE_RING = 8;
constexpr int64_t FA_PIPE_BIG = 8 << 20;          // a block of this many rows is a launch of its own
constexpr int64_t FA_PIPE_BATCH_ROWS = 32 << 20;  // rows after which a batch goes without waiting for more blocks
// A query shape as the pipeline keeps it: the compiled launch arguments with every per-block pointer null, and the bytes the shape
// was recognised by (`sig`: program, column types / scalar-ness / validity-ness, argument registers). A block is then 328 bytes: its
// pointers and its shape's index — the per-call host work of a pipelined table is one signature compare and one FaBlock fill
// (compiling the program and copying the 4 KB argument block per call was 3 us of a 65,536-row call, 21 G rows/s per thread at most).
struct FaShape { FaArgs A; bool general; int nwords; size_t lds; std::string sig; };
struct FaPending { int shape; FaBlock b; };
struct FaMerge { uint64_t seq; int64_t rows_ub; int64_t blocks; };
struct FaPipe {
  bool on = false, bound = false, slots8 = false;
  bool given_up = false;             // a checkpoint returned DBHIP_ERR_CAPACITY: plain add_block calls stop queueing (the keys are not for this kernel)
  hipStream_t stream = nullptr;
  uint64_t* rows = nullptr;          // device [cap_rows][W]
  int64_t cap_rows = 0;
  uint64_t* ctrl = nullptr;          // device, 64 B: [0] rows appended since the last merge [1] flags (sticky) [2] row errors (sticky) [3] blocks committed since the last checkpoint
  uint64_t* status_host = nullptr;   // mapped pinned host memory: [0] seq of the last merge that finished, [1] groups in the table then (bit 63: a flag was up)
  uint64_t* status_dev = nullptr;
  int64_t rows_ub = 0, window_blocks = 0;   // of the window that is being filled
  uint64_t seq = 0;
  std::deque<FaMerge> merges;        // queued, not yet seen finished
  int64_t count_seen = 0;            // groups after the last merge seen finished
  std::deque<FaPending> retained;    // launch arguments of the blocks [retained_base, submitted) since the last checkpoint
  int64_t retained_base = 0, submitted = 0;
  // multi-block launches: blocks below FA_PIPE_BIG rows wait here until FA_PIPE_BATCH of them (or FA_PIPE_BATCH_ROWS rows) can go
  // in ONE launch; their pointer tables travel through a ring of pinned staging / device buffers
  std::vector<FaPending> batch;
  int64_t batch_rows = 0;
  std::vector<FaShape> shapes;       // shapes of the blocks since the last checkpoint (normally one)
  uint64_t* tab_host[FA_PIPE_RING] = {};
  uint64_t* tab_dev[FA_PIPE_RING] = {};
  hipEvent_t tab_ev[FA_PIPE_RING] = {};
  bool tab_used[FA_PIPE_RING] = {};
  int tab_next = 0;
};

__global__ void fa_pipe_seal_kernel(uint64_t* ctrl) {
  if (ctrl[2]) ctrl[1] |= 4;
}
__global__ void fa_pipe_commit_kernel(uint64_t* ctrl, const uint64_t* gctrl, uint64_t* status, uint64_t seq, uint64_t blocks) {
  if (gctrl[1]) ctrl[1] |= 8;   // the merge ran out of table (cannot happen: the host reserved room for every row of the window)
  const bool dirty = (ctrl[1] & 15) != 0;
  if (!dirty) ctrl[3] += blocks;
  ctrl[0] = 0;                  // rows of a window that did not commit are dropped with it
  __hip_atomic_store(&status[1], gctrl[0] | (dirty ? (1ULL << 63) : 0ULL), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __hip_atomic_store(&status[0], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// what the mapped status words say right now: merges that finished leave the queue; the launch arguments of blocks that are known
// to be merged are dropped
void fa_pipe_poll(FaPipe* pp) {
  uint64_t seq0, cnt, seq1;
  do {
    seq0 = __atomic_load_n(&pp->status_host[0], __ATOMIC_ACQUIRE);
    cnt = __atomic_load_n(&pp->status_host[1], __ATOMIC_ACQUIRE);
    seq1 = __atomic_load_n(&pp->status_host[0], __ATOMIC_ACQUIRE);
  } while (seq0 != seq1);
  const bool dirty = (cnt >> 63) != 0;
  while (!pp->merges.empty() && pp->merges.front().seq <= seq0) {
    if (!dirty) {
      int64_t drop = pp->merges.front().blocks;
      while (drop-- > 0 && !pp->retained.empty()) { pp->retained.pop_front(); ++pp->retained_base; }
    }
    pp->merges.pop_front();
    pp->count_seen = (int64_t)(cnt & ~(1ULL << 63));
  }
}

int32_t fa_pipe_queue_merge(dbhip_groupby* g, FaPipe* pp) {
  if (pp->window_blocks == 0) return DBHIP_OK;
  hipStream_t s = pp->stream;
  fa_pipe_poll(pp);
  int64_t ub = pp->count_seen + pp->rows_ub;
  for (const FaMerge& m : pp->merges) ub += m.rows_ub;
  if (ub * 135 > dbhip_groupby_capacity_internal(g) * 100) {
    // the bound says the table might fill: look (drains the stream) and grow if it really has to
    const int32_t rc = dbhip_groupby_ensure_room_internal(g, pp->rows_ub, s);
    if (rc) return rc;
    fa_pipe_poll(pp);
    pp->merges.clear();
    pp->count_seen = dbhip_groupby_count_internal(g);
  }
  hipLaunchKernelGGL(fa_pipe_seal_kernel, dim3(1), dim3(1), 0, s, pp->ctrl);
  int32_t rc = dbhip_groupby_merge_rows_deferred_internal(g, pp->rows, pp->rows_ub, &pp->ctrl[0], &pp->ctrl[1], s);
  if (rc) return rc;
  ++pp->seq;
  hipLaunchKernelGGL(fa_pipe_commit_kernel, dim3(1), dim3(1), 0, s, pp->ctrl, dbhip_groupby_ctrl_internal(g), pp->status_dev, pp->seq,
                     (uint64_t)pp->window_blocks);
  DBHIP_LAUNCH_CHECK();
  pp->merges.push_back({pp->seq, pp->rows_ub, pp->window_blocks});
  pp->rows_ub = 0;
  pp->window_blocks = 0;
  return DBHIP_OK;
}

// the launch arguments of one block: its shape with the block's pointers put back
void fa_pipe_args(const FaPipe* pp, const FaPending& P, FaArgs& A) {
  A = pp->shapes[(size_t)P.shape].A;
  for (int c = 0; c < EX_MAX_INPUTS; ++c) { A.P.in_data[c] = P.b.in_data[c]; A.P.in_valid[c] = P.b.in_valid[c]; A.P.in_voff[c] = P.b.in_voff[c]; }
  for (int k = 0; k < FA_KW; ++k) { A.key[k].data = P.b.key_data[k]; A.key[k].validity = P.b.key_valid[k]; A.key[k].voff = P.b.key_voff[k]; }
  A.filter_bits = P.b.filter_bits; A.filter_off = P.b.filter_off; A.n = P.b.n;
}
// one block, one launch
int32_t fa_pipe_submit(dbhip_groupby* g, FaPipe* pp, const FaPending& P, hipStream_t s) {
  const FaShape& S = pp->shapes[(size_t)P.shape];
  FaArgs A;
  fa_pipe_args(pp, P, A);
  const int64_t nchunks = ceil_div(A.n, 64 * FA_ROWS);
  const int grid = (int)(ceil_div(nchunks, 4) < 512 ? ceil_div(nchunks, 4) : 512);
  const int64_t n_max = (int64_t)grid * FA_MAX_SLOTS;   // one partial row per (workgroup, slot)
  if (n_max > pp->cap_rows) { set_error("dbhip_groupby_add_block_program: grid too large for the pipeline's row buffer"); return DBHIP_ERR_INVALID; }
  // a window closes when the row buffer could overflow, and after FA_PIPE_WINDOW blocks at the latest: that bounds what one raised
  // flag gives back to the caller (and the launch arguments kept for a replay) while one merge still serves dozens of launches
  if (pp->rows_ub + n_max > pp->cap_rows || pp->window_blocks >= FA_PIPE_WINDOW) {
    const int32_t rc = fa_pipe_queue_merge(g, pp);
    if (rc) return rc;
  }
  A.ctrl = pp->ctrl;
  A.partial_rows = pp->rows;
  A.P.err_words = nullptr;
  A.P.err_count = (unsigned long long*)&pp->ctrl[2];
  const int32_t rc = fa_launch(A, (pp->slots8 || pp->count_seen > 4) ? 8 : 4, S.general, S.nwords, grid, S.lds, s);
  if (rc) return rc;
  DBHIP_LAUNCH_CHECK();
  pp->rows_ub += n_max;
  ++pp->window_blocks;
  ++pp->submitted;
  pp->retained.push_back(P);
  return DBHIP_OK;
}

// Launches the queued blocks together (see FaBlock). Needs the FA_MULTI specialisation of the shape: while that is being compiled
// (or cannot be), the blocks go one launch each.
// one block's pointers in the shape's packed layout (fagg_device.h: fa_blk_in_off / fa_blk_key_off / fa_blk_filter_off)
void fa_pack_block(const FaArgs& M, const FaBlock& b, uint64_t* w) {
  w[0] = (uint64_t)b.n;
  for (int ci = 0; ci < M.P.n_inputs; ++ci) {
    const int o = fa_blk_in_off(M, ci);
    w[o] = (uint64_t)b.in_data[ci];
    if (M.P.in_has_valid[ci]) { w[o + 1] = (uint64_t)b.in_valid[ci]; w[o + 2] = (uint64_t)b.in_voff[ci]; }
  }
  for (int q = 0; q < M.nkeys; ++q) {
    const int o = fa_blk_key_off(M, q);
    w[o] = (uint64_t)b.key_data[q];
    if (M.key_has_valid[q]) { w[o + 1] = (uint64_t)b.key_valid[q]; w[o + 2] = (uint64_t)b.key_voff[q]; }
  }
  if (M.has_filter) { const int o = fa_blk_filter_off(M); w[o] = (uint64_t)b.filter_bits; w[o + 1] = (uint64_t)b.filter_off; }
}

int32_t fa_pipe_flush_batch(dbhip_groupby* g, FaPipe* pp) {
  if (pp->batch.empty()) return DBHIP_OK;
  std::vector<FaPending> blocks;
  blocks.swap(pp->batch);
  pp->batch_rows = 0;
  hipStream_t s = pp->stream;
  const int nb = (int)blocks.size();
  const int slots = (pp->slots8 || pp->count_seen > 4) ? 8 : 4;
  const FaShape& S = pp->shapes[(size_t)blocks[0].shape];
  hipFunction_t jf = nb > 1 ? jit_kernel(S.A, slots, S.general, S.nwords, JIT_BACKGROUND, nullptr, true) : nullptr;
  if (!jf) {
    for (const FaPending& P : blocks) { const int32_t rc = fa_pipe_submit(g, pp, P, s); if (rc) return rc; }
    return DBHIP_OK;
  }
  const int64_t rpw = 64 * (int64_t)jit_rows(S.A);
  int64_t max_chunks = 1;
  for (const FaPending& P : blocks) { const int64_t c = ceil_div(P.b.n, rpw); max_chunks = c > max_chunks ? c : max_chunks; }
  int wpb = (int)(ceil_div(max_chunks, 4) < 512 / nb ? ceil_div(max_chunks, 4) : 512 / nb);
  if (wpb < 1) wpb = 1;
  const int grid = wpb * nb;
  const int64_t n_max = (int64_t)grid * FA_MAX_SLOTS;
  if (pp->rows_ub + n_max > pp->cap_rows || pp->window_blocks >= FA_PIPE_WINDOW) {
    const int32_t rc = fa_pipe_queue_merge(g, pp);
    if (rc) return rc;
  }
  const int slot = pp->tab_next;
  pp->tab_next = (pp->tab_next + 1) % FA_PIPE_RING;
  if (pp->tab_used[slot]) DBHIP_CHECK(hipEventSynchronize(pp->tab_ev[slot]));   // the copy that last read this staging buffer has run
  uint64_t* T = pp->tab_host[slot];
  const int bw = fa_blk_words(S.A);
  for (int i = 0; i < nb; ++i) fa_pack_block(S.A, blocks[i].b, T + (size_t)i * bw);
  for (size_t at = 0, all = (size_t)nb * bw * 8; at < all; at += FA_PIPE_COPY_BYTES)
    DBHIP_CHECK(hipMemcpyAsync((uint8_t*)pp->tab_dev[slot] + at, (const uint8_t*)T + at, all - at < FA_PIPE_COPY_BYTES ? all - at : FA_PIPE_COPY_BYTES,
                               hipMemcpyHostToDevice, s));
  DBHIP_CHECK(hipEventRecord(pp->tab_ev[slot], s));
  pp->tab_used[slot] = true;
  FaArgs A = S.A;
  A.ctrl = pp->ctrl; A.partial_rows = pp->rows;
  A.P.err_words = nullptr; A.P.err_count = (unsigned long long*)&pp->ctrl[2];
  A.blocks = pp->tab_dev[slot]; A.wgs_per_block = wpb;
  size_t asz = sizeof(A);
  void* extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, (void*)&A, HIP_LAUNCH_PARAM_BUFFER_SIZE, &asz, HIP_LAUNCH_PARAM_END};
  DBHIP_CHECK(hipModuleLaunchKernel(jf, grid, 1, 1, 256, 1, 1, 0, s, nullptr, extra));
  g_fa_jit_launches += 1;
  g_fa_multi_blocks += (uint64_t)nb;
  pp->rows_ub += n_max;
  pp->window_blocks += nb;
  pp->submitted += nb;
  for (const FaPending& P : blocks) pp->retained.push_back(P);
  return DBHIP_OK;
}

// a block enters a pipelined table: large ones are launched at once, small ones wait for company
int32_t fa_pipe_enqueue(dbhip_groupby* g, FaPipe* pp, const FaPending& P, hipStream_t s) {
  static const bool no_batch = exp_env("DBHIP_FAGG_PIPE_BATCH") && atoi(exp_env("DBHIP_FAGG_PIPE_BATCH")) == 0;
  if (P.b.n >= FA_PIPE_BIG || no_batch || jit_mode() == 0) {
    const int32_t rc = fa_pipe_flush_batch(g, pp);   // (blocks stay in call order)
    return rc ? rc : fa_pipe_submit(g, pp, P, s);
  }
  if (!pp->batch.empty() && pp->batch[0].shape != P.shape) {
    const int32_t rc = fa_pipe_flush_batch(g, pp);   // another query shape: its own launch
    if (rc) return rc;
  }
  pp->batch.push_back(P);
  pp->batch_rows += P.b.n;
  static const int batch_n = [] { const char* e = exp_env("DBHIP_FAGG_PIPE_BATCH"); const int v = e ? atoi(e) : 0; return v >= 2 && v <= FA_PIPE_BATCH_MAX ? v : FA_PIPE_BATCH; }();
  const int fit = (int)(FA_PIPE_TABLE_BYTES / ((size_t)fa_blk_words(pp->shapes[(size_t)P.shape].A) * 8));   // >= 210: a block is at most 39 words
  if ((int)pp->batch.size() >= (batch_n < fit ? batch_n : fit) || pp->batch_rows >= FA_PIPE_BATCH_ROWS) return fa_pipe_flush_batch(g, pp);
  return DBHIP_OK;
}

void fa_pipe_forget(FaPipe* pp) {
  pp->batch.clear(); pp->batch_rows = 0; pp->shapes.clear();
  pp->retained.clear(); pp->retained_base = 0; pp->submitted = 0; pp->merges.clear(); pp->rows_ub = 0; pp->window_blocks = 0;
}

int32_t fa_pipe_checkpoint(dbhip_groupby* g, FaPipe* pp, int64_t* out_committed, bool may_replay) {
  if (out_committed) *out_committed = 0;
  if (!pp->bound || (pp->submitted == 0 && pp->window_blocks == 0 && pp->batch.empty())) return DBHIP_OK;
  hipStream_t s = pp->stream;
  int32_t rc = fa_pipe_flush_batch(g, pp);
  if (rc) return rc;
  if ((rc = fa_pipe_queue_merge(g, pp))) return rc;
  uint64_t* h = pinned_words(2);
  if (!h) return DBHIP_ERR_HIP;
  DBHIP_CHECK(hipMemcpyAsync(h, pp->ctrl, 32, hipMemcpyDeviceToHost, s));
  DBHIP_CHECK(hipMemcpyAsync(h + 4, dbhip_groupby_ctrl_internal(g), 8, hipMemcpyDeviceToHost, s));
  DBHIP_CHECK(hipStreamSynchronize(s));
  const uint64_t flags = h[1], errs = h[2];
  const int64_t committed = (int64_t)h[3];
  dbhip_groupby_set_count_internal(g, (int64_t)h[4]);
  pp->merges.clear();
  pp->count_seen = (int64_t)h[4];
  if (!(flags & 15)) {
    fa_pipe_forget(pp);
    DBHIP_CHECK(hipMemsetAsync(pp->ctrl, 0, 64, s));
    if (out_committed) *out_committed = committed;
    return DBHIP_OK;
  }
  if ((flags & 15) == 1 && !pp->slots8 && may_replay) {
    // the 4-slot kernel met a fifth group inside one workgroup: the blocks that did not commit go through the 8-slot kernel,
    // as the synchronous call's second pass does
    pp->slots8 = true;
    std::deque<FaPending> again;
    const int64_t skip = committed - pp->retained_base;
    for (int64_t i = skip < 0 ? 0 : skip; i < (int64_t)pp->retained.size(); ++i) again.push_back(pp->retained[(size_t)i]);
    DBHIP_CHECK(hipMemsetAsync(pp->ctrl, 0, 24, s));   // cursor, flags, errors; [3] (committed blocks) stays
    pp->retained.clear(); pp->retained_base = committed; pp->submitted = committed; pp->rows_ub = 0; pp->window_blocks = 0;
    for (const FaPending& P : again)
      if ((rc = fa_pipe_enqueue(g, pp, P, s))) return rc;
    return fa_pipe_checkpoint(g, pp, out_committed, false);
  }
  const int64_t submitted = pp->submitted;
  fa_pipe_forget(pp);
  DBHIP_CHECK(hipMemsetAsync(pp->ctrl, 0, 64, s));
  if (out_committed) *out_committed = committed;
  if (flags & 2) {
    set_error("dbhip_groupby_checkpoint: a group key string is longer than 12 bytes; pipelined blocks [%lld, %lld) were not merged",
              (long long)committed, (long long)submitted);
    return DBHIP_ERR_UNSUPPORTED;
  }
  if (flags & 4) {
    set_error("dbhip_groupby_checkpoint: %llu row error(s) in the fused maps (Decimal overflow / divided by zero); pipelined blocks [%lld, %lld) were "
              "not merged: evaluate their maps with dbhip_expr_eval to get the rows", (unsigned long long)errs, (long long)committed, (long long)submitted);
    return DBHIP_ERR_ROW_ERRORS;
  }
  pp->given_up = true;
  set_error("dbhip_groupby_checkpoint: more than %d distinct groups inside one workgroup%s; pipelined blocks [%lld, %lld) were not merged: use the "
            "operator-at-a-time path for them", FA_MAX_SLOTS, (flags & 8) ? " (and the table filled during a merge)" : "", (long long)committed,
            (long long)submitted);
  return DBHIP_ERR_CAPACITY;
}
}  // namespace

// every other entry point that reads or changes a pipelined table's groups passes through here first (GB_DRAIN, k_groupby.hip)
int32_t dbhip_fagg_pipe_drain_internal(dbhip_groupby* g, void* pipe, hipStream_t) {
  FaPipe* pp = (FaPipe*)pipe;
  if (!pp->bound || (pp->submitted == 0 && pp->window_blocks == 0 && pp->batch.empty())) return DBHIP_OK;
  int64_t committed = 0;
  return fa_pipe_checkpoint(g, pp, &committed, true);
}
int32_t dbhip_fagg_pipe_reset_internal(void* pipe, hipStream_t s) {
  FaPipe* pp = (FaPipe*)pipe;
  if (pp->bound) DBHIP_CHECK(hipStreamSynchronize(pp->stream));
  fa_pipe_forget(pp);
  pp->count_seen = 0; pp->slots8 = false; pp->bound = false; pp->given_up = false;
  DBHIP_CHECK(hipMemsetAsync(pp->ctrl, 0, 64, s));
  DBHIP_CHECK(hipStreamSynchronize(s));
  return DBHIP_OK;
}
void dbhip_fagg_pipe_destroy_internal(void* pipe) {
  FaPipe* pp = (FaPipe*)pipe;
  if (!pp) return;
  if (pp->rows) (void)dbhip_free(pp->rows);
  if (pp->ctrl) (void)hipFree(pp->ctrl);
  if (pp->status_host) (void)hipHostFree(pp->status_host);
  for (int i = 0; i < FA_PIPE_RING; ++i) {
    if (pp->tab_host[i]) (void)hipHostFree(pp->tab_host[i]);
    if (pp->tab_dev[i]) (void)hipFree(pp->tab_dev[i]);
    if (pp->tab_ev[i]) (void)hipEventDestroy(pp->tab_ev[i]);
  }
  delete pp;
}


extern "C" {

int32_t dbhip_groupby_add_block_program(dbhip_groupby* g, const dbhip_col* keys, const dbhip_agg_program* prog, int64_t n,
                                        const uint8_t* filter_bitmap, int64_t filter_bit_offset, void* stream) {
  DBHIP_REQUIRE(g && keys && prog && prog->arg_regs, "dbhip_groupby_add_block_program: NULL argument");
  const GbLayout& L = *dbhip_groupby_layout_internal(g);
  if (!dbhip_fagg_layout_ok_internal(L)) {
    set_error("dbhip_groupby_add_block_program: layout outside the fused kernel (<= %d key words, <= %d aggregates, <= %d state words)", FA_KW, FA_MAXA, FA_MAXW);
    return DBHIP_ERR_UNSUPPORTED;
  }
  if (n == 0) return DBHIP_OK;
  hipStream_t s = resolve_stream(stream);
  if (FaPipe* pp = (FaPipe*)*dbhip_groupby_pipe_slot_internal(g); pp && pp->on && !t_prepare_only && !t_jit_only) {
    // ---- pipelined: recognise the query shape by its signature, fill the block's pointers, queue (see above) ----
    if (pp->bound && pp->stream != s) {
      set_error("dbhip_groupby_add_block_program: a pipelined table takes its blocks on ONE stream (checkpoint before changing it)");
      return DBHIP_ERR_INVALID;
    }
    DBHIP_REQUIRE(prog->n_inputs >= 0 && prog->n_inputs <= EX_MAX_INPUTS && (prog->inputs || prog->n_inputs == 0), "expression program: 0..8 input columns");
    DBHIP_REQUIRE(prog->n_ins >= 0 && prog->n_ins <= 2 * EX_MAX_INS && (prog->prog || prog->n_ins == 0), "expression program: too many instructions");
    static thread_local std::string sig;
    sig.clear();
    auto put = [&](const void* p, size_t nb) { sig.append((const char*)p, nb); };
    put(&prog->n_ins, 4); put(&prog->n_inputs, 4); put(&prog->filter_reg, 4);
    if (prog->n_ins) put(prog->prog, (size_t)prog->n_ins * sizeof(dbhip_expr_ins));
    for (int c = 0; c < prog->n_inputs; ++c) {
      const dbhip_col& col = prog->inputs[c];
      DBHIP_REQUIRE(col.data, "expression program: NULL input column");
      const int32_t t[3] = {col.type, col.is_scalar, (int32_t)(col.validity != nullptr) | ((int32_t)col.precision << 8) | ((int32_t)col.scale << 16)};
      put(t, sizeof(t));
    }
    for (int k = 0; k < L.nkeys; ++k) {
      const int32_t t[3] = {keys[k].type, keys[k].is_scalar, (int32_t)(keys[k].validity != nullptr)};
      put(t, sizeof(t));
    }
    put(prog->arg_regs, (size_t)L.naggs * 4);
    const char hf = filter_bitmap != nullptr;
    put(&hf, 1);
    int sh = -1;
    for (int i = (int)pp->shapes.size() - 1; i >= 0 && sh < 0; --i)
      if (pp->shapes[(size_t)i].sig == sig) sh = i;
    if (sh < 0) {
      FaShape S;
      FaArgs A;
      memset(&A, 0, sizeof(A));
      bool may_raise = false;
      const int32_t rc = fa_build_args(L, keys, prog, A, &S.general, &S.nwords, &may_raise);
      if (rc) return rc;
      A.has_filter = filter_bitmap != nullptr;
      S.lds = (size_t)(A.P.n_slots > 6 ? A.P.n_slots : 6) * FA_ROWS * 256 * 8;
      if (S.lds > 60 * 1024) {
        set_error("dbhip_groupby_add_block_program: %d live LDS slots exceed the register file; split the expression", A.P.n_slots);
        return DBHIP_ERR_UNSUPPORTED;
      }
      fa_shape(A, S.A);
      S.sig = sig;
      pp->shapes.push_back(std::move(S));
      sh = (int)pp->shapes.size() - 1;
    }
    FaPending P;
    memset(&P.b, 0, sizeof(P.b));
    P.shape = sh;
    for (int c = 0; c < prog->n_inputs; ++c) {
      P.b.in_data[c] = prog->inputs[c].data; P.b.in_valid[c] = prog->inputs[c].validity; P.b.in_voff[c] = prog->inputs[c].validity_offset;
    }
    for (int k = 0; k < L.nkeys; ++k) { P.b.key_data[k] = keys[k].data; P.b.key_valid[k] = keys[k].validity; P.b.key_voff[k] = keys[k].validity_offset; }
    P.b.filter_bits = filter_bitmap; P.b.filter_off = filter_bit_offset; P.b.n = n;
    pp->bound = true; pp->stream = s;
    return fa_pipe_enqueue(g, pp, P, s);
  }
  FaArgs A;
  memset(&A, 0, sizeof(A));
  bool general = false, may_raise = false;
  int nwords = 0;
  int32_t rc = fa_build_args(L, keys, prog, A, &general, &nwords, &may_raise);
  if (rc) return rc;
  A.filter_bits = filter_bitmap; A.filter_off = filter_bit_offset; A.n = n;
  A.has_filter = filter_bitmap != nullptr;
  // (at least 6 slots: the end of the kernel stages one group's 12 state words per lane in the register file)
  const size_t lds = (size_t)(A.P.n_slots > 6 ? A.P.n_slots : 6) * FA_ROWS * 256 * 8;
  if (lds > 60 * 1024) {
    set_error("dbhip_groupby_add_block_program: %d live LDS slots exceed the register file; split the expression", A.P.n_slots);
    return DBHIP_ERR_UNSUPPORTED;
  }
  // grid: whole multiples of the 256 CUs, 2 workgroups per CU (k_q1.hip's sweep: fewer, longer-running workgroups stream best)
  const int64_t nchunks = ceil_div(n, 64 * FA_ROWS);
  int grid = (int)(ceil_div(nchunks, 4) < 512 ? ceil_div(nchunks, 4) : 512);
  static const int env_grid = exp_env("DBHIP_FAGG_GRID") ? atoi(exp_env("DBHIP_FAGG_GRID")) : 0;
  if (env_grid > 0) grid = (int)(ceil_div(nchunks, 4) < env_grid ? ceil_div(nchunks, 4) : env_grid);
  if (FaPipe* pp = (FaPipe*)*dbhip_groupby_pipe_slot_internal(g); pp && !t_prepare_only) {
    int64_t committed = 0;         // a synchronous call on a table that still has queued blocks: those first
    if ((rc = fa_pipe_checkpoint(g, pp, &committed, true))) return rc;
  }
  const size_t rows_bytes = (size_t)grid * FA_MAX_SLOTS * L.W * 8;   // one partial row per (workgroup, slot)
  uint8_t* ws = (uint8_t*)scratch(rows_bytes + 64, 6, s);
  if (!ws) return DBHIP_ERR_HIP;
  uint64_t* ctrl = (uint64_t*)ws;
  A.ctrl = ctrl;
  A.partial_rows = (uint64_t*)(ws + 64);
  A.P.err_words = nullptr;   // row errors of fused maps: reported through ctrl[2] (count) below
  A.P.err_count = (unsigned long long*)&ctrl[2];
  uint64_t* host_ctrl = pinned_words(1);   // read back asynchronously while the merge is queued behind the kernel
  if (!host_ctrl) return DBHIP_ERR_HIP;
  host_ctrl[0] = host_ctrl[1] = host_ctrl[2] = 0;
  const int64_t n_max = (int64_t)grid * FA_MAX_SLOTS;
  const bool chained = (dbhip_groupby_count_internal(g) + n_max) * 135 <= dbhip_groupby_capacity_internal(g) * 100;
  if ((rc = dbhip_groupby_reserve_merge_internal(g, n_max))) return rc;
  static const bool no_chain = exp_env("DBHIP_FAGG_NOCHAIN") != nullptr;   // debugging: drain the stream between kernel and merge
  if (t_prepare_only) {
    // dbhip_groupby_prepare_program: compile the specialised kernel of this query shape now (the 4-slot variant, or the
    // 8-slot one when the table already holds more than 4 groups), launch nothing
    // (the specialised kernel is compiled for exactly `nwords` state words: no accumulator registers for words the layout lacks)
    (void)jit_kernel(A, dbhip_groupby_count_internal(g) > 4 ? 8 : 4, general, nwords, JIT_COMPILE);
    // a pipelined table launches its small blocks together: that kernel is a specialisation of its own (FA_MULTI)
    if (FaPipe* pq = (FaPipe*)*dbhip_groupby_pipe_slot_internal(g); pq && pq->on)
      (void)jit_kernel(A, dbhip_groupby_count_internal(g) > 4 ? 8 : 4, general, nwords, JIT_COMPILE, nullptr, true);
    return DBHIP_OK;
  }
  // a table that already holds more than 4 groups starts with the 8-slot variant
  for (int variant = dbhip_groupby_count_internal(g) > 4 ? 1 : 0; variant < 2; ++variant) {
    DBHIP_CHECK(hipMemsetAsync(ctrl, 0, 64, s));
    kernel_timer_start(s);
    if ((rc = fa_launch(A, variant == 0 ? 4 : 8, general, nwords, grid, lds, s))) return rc;
    kernel_timer_stop(s);
    DBHIP_LAUNCH_CHECK();
    DBHIP_CHECK(hipMemcpyAsync(host_ctrl, ctrl, 24, hipMemcpyDeviceToHost, s));
    if (chained && !may_raise && !no_chain) {
      // the merge of the partial rows is queued right behind the kernel with the row count and the give-up flags still
      // on the device: ONE host round trip per pass
      rc = dbhip_groupby_merge_rows_dev_internal(g, A.partial_rows, n_max, &ctrl[0], &ctrl[1], s);
      if (rc) return rc;  // (synchronises the stream: host_ctrl is valid now)
      if (!(host_ctrl[1] & 3)) return DBHIP_OK;
    } else {
      DBHIP_CHECK(hipStreamSynchronize(s));
    }
    if (!(host_ctrl[1] & 1)) break;
  }
  if (host_ctrl[1] & 2) {
    set_error("dbhip_groupby_add_block_program: a group key string is longer than 12 bytes");
    return DBHIP_ERR_UNSUPPORTED;
  }
  if (host_ctrl[1] & 1) {
    set_error("dbhip_groupby_add_block_program: more than %d distinct groups inside one workgroup; use the operator-at-a-time path", FA_MAX_SLOTS);
    return DBHIP_ERR_CAPACITY;
  }
  if (host_ctrl[2]) {
    // a map raised for a live row: like the reference, the block fails as a whole and nothing is merged
    set_error("dbhip_groupby_add_block_program: %llu row error(s) in the fused maps (Decimal overflow / divided by zero); "
              "evaluate the maps with dbhip_expr_eval to get the rows", (unsigned long long)host_ctrl[2]);
    return DBHIP_ERR_ROW_ERRORS;
  }
  return dbhip_groupby_merge_rows_internal(g, A.partial_rows, (int64_t)host_ctrl[0], s);
}

int32_t dbhip_groupby_set_pipelined(dbhip_groupby* g, int32_t on, void* stream) {
  DBHIP_REQUIRE(g, "dbhip_groupby_set_pipelined: NULL argument");
  void** slot = dbhip_groupby_pipe_slot_internal(g);
  FaPipe* pp = (FaPipe*)*slot;
  if (!on) {
    if (!pp) return DBHIP_OK;
    int64_t committed = 0;
    const int32_t rc = fa_pipe_checkpoint(g, pp, &committed, true);
    pp->on = false; pp->bound = false;
    return rc;
  }
  const GbLayout& L = *dbhip_groupby_layout_internal(g);
  if (!dbhip_fagg_layout_ok_internal(L)) {
    set_error("dbhip_groupby_set_pipelined: layout outside the fused kernel (<= %d key words, <= %d aggregates, <= %d state words)", FA_KW, FA_MAXA, FA_MAXW);
    return DBHIP_ERR_UNSUPPORTED;
  }
  if (pp) { pp->on = true; return DBHIP_OK; }
  pp = new (std::nothrow) FaPipe();
  DBHIP_REQUIRE(pp, "dbhip_groupby_set_pipelined: out of host memory");
  hipStream_t s = resolve_stream(stream);
  // 65,536 partial rows: 16 blocks of the largest grid (512 workgroups x 8 slots); small blocks close their window by count (FA_PIPE_WINDOW)
  pp->cap_rows = 65536;
  int32_t rc = dbhip_alloc((size_t)pp->cap_rows * L.W * 8, (void**)&pp->rows);
  hipError_t e = rc == DBHIP_OK ? hipMalloc((void**)&pp->ctrl, 64) : hipSuccess;
  if (rc == DBHIP_OK && e == hipSuccess) e = hipHostMalloc((void**)&pp->status_host, 64, hipHostMallocMapped);
  if (rc == DBHIP_OK && e == hipSuccess) e = hipHostGetDevicePointer((void**)&pp->status_dev, pp->status_host, 0);
  if (rc == DBHIP_OK && e == hipSuccess) { memset(pp->status_host, 0, 64); e = hipMemsetAsync(pp->ctrl, 0, 64, s); }
  if (rc == DBHIP_OK && e == hipSuccess) e = hipStreamSynchronize(s);
  for (int i = 0; i < FA_PIPE_RING && rc == DBHIP_OK && e == hipSuccess; ++i) {
    e = hipHostMalloc((void**)&pp->tab_host[i], FA_PIPE_TABLE_BYTES, hipHostMallocDefault);
    if (e == hipSuccess) e = hipMalloc((void**)&pp->tab_dev[i], FA_PIPE_TABLE_BYTES);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&pp->tab_ev[i], hipEventDisableTiming);
  }
  if (rc == DBHIP_OK && e == hipSuccess) rc = dbhip_groupby_reserve_merge_internal(g, pp->cap_rows);
  if (rc != DBHIP_OK || e != hipSuccess) {
    dbhip_fagg_pipe_destroy_internal(pp);
    return rc != DBHIP_OK ? rc : hip_fail(e, "dbhip_groupby_set_pipelined");
  }
  pp->count_seen = dbhip_groupby_count_internal(g);
  pp->on = true;
  *slot = pp;
  return DBHIP_OK;
}

int32_t dbhip_groupby_checkpoint(dbhip_groupby* g, int64_t* out_blocks_committed_host, void* stream) {
  DBHIP_REQUIRE(g, "dbhip_groupby_checkpoint: NULL argument");
  if (out_blocks_committed_host) *out_blocks_committed_host = 0;
  FaPipe* pp = (FaPipe*)*dbhip_groupby_pipe_slot_internal(g);
  if (!pp) return DBHIP_OK;
  if (pp->bound && stream && pp->stream != resolve_stream(stream)) {
    set_error("dbhip_groupby_checkpoint: the table's blocks were queued on another stream");
    return DBHIP_ERR_INVALID;
  }
  const int32_t rc = fa_pipe_checkpoint(g, pp, out_blocks_committed_host, true);
  pp->bound = false;   // nothing queued any more: the next block may come on another stream
  return rc;
}

int32_t dbhip_groupby_prepare_program(dbhip_groupby* g, const dbhip_col* keys, const dbhip_agg_program* prog) {
  t_prepare_only = true;
  const int32_t rc = dbhip_groupby_add_block_program(g, keys, prog, 1, nullptr, 0, nullptr);
  t_prepare_only = false;
  return rc;
}

}  // extern "C"

// add_block's own use of the fused kernel (k_groupby.hip, tables that showed <= 8 groups in their probing chunk): rows
// [row0, row0 + n) of the key / argument columns `C`, arguments as they are (an empty program whose results are input
// columns). DBHIP_ERR_UNSUPPORTED: shape outside the kernel (the caller keeps its LDS path); DBHIP_ERR_CAPACITY: a
// workgroup met a 9th group (nothing merged).
static bool fa_offset_col(const GbCol& c, int64_t row0, dbhip_col* out) {
  memset(out, 0, sizeof(*out));
  out->type = c.type; out->is_scalar = c.is_scalar; out->buffers = c.buffers; out->validity = c.validity;
  out->validity_offset = c.voff + (c.is_scalar ? 0 : row0);
  if (c.is_scalar || row0 == 0) { out->data = c.data; return true; }
  if (c.type == DBHIP_T_BOOL) {
    if (row0 & 7) return false;
    out->data = (const uint8_t*)c.data + (row0 >> 3);
    return true;
  }
  const int es = type_size(c.type);
  if (es == 0) return false;
  out->data = (const uint8_t*)c.data + (size_t)row0 * es;
  return true;
}

int32_t dbhip_fagg_add_columns_internal(dbhip_groupby* g, const GbCols& C, int64_t row0, int64_t n, bool may_compile, hipStream_t s) {
  const GbLayout& L = *dbhip_groupby_layout_internal(g);
  if (!dbhip_fagg_layout_ok_internal(L)) return DBHIP_ERR_UNSUPPORTED;
  dbhip_col keys[FA_KW], inputs[EX_MAX_INPUTS];
  int32_t arg_regs[FA_MAXA];
  int n_inputs = 0;
  for (int k = 0; k < L.nkeys; ++k)
    if (!fa_offset_col(C.key[k], row0, &keys[k])) return DBHIP_ERR_UNSUPPORTED;
  bool any_arg = false;
  for (int a = 0; a < L.naggs; ++a) {
    arg_regs[a] = DBHIP_ARG_NONE;
    if (C.arg[a].data == nullptr) {
      if (L.agg_kind[a] != DBHIP_AGG_COUNT) return DBHIP_ERR_UNSUPPORTED;
      continue;
    }
    if (!fa_arg_type_ok(C.arg[a].type)) return DBHIP_ERR_UNSUPPORTED;
    int found = -1;   // the same column under several aggregates is loaded once
    dbhip_col col;
    if (!fa_offset_col(C.arg[a], row0, &col)) return DBHIP_ERR_UNSUPPORTED;
    for (int c = 0; c < n_inputs && found < 0; ++c)
      if (inputs[c].data == col.data && inputs[c].type == col.type && inputs[c].validity == col.validity &&
          inputs[c].validity_offset == col.validity_offset && inputs[c].is_scalar == col.is_scalar) found = c;
    if (found < 0) {
      if (n_inputs >= EX_MAX_INPUTS) return DBHIP_ERR_UNSUPPORTED;
      if (col.type == DBHIP_T_DEC128) { int wide = 0; for (int c = 0; c < n_inputs; ++c) wide += inputs[c].type == DBHIP_T_DEC128; if (wide >= 2) return DBHIP_ERR_UNSUPPORTED; }
      inputs[n_inputs] = col;
      found = n_inputs++;
    }
    arg_regs[a] = DBHIP_ARG_INPUT(found);
    any_arg = true;
  }
  if (!any_arg) return DBHIP_ERR_UNSUPPORTED;  // count(*) only: the LDS path is fine for that
  dbhip_agg_program prog;
  prog.prog = nullptr; prog.n_ins = 0; prog.inputs = inputs; prog.n_inputs = n_inputs; prog.filter_reg = -1; prog.arg_regs = arg_regs;
  t_jit_only = true; t_jit_pending = false; t_jit_may_compile = may_compile;
  const int32_t rc = dbhip_groupby_add_block_program(g, keys, &prog, n, C.filter, C.filter_off + row0, (void*)s);
  t_jit_only = false;
  return rc;
}
// Plain add_block / add_block_filtered on a PIPELINED table (k_groupby.hip calls this first): the block's key and argument columns as a
// program without instructions, queued like any other block. -1: not taken (the table is not pipelined, gave up earlier, or the
// shape is outside the fused kernel) — the caller checkpoints and runs its synchronous paths.
int32_t dbhip_fagg_pipe_add_columns_internal(dbhip_groupby* g, void* pipe, const GbCols& C, int64_t n, hipStream_t s) {
  FaPipe* pp = (FaPipe*)pipe;
  if (!pp || !pp->on || pp->given_up || jit_mode() == 0) return -1;
  const GbLayout& L = *dbhip_groupby_layout_internal(g);
  if (!dbhip_fagg_layout_ok_internal(L)) return -1;
  dbhip_col keys[FA_KW], inputs[EX_MAX_INPUTS];
  int32_t arg_regs[FA_MAXA];
  int n_inputs = 0;
  for (int k = 0; k < L.nkeys; ++k)
    if (!fa_offset_col(C.key[k], 0, &keys[k])) return -1;
  bool any_arg = false;
  for (int a = 0; a < L.naggs; ++a) {
    arg_regs[a] = DBHIP_ARG_NONE;
    if (C.arg[a].data == nullptr) {
      if (L.agg_kind[a] != DBHIP_AGG_COUNT) return -1;
      continue;
    }
    if (!fa_arg_type_ok(C.arg[a].type)) return -1;
    dbhip_col col;
    if (!fa_offset_col(C.arg[a], 0, &col)) return -1;
    int found = -1;
    for (int c = 0; c < n_inputs && found < 0; ++c)
      if (inputs[c].data == col.data && inputs[c].type == col.type && inputs[c].validity == col.validity &&
          inputs[c].validity_offset == col.validity_offset && inputs[c].is_scalar == col.is_scalar) found = c;
    if (found < 0) {
      if (n_inputs >= EX_MAX_INPUTS) return -1;
      if (col.type == DBHIP_T_DEC128) { int wide = 0; for (int c = 0; c < n_inputs; ++c) wide += inputs[c].type == DBHIP_T_DEC128; if (wide >= 2) return -1; }
      inputs[n_inputs] = col;
      found = n_inputs++;
    }
    arg_regs[a] = DBHIP_ARG_INPUT(found);
    any_arg = true;
  }
  if (!any_arg) return -1;   // count(*) only
  dbhip_agg_program prog;
  prog.prog = nullptr; prog.n_ins = 0; prog.inputs = inputs; prog.n_inputs = n_inputs; prog.filter_reg = -1; prog.arg_regs = arg_regs;
  const int32_t rc = dbhip_groupby_add_block_program(g, keys, &prog, n, C.filter, C.filter_off, (void*)s);
  return rc == DBHIP_ERR_UNSUPPORTED ? -1 : rc;
}

// after DBHIP_ERR_UNSUPPORTED from dbhip_fagg_add_columns_internal: true = only for now (the kernel is being compiled)
bool dbhip_fagg_last_refusal_is_pending_internal() { return t_jit_pending; }

// Offline twin of dbhip_groupby_prepare_program (needs no device; tools/jit_offline.py): the specialised kernel's code object for
// a table layout + program, written to `code_out` (returns its size, -1 with hiprtc's log in `log_out` on failure, -2 when the
// shape is outside the fused kernel). The columns only need their types / scalar-ness / validity-ness (any non-null pointers).
int32_t dbhip_groupby_build_layout_internal(const int32_t* key_types, const uint8_t* key_nullable, int nkeys, const dbhip_agg_desc* aggs,
                                            int naggs, GbLayout* L);
extern "C" int64_t dbhip_jit_offline(const int32_t* key_types, const uint8_t* key_nullable, int32_t nkeys,
                                                   const dbhip_agg_desc* aggs, int32_t naggs, const dbhip_col* keys,
                                                   const dbhip_agg_program* prog, int32_t slots, char* code_out, int64_t code_cap,
                                                   char* log_out, int64_t log_cap) {
  GbLayout L;
  if (dbhip_groupby_build_layout_internal(key_types, key_nullable, nkeys, aggs, naggs, &L) || !dbhip_fagg_layout_ok_internal(L)) return -2;
  FaArgs A;
  memset(&A, 0, sizeof(A));
  bool general = false, may_raise = false;
  int nwords = 0;
  if (fa_build_args(L, keys, prog, A, &general, &nwords, &may_raise)) return -2;
  std::vector<char> code;
  std::string log;
  const bool multi = (slots & 0x100) != 0;   // slots | 0x100: the FA_MULTI specialisation (multi-block launches of a pipelined table)
  slots &= 0xFF;
  const bool ok = jit_compile(jit_meta(A, multi), jit_tail(slots, general, nwords), &code, &log);
  if (log_out && log_cap > 0) snprintf(log_out, (size_t)log_cap, "%s", log.c_str());
  if (!ok) return -1;
  if ((int64_t)code.size() <= code_cap) memcpy(code_out, code.data(), code.size());
  return (int64_t)code.size();
}

// Compiles the run-time specialisation of a small fixed query shape (i64 key; sum(i64 column), count(*)) and returns the
// size of the code object (> 0) or -1 with hiprtc's log in `log_out`. Needs no device: tests/test_abi.py runs it on the CPU
// box so that a header the specialised kernel cannot digest is caught where there is no GPU.
extern "C" int64_t dbhip_jit_compile_check(char* log_out, int64_t cap) {
  FaArgs A;
  memset(&A, 0, sizeof(A));
  for (int i = 0; i < EX_MAX_INPUTS; ++i) { A.P.in_slot[i] = -1; A.P.in_wide_ord[i] = -1; }
  A.P.n_inputs = 1; A.P.in_type[0] = LK_8; A.P.in_slot[0] = 0; A.P.n_slots = 1; A.P.filter_slot = -1;
  // one real instruction so that the interpreter's code is compiled too: r1 = r0 + r0 (i64)
  A.P.n_ins = 1;
  A.P.ins[0].op = EX_PLUS; A.P.ins[0].dst = 1; A.P.ins[0].a = 0; A.P.ins[0].b = 0; A.P.ins[0].acls = A.P.ins[0].bcls = A.P.ins[0].ocls = CLS_SIGNED;
  A.P.ins[0].dec_idx = -1; A.P.n_slots = 2;
  A.nkeys = 1; A.nkey_words = 1; A.validity_word = -1; A.hash_word = 1; A.W = 4; A.naggs = 2; A.nwords = 2; A.state_off = 2;
  A.key[0].type = DBHIP_T_I64; A.key_type[0] = DBHIP_T_I64; A.key_off[0] = 0; A.key_words[0] = 1;
  A.wm[0] = (uint32_t)W_ADD1 | ((uint32_t)C_LO << 3) | (1u << 6) | (1u << 13) | (1u << 15) | ((uint32_t)DBHIP_T_I64 << 24);
  A.wm[1] = (uint32_t)W_ADD1 | ((uint32_t)C_FLAG << 3) | (1u << 15);
  std::vector<char> code;
  std::string log;
  const bool ok = jit_compile(jit_meta(A), jit_tail(4, false, 4), &code, &log);
  if (log_out && cap > 0) { snprintf(log_out, (size_t)cap, "%s", log.c_str()); }
  return ok ? (int64_t)code.size() : -1;
}

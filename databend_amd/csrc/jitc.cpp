// dbhip_jitc — out-of-process hiprtc driver for the run-time specialised kernels of libdbhip.so (k_fagg.hip).
//
//   dbhip_jitc [--detach] [--publish <final.co>] [--rmdir <dir>] <source.hip> <out.co> [hiprtc options...]
//
//   --detach    fork, let the parent return at once and compile in the (re-parented) child: the library's non-blocking mode —
//               a query never waits for the compiler, the kernel is found in the on-disk cache once it exists
//   --publish   rename <out.co> to <final.co> when the compile succeeded (atomic: a reader sees the whole file or none)
//   --rmdir     afterwards remove every file in <dir> and <dir> itself (the temporary directory with the sources)
//
// Why a process of its own: hiprtc (comgr) called inside a process that also drives the GPU hung intermittently on the MI355X
// boxes of round 2 (r02j3 on a worker thread, r02j6 on the calling thread: the second compile of a process never returned).
// The library therefore never calls the compiler itself: it writes the sources to a temporary directory, runs this helper
// with a deadline, and loads the code object — a helper that hangs is killed and the interpreting kernel stays in place.
#include <hip/hiprtc.h>
#include <dirent.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include <string>
#include <vector>

static bool slurp(const char* path, std::string* out) {
  FILE* f = fopen(path, "rb");
  if (!f) return false;
  char buf[65536];
  size_t n;
  while ((n = fread(buf, 1, sizeof(buf), f)) > 0) out->append(buf, n);
  fclose(f);
  return true;
}

static void remove_dir(const char* dir) {
  if (DIR* d = opendir(dir)) {
    while (dirent* e = readdir(d)) {
      if (!strcmp(e->d_name, ".") || !strcmp(e->d_name, "..")) continue;
      unlink((std::string(dir) + "/" + e->d_name).c_str());
    }
    closedir(d);
  }
  rmdir(dir);
}

static int compile(int argc, char** argv);

int main(int argc, char** argv) {
  bool detach = false;
  const char* publish = nullptr;
  const char* rm = nullptr;
  int i = 1;
  for (; i < argc; ++i) {
    if (!strcmp(argv[i], "--detach")) detach = true;
    else if (!strcmp(argv[i], "--publish") && i + 1 < argc) publish = argv[++i];
    else if (!strcmp(argv[i], "--rmdir") && i + 1 < argc) rm = argv[++i];
    else break;
  }
  if (argc - i < 2) { fprintf(stderr, "usage: dbhip_jitc [--detach] [--publish final.co] [--rmdir dir] <source.hip> <out.co> [hiprtc options...]\n"); return 2; }
  if (detach) {
    const pid_t p = fork();
    if (p > 0) return 0;       // the caller's waitpid returns at once
    if (p == 0) setsid();      // (p < 0: compile in the foreground)
  }
  int rc = compile(argc - i + 1, argv + i - 1);
  if (rc == 0 && publish && rename(argv[i + 1], publish) != 0) rc = 6;
  if (rm) remove_dir(rm);
  return rc;
}

static int compile(int argc, char** argv) {
  std::string src;
  if (!slurp(argv[1], &src)) { fprintf(stderr, "dbhip_jitc: cannot read %s\n", argv[1]); return 2; }
  hiprtcProgram prog;
  if (hiprtcCreateProgram(&prog, src.c_str(), "dbhip_jit.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS) { fprintf(stderr, "dbhip_jitc: hiprtcCreateProgram failed\n"); return 3; }
  std::vector<const char*> opts(argv + 3, argv + argc);
  const hiprtcResult r = hiprtcCompileProgram(prog, (int)opts.size(), opts.data());
  size_t ls = 0;
  hiprtcGetProgramLogSize(prog, &ls);
  if (ls > 1) {
    std::string log(ls, 0);
    hiprtcGetProgramLog(prog, &log[0]);
    fputs(log.c_str(), stderr);
  }
  if (r != HIPRTC_SUCCESS) return 4;
  size_t cs = 0;
  hiprtcGetCodeSize(prog, &cs);
  std::vector<char> code(cs);
  hiprtcGetCode(prog, code.data());
  FILE* f = fopen(argv[2], "wb");
  if (!f || fwrite(code.data(), 1, cs, f) != cs) { fprintf(stderr, "dbhip_jitc: cannot write %s\n", argv[2]); return 5; }
  fclose(f);
  return 0;
}

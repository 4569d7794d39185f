// div_host_check.cpp — TEST INFRASTRUCTURE: dev_common.h's exact 128 / 64 division (two double-precision estimates, round 5) against the
// host compiler's 128-bit `/` and `%`: edge values x edge divisors, then random triples.   div_host_check [millions of random cases]
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <random>

#define __host__
#define __device__
#define __forceinline__ inline
typedef unsigned __int128 u128;
typedef __int128 i128;
// the two functions under test, cut out of dev_common.h by the test driver (tests/test_div_cpu.py) into div_under_test.h
#include "div_under_test.h"

static int check(u128 n, uint64_t d) {
  uint64_t rem = 0;
  const u128 q = udiv128_by_64(n, d, &rem);
  if (q != n / d || rem != (uint64_t)(n % d)) {
    printf("MISMATCH n = %016llx%016llx d = %016llx\n", (unsigned long long)(n >> 64), (unsigned long long)n, (unsigned long long)d);
    return 1;
  }
  return 0;
}

int main(int argc, char** argv) {
  const long long millions = argc > 1 ? atoll(argv[1]) : 20;
  const uint64_t edge[] = {0, 1, 2, 3, 9, 10, 99, 100, 0xFFFFFFFFULL, 0x100000000ULL, 0x100000001ULL, 0x7FFFFFFFFFFFFFFFULL, 0x8000000000000000ULL,
                           0x8000000000000001ULL, 0xFFFFFFFFFFFFFFFEULL, 0xFFFFFFFFFFFFFFFFULL, 10000000000000000000ULL, 1000000000000000000ULL,
                           (1ULL << 53) - 1, 1ULL << 53, (1ULL << 53) + 1, (1ULL << 52) + 12345, 0x001FFFFFFFFFFFFFULL, 0x0020000000000001ULL};
  const int ne = (int)(sizeof(edge) / sizeof(edge[0]));
  long long bad = 0, done = 0;
  for (int a = 0; a < ne; ++a)
    for (int b = 0; b < ne; ++b)
      for (int c = 1; c < ne; ++c) {
        if (edge[c] == 0) continue;
        bad += check(((u128)edge[a] << 64) | edge[b], edge[c]);
        // just below / above multiples of the divisor
        const u128 m = (u128)edge[a] * edge[c];
        bad += check(m, edge[c]);
        bad += check(m + edge[c] - 1, edge[c]);
        if (m) bad += check(m - 1, edge[c]);
        done += 4;
      }
  std::mt19937_64 r(12345);
  for (long long i = 0; i < millions * 1000000LL && !bad; ++i) {
    uint64_t hi = r(), lo = r(), d = r();
    const int kind = (int)(i & 7);
    if (kind == 1) d >>= (r() & 63);                // small divisors
    if (kind == 2) hi >>= (r() & 63);               // small numerators
    if (kind == 3) { d >>= (r() & 63); hi = 0; }    // 64 / 64
    if (kind == 4) d = 10000000000000000000ULL >> (r() % 20) | 1;   // around powers of ten's neighbourhood
    if (kind == 5) { static const uint64_t p10[] = {10ULL, 10000ULL, 100000000ULL, 1000000000000ULL, 10000000000000000000ULL}; d = p10[r() % 5]; }
    if (kind == 6) { d |= 0x8000000000000000ULL; }   // top bit set
    if (d == 0) d = 1;
    bad += check(((u128)hi << 64) | lo, d);
    ++done;
  }
  if (bad) { printf("FAILED %lld\n", bad); return 1; }
  printf("ok %lld divisions\n", done);
  return 0;
}

// Host twin of the device's Decimal256 row functions: compiles databend_amd/csrc/dev_i256.h with g++ (the header is
// __host__ __device__ and free of HIP types) and evaluates one request per input line, so tests/test_dec256_cpu.py can check
// the PRODUCT's 256-bit arithmetic (32-bit-limb long division, wrap rules, BigInt fallbacks) against Python integers without a
// GPU. Values travel as 64 hex digits of the two's complement bit pattern.
//   B op a_dec ap as b_dec bp bs x y       -> "ok <hex>" | "err" | "nofn"
//   C src_bits fp fs dp ds rounding x      -> "ok <hex>" | "err" | "nofn"
//   K abits(ignored) ap as bp bs x y       -> "-1" | "0" | "1"
#include <cstdio>
#include <cstring>
#include <string>

#include "../databend_amd/csrc/dev_i256.h"
using namespace dbhip;

static I256 parse_hex(const char* s) {
  I256 r{};
  for (int i = 0; i < 64; ++i) {
    const char c = s[i];
    const uint64_t d = c <= '9' ? c - '0' : (c | 32) - 'a' + 10;
    const int bit = (63 - i) * 4;
    r.w[bit >> 6] |= d << (bit & 63);
  }
  return r;
}
static void print_hex(const I256& v) {
  printf("ok %016llx%016llx%016llx%016llx\n", (unsigned long long)v.w[3], (unsigned long long)v.w[2], (unsigned long long)v.w[1],
         (unsigned long long)v.w[0]);
}

int main() {
  char line[1024], xs[128], ys[128];
  while (fgets(line, sizeof line, stdin)) {
    if (line[0] == 'B') {
      int op, ad, ap, as, bd, bp, bs;
      if (sscanf(line + 1, "%d %d %d %d %d %d %d %64s %64s", &op, &ad, &ap, &as, &bd, &bp, &bs, xs, ys) != 9) return 2;
      Dec256Op p;
      DecSize ret;
      if (!dec256_make_op(op, ad, {ap, as}, bd, {bp, bs}, &p, &ret)) { puts("nofn"); continue; }
      I256 out;
      if (dec256_row(p, parse_hex(xs), parse_hex(ys), &out)) print_hex(out); else puts("err");
    } else if (line[0] == 'C') {
      int sb, fp, fs, dp, ds, rnd;
      if (sscanf(line + 1, "%d %d %d %d %d %d %64s", &sb, &fp, &fs, &dp, &ds, &rnd, xs) != 7) return 2;
      Dec256Cast c;
      if (!dec256_make_cast(sb, {fp, fs}, {dp, ds}, rnd != 0, &c)) { puts("nofn"); continue; }
      I256 out;
      if (dec256_cast_row(c, parse_hex(xs), &out)) print_hex(out); else puts("err");
    } else if (line[0] == 'K') {
      int ab, ap, as, bp, bs;
      if (sscanf(line + 1, "%d %d %d %d %d %64s %64s", &ab, &ap, &as, &bp, &bs, xs, ys) != 7) return 2;
      const int scale = as > bs ? as : bs;
      int precision = ((ap - as) > (bp - bs) ? (ap - as) : (bp - bs)) + scale;
      const int cap = (ap <= 38 && bp <= 38) ? 38 : 76;
      if (precision > cap) precision = cap;
      const int bits = dec_storage_bits(precision);
      const I256 fa = i256_pow10(scale - as), fb = i256_pow10(scale - bs);
      printf("%d\n", dec256_cmp3(parse_hex(xs), parse_hex(ys), fa, fb, scale == as, scale == bs, as == bs, bits));
    }
  }
  return 0;
}

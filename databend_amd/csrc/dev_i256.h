// dev_i256.h — the 256-bit integer class behind DecimalColumn::Decimal256 (the reference: ethnum::I256 wrapped as `i256`,
// src/query/expression/src/types/decimal.rs:1282-1500, :2860-2990) and the row functions of the decimal operators that
// involve it:
//   binary_decimal in T = i256      functions/src/scalars/decimal/src/arithmetic.rs:190-316
//   do_round_mul / do_round_div     decimal.rs:1343-1404 (checked path that wraps like ethnum's release-mode + - *, BigInt
//                                   fallback when the 256-bit product overflows), from_bigint :1460-1487
//   DecimalCmp across sizes         decimal/src/comparison.rs:326-441
//   decimal -> decimal CAST         decimal/src/cast.rs:790-1035 (shrink / expand / scale reduction with rounding_mode),
//   integer -> decimal              cast.rs:701-753
// Layout: 4 little-endian u64 limbs, two's complement (`to_u64_array` is a transmute, decimal.rs:1285-1291).
// Wide products and quotients are computed on sign + magnitude with 32-bit limbs (Knuth's algorithm D); nothing here is
// shared with the CPU checker (oracle/decimal256.c: one 640-bit type, bit-serial division).
// Everything is __host__ __device__ and free of HIP types, so tests/test_dec256_cpu.py's twin (tests/i256_host_check.cpp)
// compiles this very header with g++ and checks it against Python integers without a GPU.
#pragma once
#include <stdint.h>

#ifndef DBHIP_HD
#if defined(__HIPCC__)
#define DBHIP_HD __host__ __device__ inline
#else
#define DBHIP_HD inline
#endif
#endif

namespace dbhip {

struct I256 {
  uint64_t w[4];
};

DBHIP_HD I256 i256_from_i64(int64_t v) {
  I256 r;
  r.w[0] = (uint64_t)v;
  r.w[1] = r.w[2] = r.w[3] = v < 0 ? ~0ULL : 0ULL;
  return r;
}
DBHIP_HD I256 i256_from_u64(uint64_t v) {
  I256 r;
  r.w[0] = v;
  r.w[1] = r.w[2] = r.w[3] = 0;
  return r;
}
DBHIP_HD I256 i256_from_i128_words(uint64_t lo, uint64_t hi) {
  I256 r;
  r.w[0] = lo;
  r.w[1] = hi;
  r.w[2] = r.w[3] = (hi >> 63) ? ~0ULL : 0ULL;
  return r;
}
DBHIP_HD bool i256_is_neg(const I256& a) { return (a.w[3] >> 63) != 0; }
DBHIP_HD bool i256_is_zero(const I256& a) { return (a.w[0] | a.w[1] | a.w[2] | a.w[3]) == 0; }
DBHIP_HD bool i256_eq(const I256& a, const I256& b) {
  return a.w[0] == b.w[0] && a.w[1] == b.w[1] && a.w[2] == b.w[2] && a.w[3] == b.w[3];
}
DBHIP_HD int i256_ucmp(const I256& a, const I256& b) {
#pragma unroll
  for (int i = 3; i >= 0; --i)
    if (a.w[i] != b.w[i]) return a.w[i] > b.w[i] ? 1 : -1;
  return 0;
}
DBHIP_HD int i256_cmp(const I256& a, const I256& b) {
  const bool na = i256_is_neg(a), nb = i256_is_neg(b);
  if (na != nb) return na ? -1 : 1;
  return i256_ucmp(a, b);
}
DBHIP_HD I256 i256_add(const I256& a, const I256& b) {  // wrapping
  I256 r;
  uint64_t c = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const uint64_t s = a.w[i] + b.w[i];
    const uint64_t c1 = s < a.w[i];
    r.w[i] = s + c;
    c = c1 | (r.w[i] < s);
  }
  return r;
}
DBHIP_HD I256 i256_not(const I256& a) {
  I256 r;
#pragma unroll
  for (int i = 0; i < 4; ++i) r.w[i] = ~a.w[i];
  return r;
}
DBHIP_HD I256 i256_neg(const I256& a) { return i256_add(i256_not(a), i256_from_u64(1)); }  // wrapping
DBHIP_HD I256 i256_sub(const I256& a, const I256& b) { return i256_add(a, i256_neg(b)); }
DBHIP_HD I256 i256_abs(const I256& a) { return i256_is_neg(a) ? i256_neg(a) : a; }           // |MIN| = 2^255 as unsigned
// sign-extends the low `bits` (64 / 128 / 256): `as_decimal::<T>()` into a narrower T and back into the carrier
DBHIP_HD I256 i256_wrap(const I256& a, int bits) {
  if (bits == 64) return i256_from_i64((int64_t)a.w[0]);
  if (bits == 128) return i256_from_i128_words(a.w[0], a.w[1]);
  return a;
}
DBHIP_HD bool i256_fits(const I256& a, int bits) { return i256_eq(i256_wrap(a, bits), a); }

// ---- magnitudes: little-endian 32-bit limbs -----------------------------------------------------------------------------
constexpr int I256_MAG_MAX = 20;  // |x| * 10^82 needs 18 limbs, + 1 for the normalisation shift

DBHIP_HD void mag_from_i256(const I256& a, uint32_t* out /* 8 */) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    out[2 * i] = (uint32_t)a.w[i];
    out[2 * i + 1] = (uint32_t)(a.w[i] >> 32);
  }
}
DBHIP_HD int mag_len(const uint32_t* a, int n) {
  while (n > 0 && a[n - 1] == 0) --n;
  return n;
}
// out[na + nb] = a * b
DBHIP_HD void mag_mul(const uint32_t* a, int na, const uint32_t* b, int nb, uint32_t* out) {
  for (int i = 0; i < na + nb; ++i) out[i] = 0;
  for (int i = 0; i < na; ++i) {
    uint64_t c = 0;
    for (int j = 0; j < nb; ++j) {
      c += (uint64_t)a[i] * b[j] + out[i + j];
      out[i + j] = (uint32_t)c;
      c >>= 32;
    }
    out[i + nb] = (uint32_t)c;
  }
}
// a[na] += b[nb] (nb <= na); returns the carry out
DBHIP_HD uint32_t mag_add_into(uint32_t* a, int na, const uint32_t* b, int nb) {
  uint64_t c = 0;
  for (int i = 0; i < na; ++i) {
    c += (uint64_t)a[i] + (i < nb ? b[i] : 0u);
    a[i] = (uint32_t)c;
    c >>= 32;
  }
  return (uint32_t)c;
}
// q[nu] = floor(u / v): u of nu limbs, v of nv limbs (v != 0). Knuth TAOCP 4.3.1 algorithm D, base 2^32.
DBHIP_HD void mag_div(const uint32_t* u, int nu, const uint32_t* v, int nv, uint32_t* q) {
  for (int i = 0; i < nu; ++i) q[i] = 0;
  const int m = mag_len(u, nu), n = mag_len(v, nv);
  if (m < n || n == 0) return;
  if (n == 1) {
    uint64_t r = 0;
    for (int i = m - 1; i >= 0; --i) {
      const uint64_t cur = (r << 32) | u[i];
      q[i] = (uint32_t)(cur / v[0]);
      r = cur % v[0];
    }
    return;
  }
  uint32_t un[I256_MAG_MAX + 1], vn[I256_MAG_MAX];
  const int s = __builtin_clz(v[n - 1]);
  for (int i = n - 1; i > 0; --i) vn[i] = s ? (v[i] << s) | (v[i - 1] >> (32 - s)) : v[i];
  vn[0] = v[0] << s;
  un[m] = s ? u[m - 1] >> (32 - s) : 0;
  for (int i = m - 1; i > 0; --i) un[i] = s ? (u[i] << s) | (u[i - 1] >> (32 - s)) : u[i];
  un[0] = u[0] << s;
  for (int j = m - n; j >= 0; --j) {
    const uint64_t num = ((uint64_t)un[j + n] << 32) | un[j + n - 1];
    uint64_t qhat = num / vn[n - 1], rhat = num % vn[n - 1];
    while (qhat >= (1ULL << 32) || qhat * vn[n - 2] > ((rhat << 32) | un[j + n - 2])) {
      --qhat;
      rhat += vn[n - 1];
      if (rhat >= (1ULL << 32)) break;
    }
    int64_t borrow = 0, t;
    for (int i = 0; i < n; ++i) {
      const uint64_t p = qhat * vn[i];
      t = (int64_t)un[i + j] - borrow - (int64_t)(p & 0xFFFFFFFFULL);
      un[i + j] = (uint32_t)t;
      borrow = (int64_t)(p >> 32) - (t >> 32);
    }
    t = (int64_t)un[j + n] - borrow;
    un[j + n] = (uint32_t)t;
    if (t < 0) {  // qhat was one too large: add the divisor back
      --qhat;
      uint64_t c = 0;
      for (int i = 0; i < n; ++i) {
        c += (uint64_t)un[i + j] + vn[i];
        un[i + j] = (uint32_t)c;
        c >>= 32;
      }
      un[j + n] += (uint32_t)c;
    }
    q[j] = (uint32_t)qhat;
  }
}
// magnitude of <= 8 limbs -> I256 bit pattern (unsigned)
DBHIP_HD I256 i256_from_mag(const uint32_t* a) {
  I256 r;
#pragma unroll
  for (int i = 0; i < 4; ++i) r.w[i] = (uint64_t)a[2 * i] | ((uint64_t)a[2 * i + 1] << 32);
  return r;
}

// i256::from_bigint (decimal.rs:1460-1487) on sign + magnitude; its quirk included: -2^255 comes back as DECIMAL_MIN
// (= -(10^76 - 1), passed in). false = None.
DBHIP_HD bool i256_from_bigint(bool neg, const uint32_t* mag, int n, const I256& decimal_max76, I256* out) {
  if (mag_len(mag, n) > 8) return false;
  uint32_t lo[8];
  for (int i = 0; i < 8; ++i) lo[i] = i < n ? mag[i] : 0u;
  const I256 v = i256_from_mag(lo);
  const bool top = (v.w[3] >> 63) != 0;  // magnitude >= 2^255
  if (i256_is_zero(v)) {
    *out = v;
    return true;
  }
  if (!neg) {
    if (top) return false;
    *out = v;
    return true;
  }
  if (!top) {
    *out = i256_neg(v);
    return true;
  }
  if (v.w[3] == (1ULL << 63) && (v.w[0] | v.w[1] | v.w[2]) == 0) {
    *out = i256_neg(decimal_max76);
    return true;
  }
  return false;
}

DBHIP_HD I256 i256_mul_lo(const I256& a, const I256& b) {  // wrapping (ethnum release-mode `*`)
  uint32_t x[8], y[8], p[16];
  mag_from_i256(a, x);
  mag_from_i256(b, y);
  mag_mul(x, 8, y, 8, p);  // the low 256 bits of the unsigned product are those of the two's complement product
  return i256_from_mag(p);
}
// checked_mul within a `bits`-wide signed type: false on overflow
DBHIP_HD bool i256_checked_mul(const I256& a, const I256& b, int bits, I256* out) {
  const bool neg = i256_is_neg(a) != i256_is_neg(b);
  uint32_t x[8], y[8], p[16];
  mag_from_i256(i256_abs(a), x);
  mag_from_i256(i256_abs(b), y);
  mag_mul(x, 8, y, 8, p);
  if (mag_len(p, 16) > 8) return false;
  const I256 m = i256_from_mag(p);
  const I256 r = neg ? i256_neg(m) : m;
  // the magnitude must be < 2^(bits-1), or == 2^(bits-1) for a negative product
  if (i256_is_zero(m)) {
    *out = m;
    return true;
  }
  if (i256_is_neg(r) != neg) return false;  // magnitude >= 2^255 (and not exactly -2^255)
  if (!i256_fits(r, bits)) return false;
  *out = r;
  return true;
}
// truncating signed division (Rust `/`), b != 0; MIN / -1 wraps
DBHIP_HD I256 i256_div(const I256& a, const I256& b) {
  const bool neg = i256_is_neg(a) != i256_is_neg(b);
  uint32_t x[8], y[8], q[8];
  mag_from_i256(i256_abs(a), x);
  mag_from_i256(i256_abs(b), y);
  mag_div(x, 8, y, 8, q);
  const I256 m = i256_from_mag(q);
  return neg ? i256_neg(m) : m;
}
DBHIP_HD I256 i256_half(const I256& b) {  // b / 2, truncating toward zero
  I256 m = i256_abs(b);
  m.w[0] = (m.w[0] >> 1) | (m.w[1] << 63);
  m.w[1] = (m.w[1] >> 1) | (m.w[2] << 63);
  m.w[2] = (m.w[2] >> 1) | (m.w[3] << 63);
  m.w[3] >>= 1;
  return i256_is_neg(b) ? i256_neg(m) : m;
}
DBHIP_HD bool i256_out_of_range(const I256& v, const I256& mx) { return i256_cmp(v, mx) > 0 || i256_cmp(v, i256_neg(mx)) < 0; }

// ---- binary_decimal in T = i256 -----------------------------------------------------------------------------------------
struct Dec256Op {
  int op;                    // DBHIP_OP_PLUS / MINUS / MULTIPLY / DIVIDE (0..3)
  int a_check, b_check;      // operand conversion multiplies by a_mul / b_mul and range-checks against a_max / b_max
  int overflow;              // result precision == 76
  int scale_mul;             // multiply: sa + sb - sr; divide: sb + sr - sa
  I256 a_mul, a_max, b_mul, b_max;
  I256 ret_max;              // 10^ret_precision - 1
  I256 max76;                // 10^76 - 1 (i256::DECIMAL_MAX)
  I256 div, half;            // multiply: 10^scale_mul and its half
  uint32_t mul_mag[10];      // divide: 10^scale_mul (up to 10^82: 273 bits)
  int mul_len;
};

DBHIP_HD bool dec256_convert(const I256& x, int check, const I256& mul, const I256& mx, I256* out) {
  if (!check) {
    *out = x;
    return true;
  }
  I256 r;
  if (!i256_checked_mul(x, mul, 256, &r) || i256_out_of_range(r, mx)) return false;
  *out = r;
  return true;
}

// one row; false = the row raises (the caller stores 1, like the reference's builders)
DBHIP_HD bool dec256_row(const Dec256Op& p, const I256& xv, const I256& yv, I256* out) {
  I256 x, y;
  bool ok = dec256_convert(xv, p.a_check, p.a_mul, p.a_max, &x);
  ok = dec256_convert(yv, p.b_check, p.b_mul, p.b_max, &y) && ok;
  if (!ok) return false;
  const bool same = i256_is_neg(x) == i256_is_neg(y);
  if (p.op == 0 || p.op == 1) {
    const I256 t = p.op == 0 ? i256_add(x, y) : i256_sub(x, y);
    if (p.overflow && i256_out_of_range(t, p.ret_max)) return false;
    *out = t;
    return true;
  }
  if (p.op == 2) {
    if (p.scale_mul == 0) {
      *out = i256_mul_lo(x, y);
      return true;
    }
    I256 prod;
    if (!p.overflow) {  // decimal.rs:1349-1356: everything wraps
      prod = i256_mul_lo(x, y);
    } else if (!i256_checked_mul(x, y, 256, &prod)) {  // BigInt fallback :1367-1376
      uint32_t a[8], b[8], pr[I256_MAG_MAX], h[8], d[8], q[I256_MAG_MAX];
      mag_from_i256(i256_abs(x), a);
      mag_from_i256(i256_abs(y), b);
      mag_mul(a, 8, b, 8, pr);
      for (int i = 16; i < I256_MAG_MAX; ++i) pr[i] = 0;
      mag_from_i256(p.half, h);
      mag_add_into(pr, 17, h, 8);  // |a b| >= 2^255 > half: a*b +- half keeps the sign of the product
      mag_from_i256(p.div, d);
      mag_div(pr, 17, d, 8, q);
      return i256_from_bigint(!same, q, 17, p.max76, out);
    }
    *out = i256_div(same ? i256_add(prod, p.half) : i256_sub(prod, p.half), p.div);
    return true;
  }
  // divide: binary_decimal :212-243, do_round_div decimal.rs:1378-1404
  if (i256_is_zero(y)) return false;
  const I256 hb = i256_half(y);
  if (p.scale_mul < 76) {
    I256 xm;
    const I256 mul = i256_from_mag(p.mul_mag);  // < 10^76 fits
    if (i256_checked_mul(x, mul, 256, &xm)) {
      *out = i256_div(same ? i256_add(xm, hb) : i256_sub(xm, hb), y);
      return true;
    }
  }
  uint32_t a[8], num[I256_MAG_MAX], h[8], d[8], q[I256_MAG_MAX];
  mag_from_i256(i256_abs(x), a);
  mag_mul(a, 8, p.mul_mag, 10, num);  // 18 limbs
  num[18] = num[19] = 0;
  mag_from_i256(i256_abs(hb), h);
  mag_add_into(num, 19, h, 8);        // a*mul and +-(b/2) pull the same way (see `same`)
  mag_from_i256(i256_abs(y), d);
  mag_div(num, 19, d, 8, q);
  return i256_from_bigint(!same, q, 19, p.max76, out);
}

// ---- DecimalCmp across DecimalSizes (comparison.rs:326-441) in a T of `bits` ---------------------------------------------
DBHIP_HD int dec256_cmp3(const I256& av, const I256& bv, const I256& fa, const I256& fb, bool fa_one, bool fb_one, bool same_f,
                         int bits) {
  const I256 a = i256_wrap(av, bits), b = i256_wrap(bv, bits);  // as_decimal::<T>()
  if (same_f) return i256_cmp(a, b);
  const I256 zero = i256_from_u64(0);
  const int sa = i256_cmp(a, zero), sb = i256_cmp(b, zero);
  if (sa != sb) return i256_cmp(a, b);
  I256 x = a, y = b;
  if (!fa_one && !i256_checked_mul(a, fa, bits, &x)) return sa > 0 ? 1 : -1;
  if (!fb_one && !i256_checked_mul(b, fb, bits, &y)) return sb > 0 ? -1 : 1;
  return i256_cmp(x, y);
}

// ---- to_decimal(decimal | integer) (cast.rs:470-483, 701-753, 790-1035) --------------------------------------------------
struct Dec256Cast {
  int src_is_decimal;
  int mode;          // 0 copy (no check), 1 range check only, 2 multiply by `factor` then check, 3 scale reduction
  int cbits, dbits;  // width of the type the reference computes in; width of the destination storage class
  int rounding;      // rounding_mode && scale_diff != 0
  I256 factor;       // 10^|scale difference|
  I256 factor_m1;    // 10^(scale_diff - 1) (rounding)
  I256 mx;           // 10^dst_precision - 1
  I256 mxs;          // 10^from_scale - 1 (int_part_is_zero)
};

DBHIP_HD bool dec256_cast_row(const Dec256Cast& c, const I256& x, I256* out) {
  I256 y;
  switch (c.mode) {
    case 0:
      y = x;
      break;
    case 1:
      if (i256_out_of_range(x, c.mx)) return false;
      y = x;
      break;
    case 2:
      if (!c.src_is_decimal && !i256_fits(x, c.cbits)) return false;  // T::from_i128(x): None when x does not fit T
      if (!i256_checked_mul(x, c.factor, c.cbits, &y) || i256_out_of_range(y, c.mx)) return false;
      break;
    default: {  // decimal_scale_reduction + get_round_val
      if (c.rounding) {
        const I256 q2 = i256_div(x, c.factor_m1), ten = i256_from_u64(10);
        const I256 q = i256_div(q2, ten);  // trunc(trunc(x / 10^(d-1)) / 10) == trunc(x / 10^d)
        const I256 m = i256_sub(q2, i256_mul_lo(q, ten));
        y = q;
        if (i256_cmp(m, i256_from_i64(5)) >= 0) y = i256_add(q, i256_from_i64(1));
        else if (i256_cmp(m, i256_from_i64(-5)) <= 0) y = i256_add(q, i256_from_i64(-1));
      } else {
        y = i256_div(x, c.factor);
      }
      if (!i256_fits(y, c.cbits)) return false;  // checked_add
      const bool int_part_zero = !i256_is_neg(x) ? i256_cmp(x, c.mxs) <= 0 : i256_cmp(x, i256_neg(c.mxs)) >= 0;
      if (i256_out_of_range(y, c.mx) || (i256_is_zero(y) && !int_part_zero)) return false;
    } break;
  }
  *out = i256_wrap(y, c.dbits);
  return true;
}

// host: 10^k as an I256 (k <= 76) and as a 32-bit-limb magnitude (k <= 82)
inline I256 i256_pow10(int k) {
  I256 r = i256_from_u64(1);
  const I256 ten = i256_from_u64(10);
  for (int i = 0; i < k; ++i) r = i256_mul_lo(r, ten);
  return r;
}
inline int mag_pow10(int k, uint32_t* out, int cap) {
  for (int i = 0; i < cap; ++i) out[i] = 0;
  out[0] = 1;
  for (int s = 0; s < k; ++s) {
    uint64_t c = 0;
    for (int i = 0; i < cap; ++i) {
      c += (uint64_t)out[i] * 10u;
      out[i] = (uint32_t)c;
      c >>= 32;
    }
  }
  return mag_len(out, cap);
}


// ---- host: decode one call node into the row functions' parameters --------------------------------------------------------
struct DecSize {
  int p, s;
};
inline int dec_storage_bits(int precision) { return precision <= 18 ? 64 : (precision <= 38 ? 128 : 256); }

// ArithmeticOp::result_size (arithmetic.rs:80-139) for operands of any precision up to 76: the result is clamped to 38 digits
// when both operands have at most 38, to 76 otherwise. false = no such function (DecimalSize::new fails).
inline bool dec_result_size(int op, DecSize a, DecSize b, DecSize* left, DecSize* right, DecSize* ret) {
  int precision, scale;
  const int la = a.p - a.s, lb = b.p - b.s;
  auto mn = [](int x, int y) { return x < y ? x : y; };
  auto mx = [](int x, int y) { return x > y ? x : y; };
  switch (op) {
    case 2: scale = mn(a.s + b.s, mx(mx(a.s, b.s), 12)); precision = la + lb + scale; break;
    case 3: scale = mx(a.s, mn(a.s + 6, 12)); precision = la + b.s + scale; break;
    case 0: case 1: scale = mx(a.s, b.s); precision = mx(la, lb) + scale + 1; break;
    default: return false;
  }
  precision = mn(precision, (a.p <= 38 && b.p <= 38) ? 38 : 76);
  if (precision < 1 || scale > precision) return false;
  *ret = {precision, scale};
  if (op == 2) {
    *left = {precision, a.s};
    *right = {precision, b.s};
  } else if (op == 3) {
    const int pp = mx(precision, mx(a.p, b.p));
    *left = {pp, a.s};
    *right = {pp, b.s};
  } else {
    *left = *ret;
    *right = *ret;
  }
  return left->s <= left->p && right->s <= right->p;
}

// operands (decimal of size a / b, or an integer with its decimal properties and is_dec = false) -> Dec256Op; only for
// results beyond 38 digits (T = i256)
inline bool dec256_make_op(int op, bool a_dec, DecSize a, bool b_dec, DecSize b, Dec256Op* out, DecSize* ret) {
  DecSize l, r;
  if (!dec_result_size(op, a, b, &l, &r, ret) || ret->p <= 38) return false;
  Dec256Op& p = *out;
  p.op = op;
  // convert_to_decimal: a decimal of the bound scale passes through; an integer bound to scale 0 passes through
  p.a_check = a_dec ? (a.s != l.s) : (l.s != 0);
  p.b_check = b_dec ? (b.s != r.s) : (r.s != 0);
  const I256 one = i256_from_u64(1);
  p.a_mul = i256_pow10(a_dec ? l.s - a.s : l.s);
  p.b_mul = i256_pow10(b_dec ? r.s - b.s : r.s);
  p.a_max = i256_sub(i256_pow10(l.p), one);
  p.b_max = i256_sub(i256_pow10(r.p), one);
  p.ret_max = i256_sub(i256_pow10(ret->p), one);
  p.max76 = i256_sub(i256_pow10(76), one);
  p.overflow = ret->p == 76;
  p.scale_mul = op == 2 ? a.s + b.s - ret->s : (op == 3 ? b.s + ret->s - a.s : 0);
  if (p.scale_mul < 0 || p.scale_mul > 82) return false;
  p.div = i256_pow10(op == 2 ? p.scale_mul : 0);
  p.half = i256_half(p.div);
  p.mul_len = mag_pow10(op == 3 ? p.scale_mul : 0, p.mul_mag, 10);
  return true;
}

// decimal_to_decimal / integer_to_decimal (cast.rs:701-753, 981-1035): src_bits = 0 for an integer source
inline bool dec256_make_cast(int src_bits, DecSize from, DecSize to, bool rounding_mode, Dec256Cast* out) {
  if (to.p < 1 || to.p > 76 || to.s > to.p) return false;
  Dec256Cast& c = *out;
  const I256 one = i256_from_u64(1);
  c.src_is_decimal = src_bits != 0;
  c.dbits = dec_storage_bits(to.p);
  c.mx = i256_sub(i256_pow10(to.p), one);
  c.rounding = 0;
  c.factor = c.factor_m1 = one;
  c.mxs = i256_from_u64(0);
  if (!src_bits) {
    c.cbits = c.dbits;
    c.mode = to.s == 0 ? 0 : 2;
    c.factor = i256_pow10(to.s);
    return true;
  }
  if (from.p < 1 || from.p > 76 || from.s > from.p) return false;
  const bool expand = src_bits == 64 || (src_bits == 128 && c.dbits >= 128) || (src_bits == 256 && c.dbits == 256);
  c.cbits = expand ? c.dbits : src_bits;
  if (expand && from.s == to.s && from.p <= to.p) c.mode = 0;
  else if (from.s == to.s) c.mode = 1;
  else if (to.s > from.s) {
    c.mode = 2;
    c.factor = i256_pow10(to.s - from.s);
  } else {
    const int d = from.s - to.s;
    c.mode = 3;
    c.factor = i256_pow10(d);
    c.factor_m1 = i256_pow10(d - 1);
    c.rounding = rounding_mode ? 1 : 0;
    c.mxs = i256_sub(i256_pow10(from.s), one);
  }
  return true;
}

}  // namespace dbhip

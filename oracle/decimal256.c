/*
 * decimal256.c — CPU restatement of the reference's Decimal256 (i256) class and of the decimal functions that span
 * storage classes: binary arithmetic in T = i256, unary minus, decimal -> decimal / integer -> decimal CAST, comparison of
 * decimals of different DecimalSize.  TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * Reference (src/query):
 *   functions/src/scalars/decimal/src/arithmetic.rs:80-139   result_size (clamp 38 / 76)
 *                                                  :190-316  binary_decimal
 *                                                  :514-590  unary minus
 *   functions/src/scalars/decimal/src/cast.rs:701-753        integer_to_decimal
 *                                            :790-1035       get_round_val / shrink / scale_reduction / expand /
 *                                                            decimal_to_decimal
 *   functions/src/scalars/decimal/src/comparison.rs:326-441  DecimalCmp, calc_size, CmpOp::compare
 *   expression/src/types/decimal.rs:1343-1404                i256 do_round_mul / do_round_div (checked path, BigInt fallback)
 *                                  :1460-1487                from_bigint (incl. "-2^255 -> DECIMAL_MIN")
 *   i256 is ethnum::I256: + - * wrap in release builds (Cargo.toml:577 overflow-checks = false), / truncates.
 *
 * Everything is computed in ONE wide signed integer type (640 bits, two's complement) and narrowed with wrap(bits) exactly
 * where the reference's fixed-width type would wrap; the division is a bit-serial shift-subtract — deliberately nothing in
 * common with the device's 32-bit-limb long division (databend_amd/csrc/dev_i256.h).
 * Pinned by: the Decimal(76,x) cases of the reference's arithmetic.txt and decimal_to_decimal_cast.txt (tests/golden/),
 * and tests/dec256_ref.py (Python big integers) on random operands.
 */
#include <stdlib.h>
#include <string.h>

#include "oracle.h"

#define WL 10 /* limbs */
typedef struct { uint64_t w[WL]; } W;
typedef __int128 i128;
typedef unsigned __int128 u128;

static W w_zero(void) { W r; memset(&r, 0, sizeof r); return r; }
static W w_from_i64(int64_t v) { W r; r.w[0] = (uint64_t)v; for (int i = 1; i < WL; ++i) r.w[i] = v < 0 ? ~0ULL : 0; return r; }
static W w_from_u64(uint64_t v) { W r = w_zero(); r.w[0] = v; return r; }
/* sign-extends the low `bits` (64 / 128 / 256) of a little-endian limb array */
static W w_from_limbs(const uint64_t* p, int bits) {
  W r; int n = bits / 64;
  uint64_t ext = (p[n - 1] >> 63) ? ~0ULL : 0;
  for (int i = 0; i < WL; ++i) r.w[i] = i < n ? p[i] : ext;
  return r;
}
static int w_neg_p(W a) { return (int)(a.w[WL - 1] >> 63); }
static int w_is_zero(W a) { for (int i = 0; i < WL; ++i) if (a.w[i]) return 0; return 1; }
static W w_add(W a, W b) { W r; u128 c = 0; for (int i = 0; i < WL; ++i) { c += (u128)a.w[i] + b.w[i]; r.w[i] = (uint64_t)c; c >>= 64; } return r; }
static W w_not(W a) { for (int i = 0; i < WL; ++i) a.w[i] = ~a.w[i]; return a; }
static W w_negate(W a) { return w_add(w_not(a), w_from_u64(1)); }
static W w_sub(W a, W b) { return w_add(a, w_negate(b)); }
static W w_abs(W a) { return w_neg_p(a) ? w_negate(a) : a; }
static int w_ucmp(W a, W b) { for (int i = WL - 1; i >= 0; --i) if (a.w[i] != b.w[i]) return a.w[i] > b.w[i] ? 1 : -1; return 0; }
static int w_cmp(W a, W b) { int na = w_neg_p(a), nb = w_neg_p(b); if (na != nb) return na ? -1 : 1; return w_ucmp(a, b); }
static W w_mul(W a, W b) { /* low 640 bits */
  W r = w_zero();
  for (int i = 0; i < WL; ++i) { u128 c = 0; for (int j = 0; i + j < WL; ++j) { c += (u128)a.w[i] * b.w[j] + r.w[i + j]; r.w[i + j] = (uint64_t)c; c >>= 64; } }
  return r;
}
/* truncating signed division (Rust `/`); b != 0 */
static W w_div(W a, W b) {
  int neg = w_neg_p(a) != w_neg_p(b);
  W n = w_abs(a), d = w_abs(b), q = w_zero(), r = w_zero();
  for (int bit = WL * 64 - 1; bit >= 0; --bit) {
    for (int i = WL - 1; i > 0; --i) r.w[i] = (r.w[i] << 1) | (r.w[i - 1] >> 63);
    r.w[0] = (r.w[0] << 1) | ((n.w[bit >> 6] >> (bit & 63)) & 1);
    if (w_ucmp(r, d) >= 0) { r = w_sub(r, d); q.w[bit >> 6] |= 1ULL << (bit & 63); }
  }
  return neg ? w_negate(q) : q;
}
static W w_rem(W a, W b) { return w_sub(a, w_mul(w_div(a, b), b)); }
static W w_pow10(int k) { W r = w_from_u64(1), ten = w_from_u64(10); while (k-- > 0) r = w_mul(r, ten); return r; }
/* value as a `bits`-wide two's complement integer (sign-extended back into W) */
static W w_wrap(W a, int bits) { return w_from_limbs(a.w, bits); }
static int w_fits(W a, int bits) { return w_cmp(w_wrap(a, bits), a) == 0; }
static void w_store(W a, void* out, int bits) { memcpy(out, a.w, (size_t)bits / 8); }
static W w_half(W b) { return w_div(b, w_from_u64(2)); }

static int storage_bits(int p) { return p <= 18 ? 64 : (p <= 38 ? 128 : 256); }
static int bits_of_type(int type) { return type == ORC_T_DEC64 ? 64 : (type == ORC_T_DEC128 ? 128 : (type == ORC_T_DEC256 ? 256 : 0)); }
static int is_dec(int type) { return bits_of_type(type) != 0; }
static int valid_at(const orc_col* c, int64_t i) {
  if (!c->validity) return 1;
  int64_t j = c->validity_offset + (c->is_scalar ? 0 : i);
  return (c->validity[j >> 3] >> (j & 7)) & 1;
}
static W load_any(const orc_col* c, int64_t i) {
  int64_t j = c->is_scalar ? 0 : i;
  switch (c->type) {
    case ORC_T_DEC64: case ORC_T_I64: return w_from_i64(((const int64_t*)c->data)[j]);
    case ORC_T_DEC128: return w_from_limbs((const uint64_t*)c->data + 2 * j, 128);
    case ORC_T_DEC256: return w_from_limbs((const uint64_t*)c->data + 4 * j, 256);
    case ORC_T_I8: return w_from_i64(((const int8_t*)c->data)[j]);
    case ORC_T_I16: return w_from_i64(((const int16_t*)c->data)[j]);
    case ORC_T_I32: return w_from_i64(((const int32_t*)c->data)[j]);
    case ORC_T_U8: return w_from_u64(((const uint8_t*)c->data)[j]);
    case ORC_T_U16: return w_from_u64(((const uint16_t*)c->data)[j]);
    case ORC_T_U32: return w_from_u64(((const uint32_t*)c->data)[j]);
    default: return w_from_u64(((const uint64_t*)c->data)[j]);
  }
}
typedef struct { int p, s; } dsz;
static int props(const orc_col* c, dsz* o) {
  switch (c->type) {
    case ORC_T_DEC64: case ORC_T_DEC128: case ORC_T_DEC256: o->p = c->precision; o->s = c->scale; return o->p >= 1 && o->p <= 76 && o->s <= o->p;
    case ORC_T_I8: case ORC_T_U8: o->p = 3; o->s = 0; return 1;
    case ORC_T_I16: case ORC_T_U16: o->p = 5; o->s = 0; return 1;
    case ORC_T_I32: case ORC_T_U32: o->p = 10; o->s = 0; return 1;
    case ORC_T_I64: o->p = 19; o->s = 0; return 1;
    case ORC_T_U64: o->p = 20; o->s = 0; return 1;
  }
  return 0;
}
static int imin(int a, int b) { return a < b ? a : b; }
static int imax(int a, int b) { return a > b ? a : b; }
static int out_of_range(W v, int precision) { W mx = w_sub(w_pow10(precision), w_from_u64(1)); return w_cmp(v, mx) > 0 || w_cmp(v, w_negate(mx)) < 0; }
static void raise_row(const orc_col* a, const orc_col* b, int64_t i, uint8_t* err, uint64_t* cnt) {
  if (!valid_at(a, i) || (b && !valid_at(b, i))) return; /* NULL rows never raise (function.rs:536-543) */
  if (err) err[i >> 3] &= (uint8_t)~(1u << (i & 7));
  if (cnt) ++*cnt;
}

/* ArithmeticOp::result_size arithmetic.rs:80-139, with the 38 / 76 clamp */
int orc_dec_result_size(int op, int lp, int ls, int rp, int rs, int* lp_o, int* ls_o, int* rp_o, int* rs_o, int* out_p, int* out_s) {
  int precision, scale, la = lp - ls, lb = rp - rs;
  switch (op) {
    case ORC_OP_MULTIPLY: scale = imin(ls + rs, imax(imax(ls, rs), 12)); precision = la + lb + scale; break;
    case ORC_OP_DIVIDE: scale = imax(ls, imin(ls + 6, 12)); precision = la + rs + scale; break;
    case ORC_OP_PLUS: case ORC_OP_MINUS: scale = imax(ls, rs); precision = imax(la, lb) + scale + 1; break;
    default: return 1;
  }
  precision = imin(precision, (lp <= 38 && rp <= 38) ? 38 : 76);
  if (precision < 1 || scale > precision) return 1;
  int l_p, l_s, r_p, r_s;
  if (op == ORC_OP_MULTIPLY) { l_p = precision; l_s = ls; r_p = precision; r_s = rs; }
  else if (op == ORC_OP_DIVIDE) { int pp = imax(precision, imax(lp, rp)); l_p = pp; l_s = ls; r_p = pp; r_s = rs; }
  else { l_p = r_p = precision; l_s = r_s = scale; }
  if (l_s > l_p || r_s > r_p) return 1;
  if (lp_o) { *lp_o = l_p; *ls_o = l_s; *rp_o = r_p; *rs_o = r_s; }
  *out_p = precision; *out_s = scale;
  return 0;
}

/* i256::from_bigint decimal.rs:1460-1487 -> 0 = None */
static int from_bigint(W v, W* out) {
  W mag = w_abs(v), lim = w_zero();
  lim.w[3] = 1ULL << 63; /* 2^255 */
  for (int i = 4; i < WL; ++i) if (mag.w[i]) return 0; /* more than four u64 digits */
  int c = w_ucmp(mag, lim);
  if (!w_neg_p(v)) { if (c >= 0) return 0; *out = v; return 1; }
  if (c < 0) { *out = v; return 1; }
  if (c == 0) { *out = w_negate(w_sub(w_pow10(76), w_from_u64(1))); return 1; } /* Some(i256::DECIMAL_MIN) */
  return 0;
}

/* convert_to_decimal arithmetic.rs:141-153 (integer_to_decimal / decimal_expand_cast), in T of `bits` */
static int convert_operand(W x, int dec, int from_s, dsz to, int bits, W* out) {
  if (!dec) {
    if (to.s == 0) { *out = w_wrap(x, bits); return 1; }
    if (!w_fits(x, bits)) return 0;
    W r = w_mul(x, w_pow10(to.s));
    if (!w_fits(r, bits) || out_of_range(r, to.p)) return 0;
    *out = r; return 1;
  }
  if (from_s == to.s) { *out = w_wrap(x, bits); return 1; }
  W r = w_mul(w_wrap(x, bits), w_pow10(to.s - from_s));
  if (!w_fits(r, bits) || out_of_range(r, to.p)) return 0;
  *out = r; return 1;
}

/* binary_decimal with T = i256 (the i64 / i128 classes are oracle.c's orc_decimal_arith, which routes here when the result
 * precision exceeds 38). Operands of any storage class; out = i256[n]. */
int orc_decimal256_arith(int op, const orc_col* lhs, const orc_col* rhs, int64_t n, int out_p, int out_s, void* out,
                         uint8_t* err, uint64_t* err_count) {
  dsz a, b, l, r, ret;
  if (!props(lhs, &a) || !props(rhs, &b)) return 1;
  if (orc_dec_result_size(op, a.p, a.s, b.p, b.s, &l.p, &l.s, &r.p, &r.s, &ret.p, &ret.s)) return 1;
  if (ret.p <= 38 || ret.p != out_p || ret.s != out_s) return 1;
  const int bits = 256, overflow = ret.p == 76;
  int a_dec = is_dec(lhs->type), b_dec = is_dec(rhs->type);
  if (err) memset(err, 0xFF, (size_t)((n + 31) / 32) * 4);
  for (int64_t i = 0; i < n; ++i) {
    W x, y, res = w_from_u64(1);
    int ok = convert_operand(load_any(lhs, i), a_dec, a.s, l, bits, &x);
    ok = convert_operand(load_any(rhs, i), b_dec, b.s, r, bits, &y) && ok;
    int same = ok && (w_neg_p(x) == w_neg_p(y));
    if (ok) switch (op) {
      case ORC_OP_PLUS: case ORC_OP_MINUS: {
        W t = w_wrap(op == ORC_OP_PLUS ? w_add(x, y) : w_sub(x, y), bits);
        if (overflow && out_of_range(t, ret.p)) ok = 0;
        res = t;
      } break;
      case ORC_OP_MULTIPLY: {
        int sm = a.s + b.s - ret.s;
        if (sm == 0) { res = w_wrap(w_mul(x, y), bits); break; }
        W div = w_pow10(sm), half = w_half(div), exact = w_mul(x, y);
        if (!overflow) { /* decimal.rs:1349-1356 */
          W p = w_wrap(exact, bits);
          res = w_div(w_wrap(same ? w_add(p, half) : w_sub(p, half), bits), div);
        } else if (w_fits(exact, bits)) { /* checked_mul succeeded :1359-1365 */
          res = w_div(w_wrap(same ? w_add(exact, half) : w_sub(exact, half), bits), div);
        } else { /* BigInt fallback :1367-1376 */
          if (!from_bigint(w_div(same ? w_add(exact, half) : w_sub(exact, half), div), &res)) ok = 0;
        }
      } break;
      default: { /* divide: binary_decimal :212-243, do_round_div decimal.rs:1378-1404 */
        int ms = b.s + ret.s - a.s;
        if (w_is_zero(y)) { ok = 0; break; }
        W hb = w_half(y), xm = w_mul(x, w_pow10(ms));
        if (ms < 76 && w_fits(xm, bits)) {
          res = w_wrap(w_div(w_wrap(same ? w_add(xm, hb) : w_sub(xm, hb), bits), y), bits);
        } else {
          if (!from_bigint(w_div(same ? w_add(xm, hb) : w_sub(xm, hb), y), &res)) ok = 0;
        }
      } break;
    }
    if (!ok) { raise_row(lhs, rhs, i, err, err_count); res = w_from_u64(1); }
    w_store(res, (uint8_t*)out + 32 * i, 256);
  }
  return 0;
}

/* unary minus (arithmetic.rs:514-590): `-t` in the column's storage class */
int orc_decimal_neg(const orc_col* src, int64_t n, void* out) {
  int bits = bits_of_type(src->type);
  if (!bits) return 1;
  for (int64_t i = 0; i < n; ++i) w_store(w_wrap(w_negate(load_any(src, i)), bits), (uint8_t*)out + (bits / 8) * i, bits);
  return 0;
}

/* DecimalCmp::eval (comparison.rs:326-441), any two storage classes / DecimalSizes */
int orc_cmp_decimal_any(int op, const orc_col* lhs, const orc_col* rhs, int64_t n, uint8_t* out) {
  if (!is_dec(lhs->type) || !is_dec(rhs->type)) return 1;
  int scale = imax(lhs->scale, rhs->scale);
  int precision = imax(lhs->precision - lhs->scale, rhs->precision - rhs->scale) + scale;
  precision = imin(precision, (lhs->precision <= 38 && rhs->precision <= 38) ? 38 : 76);
  int bits = storage_bits(precision);
  W fa = w_pow10(scale - lhs->scale), fb = w_pow10(scale - rhs->scale), one = w_from_u64(1), zero = w_zero();
  memset(out, 0, (size_t)((n + 7) / 8));
  for (int64_t i = 0; i < n; ++i) {
    W a = w_wrap(load_any(lhs, i), bits), b = w_wrap(load_any(rhs, i), bits); /* as_decimal::<T>() */
    int c;
    if (w_cmp(fa, fb) == 0) c = w_cmp(a, b);
    else {
      int sa = w_cmp(a, zero), sb = w_cmp(b, zero);
      if (sa != sb) c = w_cmp(a, b);
      else {
        W x = a, y = b; int done = 0;
        if (w_cmp(fa, one) != 0) { x = w_mul(a, fa); if (!w_fits(x, bits)) { c = sa > 0 ? 1 : -1; done = 1; } }
        if (!done && w_cmp(fb, one) != 0) { y = w_mul(b, fb); if (!w_fits(y, bits)) { c = sb > 0 ? -1 : 1; done = 1; } }
        if (!done) c = w_cmp(x, y);
      }
    }
    int r;
    switch (op) {
      case ORC_CMP_EQ: r = c == 0; break;
      case ORC_CMP_NOTEQ: r = c != 0; break;
      case ORC_CMP_LT: r = c < 0; break;
      case ORC_CMP_LTE: r = c <= 0; break;
      case ORC_CMP_GT: r = c > 0; break;
      default: r = c >= 0; break;
    }
    if (r) out[i >> 3] |= (uint8_t)(1u << (i & 7));
  }
  return 0;
}

/* decimal_scale_reduction + get_round_val (cast.rs:790-808,884-899), in a type of `bits` */
static int scale_reduction(W x, int dst_p, int from_s, int scale_diff, int rounding_mode, int bits, W* out) {
  W q = w_div(x, w_pow10(scale_diff)), y = q;
  if (rounding_mode && scale_diff != 0) {
    W m = w_rem(w_div(x, w_pow10(scale_diff - 1)), w_from_u64(10));
    if (w_cmp(m, w_from_i64(5)) >= 0) y = w_add(q, w_from_i64(1));
    else if (w_cmp(m, w_from_i64(-5)) <= 0) y = w_add(q, w_from_i64(-1));
  }
  if (!w_fits(y, bits)) return 0; /* checked_add */
  W mxs = w_sub(w_pow10(from_s), w_from_u64(1));
  int int_part_zero = !w_neg_p(x) ? w_cmp(x, mxs) <= 0 : w_cmp(x, w_negate(mxs)) >= 0;
  if (out_of_range(y, dst_p) || (w_is_zero(y) && !int_part_zero)) return 0;
  *out = y; return 1;
}

/* to_decimal / try_to_decimal for decimal and integer sources (cast.rs:470-483 convert_to_decimal): row errors are
 * "Decimal overflow" (value 1, bit of `bitmap` cleared), or NULL for is_try (bitmap = result validity). The destination is
 * stored by its precision (DecimalDataType::from(size)). */
int orc_decimal_cast(const orc_col* src, int dst_p, int dst_s, int is_try, int rounding_mode, int64_t n, void* out,
                     uint8_t* bitmap, uint64_t* n_errors) {
  if (dst_p < 1 || dst_p > 76 || dst_s > dst_p) return 1;
  int dbits = storage_bits(dst_p), sbits = bits_of_type(src->type);
  dsz from;
  if (!props(src, &from)) return 1;
  if (bitmap) memset(bitmap, 0xFF, (size_t)((n + 63) / 64) * 8);
  for (int64_t i = 0; i < n; ++i) {
    W x = load_any(src, i), y = w_from_u64(1);
    int ok = 1;
    if (!sbits) { /* integer_to_decimal<T = destination class> cast.rs:701-753 */
      if (dst_s == 0) y = w_wrap(x, dbits);
      else {
        if (!w_fits(x, dbits)) ok = 0;
        else { y = w_mul(x, w_pow10(dst_s)); if (!w_fits(y, dbits) || out_of_range(y, dst_p)) ok = 0; }
      }
    } else {
      int expand = sbits == 64 || (sbits == 128 && dbits >= 128) || (sbits == 256 && dbits == 256);
      int cbits = expand ? dbits : sbits;
      if (expand && from.s == dst_s && from.p <= dst_p) y = w_wrap(x, dbits);              /* faster path :909-923 */
      else if (dst_s == from.s) { if (out_of_range(x, dst_p)) ok = 0; else y = w_wrap(x, dbits); }
      else if (dst_s > from.s) {
        y = w_mul(x, w_pow10(dst_s - from.s));
        if (!w_fits(y, cbits) || out_of_range(y, dst_p)) ok = 0; else y = w_wrap(y, dbits);
      } else {
        ok = scale_reduction(x, dst_p, from.s, from.s - dst_s, rounding_mode, cbits, &y);
        if (ok) y = w_wrap(y, dbits);
      }
    }
    int valid = valid_at(src, i);
    if (!ok) {
      y = w_from_u64(1);
      if (is_try) { if (bitmap) bitmap[i >> 3] &= (uint8_t)~(1u << (i & 7)); }
      else if (valid) { if (bitmap) bitmap[i >> 3] &= (uint8_t)~(1u << (i & 7)); if (n_errors) ++*n_errors; }
    }
    if (is_try && !valid && bitmap) bitmap[i >> 3] &= (uint8_t)~(1u << (i & 7));
    w_store(y, (uint8_t*)out + (dbits / 8) * i, dbits);
  }
  return 0;
}

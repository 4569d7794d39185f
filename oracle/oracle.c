/*
 * oracle.c — CPU restatement of the reference algorithms (TEST INFRASTRUCTURE,
 * see oracle.h). Plain scalar C, one row at a time, written to mirror the
 * reference's structure rather than to be fast. gcc -O2.
 */
#define _GNU_SOURCE
#include "oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

typedef __int128 i128;
typedef unsigned __int128 u128;

/* ------------------------------------------------------------------------ */
/* typed scalar values with Rust `as` cast semantics                          */
/* ------------------------------------------------------------------------ */
typedef struct {
  int cls; /* 0 signed, 1 unsigned, 2 float */
  int bits;
  int64_t i;
  uint64_t u;
  double f; /* f32 values are kept exactly in a double */
} val;

static int t_cls(int t) {
  switch (t) {
    case ORC_T_I8: case ORC_T_I16: case ORC_T_I32: case ORC_T_I64: case ORC_T_DATE: case ORC_T_TIMESTAMP:
    case ORC_T_DEC64: return 0;
    case ORC_T_U8: case ORC_T_U16: case ORC_T_U32: case ORC_T_U64: return 1;
    case ORC_T_F32: case ORC_T_F64: return 2;
  }
  return -1;
}
static int t_bits(int t) {
  switch (t) {
    case ORC_T_I8: case ORC_T_U8: return 8;
    case ORC_T_I16: case ORC_T_U16: return 16;
    case ORC_T_I32: case ORC_T_U32: case ORC_T_F32: case ORC_T_DATE: return 32;
    default: return 64;
  }
}
static int t_size(int t) {
  if (t == ORC_T_DEC256) return 32;
  if (t == ORC_T_DEC128 || t == ORC_T_STRING) return 16;
  return t_bits(t) / 8;
}
static int bit_get(const uint8_t* bm, int64_t i) { return (bm[i >> 3] >> (i & 7)) & 1; }
static int col_valid(const orc_col* c, int64_t i) {
  return !c->validity || bit_get(c->validity, c->validity_offset + (c->is_scalar ? 0 : i));
}

static val load_val(const orc_col* c, int64_t i) {
  int64_t j = c->is_scalar ? 0 : i;
  val v; memset(&v, 0, sizeof v);
  v.cls = t_cls(c->type); v.bits = t_bits(c->type);
  switch (c->type) {
    case ORC_T_I8: v.i = ((const int8_t*)c->data)[j]; break;
    case ORC_T_I16: v.i = ((const int16_t*)c->data)[j]; break;
    case ORC_T_I32: case ORC_T_DATE: v.i = ((const int32_t*)c->data)[j]; break;
    case ORC_T_I64: case ORC_T_TIMESTAMP: case ORC_T_DEC64: v.i = ((const int64_t*)c->data)[j]; break;
    case ORC_T_U8: v.u = ((const uint8_t*)c->data)[j]; break;
    case ORC_T_U16: v.u = ((const uint16_t*)c->data)[j]; break;
    case ORC_T_U32: v.u = ((const uint32_t*)c->data)[j]; break;
    case ORC_T_U64: v.u = ((const uint64_t*)c->data)[j]; break;
    case ORC_T_F32: v.f = ((const float*)c->data)[j]; break;
    case ORC_T_F64: v.f = ((const double*)c->data)[j]; break;
  }
  return v;
}

/* Rust float -> int `as`: truncate, saturate, NaN -> 0 */
static int64_t f_to_i(double x, int bits) {
  if (x != x) return 0;
  double lo = -ldexp(1.0, bits - 1), hi = ldexp(1.0, bits - 1);
  if (x <= lo) return bits == 64 ? INT64_MIN : -((int64_t)1 << (bits - 1));
  if (x >= hi) return bits == 64 ? INT64_MAX : (((int64_t)1 << (bits - 1)) - 1);
  return (int64_t)x;
}
static uint64_t f_to_u(double x, int bits) {
  if (x != x || x <= 0.0) return 0;
  if (x >= ldexp(1.0, bits)) return bits == 64 ? UINT64_MAX : (((uint64_t)1 << bits) - 1);
  return (uint64_t)x;
}
static int64_t wrap_i(uint64_t w, int bits) {
  switch (bits) {
    case 8: return (int8_t)w;
    case 16: return (int16_t)w;
    case 32: return (int32_t)w;
    default: return (int64_t)w;
  }
}
static uint64_t wrap_u(uint64_t w, int bits) { return bits == 64 ? w : (w & (((uint64_t)1 << bits) - 1)); }

/* `v as T` */
static val cast_val(val v, int to_type) {
  val r; memset(&r, 0, sizeof r);
  r.cls = t_cls(to_type); r.bits = t_bits(to_type);
  if (r.cls == 2) {
    double d = v.cls == 2 ? v.f : (v.cls == 0 ? (double)v.i : (double)v.u);
    if (r.bits == 32) d = (double)(float)(v.cls == 2 ? v.f : (v.cls == 0 ? (float)v.i : (float)v.u));
    r.f = d;
  } else if (r.cls == 0) {
    r.i = v.cls == 2 ? f_to_i(v.f, r.bits) : wrap_i(v.cls == 0 ? (uint64_t)v.i : v.u, r.bits);
  } else {
    r.u = v.cls == 2 ? f_to_u(v.f, r.bits) : wrap_u(v.cls == 0 ? (uint64_t)v.i : v.u, r.bits);
  }
  return r;
}

static void store_val(void* out, int type, int64_t i, val v) {
  switch (type) {
    case ORC_T_I8: ((int8_t*)out)[i] = (int8_t)v.i; break;
    case ORC_T_I16: ((int16_t*)out)[i] = (int16_t)v.i; break;
    case ORC_T_I32: case ORC_T_DATE: ((int32_t*)out)[i] = (int32_t)v.i; break;
    case ORC_T_I64: case ORC_T_TIMESTAMP: case ORC_T_DEC64: ((int64_t*)out)[i] = v.i; break;
    case ORC_T_U8: ((uint8_t*)out)[i] = (uint8_t)v.u; break;
    case ORC_T_U16: ((uint16_t*)out)[i] = (uint16_t)v.u; break;
    case ORC_T_U32: ((uint32_t*)out)[i] = (uint32_t)v.u; break;
    case ORC_T_U64: ((uint64_t*)out)[i] = v.u; break;
    case ORC_T_F32: ((float*)out)[i] = (float)v.f; break;
    case ORC_T_F64: ((double*)out)[i] = v.f; break;
  }
}

/* ------------------------------------------------------------------------ */
/* to_<number> / try_to_<number> between number types: register_number_to_number                         */
/* (src/query/functions/src/scalars/arithmetic/src/arithmetic.rs:448-700):                                */
/*   lossless (NumberDataType::can_lossless_cast_to, types/number.rs:426-443)  -> `as`                    */
/*   round cast (float -> integer, need_round_cast_to :445-450): rounding_mode ? f64::round first : as is, */
/*     then num_traits::cast; None -> row error "number overflowed" (try_: NULL)                          */
/*   lossy (everything else): num_traits::cast; None -> row error (try_: NULL)                            */
/* num_traits::cast is the third-party crate num-traits 0.2.19 (Cargo.lock:12876; not under /root/reference); its published  */
/* rules, restated: int -> int: Some iff the value is in the destination's range; int -> float, float -> float: always      */
/* Some(`as`); float -> int truncates and is Some iff  MIN - 1 < x < MAX + 1  when the float is wider than the integer      */
/* (both bounds exact), else  MIN <= x < MAX + 1  (MIN - 1 is not representable; MAX + 1 = 2^bits is); unsigned: -1 < x.    */
/* NaN compares false everywhere -> None. Pinned by the number cases of tests/it/scalars/testdata/cast.txt.                  */
/* ------------------------------------------------------------------------ */
static int is_number(int t);
static int can_lossless(int s, int d) {
  const int sf = t_cls(s) == 2, df = t_cls(d) == 2, sb = t_bits(s), db = t_bits(d);
  if (sf && df) return sb <= db;
  if (sf && !df) return 0;
  if (!sf && df) return sb < db;
  const int ss = t_cls(s) == 0, ds = t_cls(d) == 0;
  if (ss == ds) return sb <= db;
  if (!ss && ds) return sb < 64 && sb * 2 <= db; /* next_bit_width(64) = None */
  return 0;
}
/* num_traits::cast::<Src, Dst>(v): 1 = Some (result in *out), 0 = None */
static int nt_cast(val v, int src_type, int dst_type, val* out) {
  const int dc = t_cls(dst_type), db = t_bits(dst_type);
  *out = cast_val(v, dst_type);
  if (dc == 2) return 1;
  if (v.cls != 2) { /* int -> int: in range? */
    if (dc == 0) {
      const int64_t lo = db == 64 ? INT64_MIN : -((int64_t)1 << (db - 1)), hi = db == 64 ? INT64_MAX : (((int64_t)1 << (db - 1)) - 1);
      if (v.cls == 0) return v.i >= lo && v.i <= hi;
      return v.u <= (uint64_t)hi;
    }
    const uint64_t hi = db == 64 ? UINT64_MAX : (((uint64_t)1 << db) - 1);
    if (v.cls == 0) return v.i >= 0 && (uint64_t)v.i <= hi;
    return v.u <= hi;
  }
  /* float -> int. The comparison happens in the SOURCE float type (f32 constants for an f32 source). */
  const int fb = t_bits(src_type); /* 32 or 64 */
  const double x = v.f;            /* exact also for f32 sources */
  const int wider = fb > db;       /* size_of::<F>() > size_of::<I>() */
  if (dc == 0) {
    const double min = -ldexp(1.0, db - 1), max_p1 = ldexp(1.0, db - 1);
    if (wider) return x > min - 1.0 && x < max_p1; /* exact: db <= 32 < 53 bits (f64), db <= 16 < 24 bits (f32) */
    return x >= min && x < max_p1;
  }
  const double max_p1 = ldexp(1.0, db);
  return x > -1.0 && x < max_p1;
}
/* out: dst_type values; `bitmap`: cast -> preset all ones by the caller, bit cleared for error rows (NULL input rows never
 * raise), try_ -> the result validity (input validity AND "the cast was Some"); *n_errors counts cleared bits (cast only) */
int orc_cast_num(const orc_col* src, int dst_type, int is_try, int rounding_mode, int64_t n, void* out, uint8_t* bitmap, uint64_t* n_errors) {
  if (!is_number(src->type) || !is_number(dst_type)) return -1;
  const int lossless = src->type == dst_type || can_lossless(src->type, dst_type);
  const int round_cast = t_cls(src->type) == 2 && t_cls(dst_type) != 2;
  for (int64_t i = 0; i < n; ++i) {
    val v = load_val(src, i), r;
    int some = 1;
    if (lossless) r = cast_val(v, dst_type);
    else {
      if (round_cast && rounding_mode) { v.f = round(v.f); v.bits = 64; some = nt_cast(v, ORC_T_F64, dst_type, &r); } /* as_::<f64>().round() */
      else some = nt_cast(v, src->type, dst_type, &r);
      if (!some) memset(&r, 0, sizeof r), r.cls = t_cls(dst_type), r.bits = t_bits(dst_type); /* DestType::default() */
    }
    store_val(out, dst_type, i, r);
    const int valid = col_valid(src, i);
    if (is_try) {
      if (valid && some) bitmap[i >> 3] |= (uint8_t)(1u << (i & 7)); else bitmap[i >> 3] &= (uint8_t)~(1u << (i & 7));
    } else if (!some && valid) {
      if (bitmap) bitmap[i >> 3] &= (uint8_t)~(1u << (i & 7));
      if (n_errors) ++*n_errors;
    }
  }
  return 0;
}

/* ------------------------------------------------------------------------ */
/* numeric arithmetic: numeric_basic_arithmetic.rs:255-544, arithmetic_modulo.rs */
/* result types: src/query/codegen/src/writes/arithmetics_type.rs:222-250      */
/* ------------------------------------------------------------------------ */
static int next_bits(int w) { return w < 64 ? w * 2 : 64; }
static int mk_num(int bits, int is_signed, int is_float) {
  if (is_float) return bits == 32 ? ORC_T_F32 : (bits == 64 ? ORC_T_F64 : -1);
  switch (bits) {
    case 8: return is_signed ? ORC_T_I8 : ORC_T_U8;
    case 16: return is_signed ? ORC_T_I16 : ORC_T_U16;
    case 32: return is_signed ? ORC_T_I32 : ORC_T_U32;
    case 64: return is_signed ? ORC_T_I64 : ORC_T_U64;
  }
  return -1;
}
static int is_number(int t) { return t >= ORC_T_I8 && t <= ORC_T_F64; }
static int coerce(int op, int a, int b) {
  if (!is_number(a) || !is_number(b)) return -1;
  int as = t_cls(a) != 1, bs = t_cls(b) != 1, af = t_cls(a) == 2, bf = t_cls(b) == 2;
  int sg = as || bs, fl = af || bf;
  int bw = t_bits(a) > t_bits(b) ? t_bits(a) : t_bits(b);
  switch (op) {
    case ORC_OP_PLUS: case ORC_OP_MULTIPLY: return mk_num(next_bits(bw), sg, fl);
    case ORC_OP_MINUS: return mk_num(next_bits(bw), 1, fl);
    case ORC_OP_DIVIDE: case ORC_OP_DIV0: case ORC_OP_DIVNULL: return ORC_T_F64;
    case ORC_OP_INTDIV: return mk_num(bw, sg, 0);
    case ORC_OP_MODULO:
      if (fl) return ORC_T_F64;
      return mk_num(as ? next_bits(t_bits(b)) : t_bits(b), as, 0);
    case 100: return mk_num(bw, sg, fl); /* LeastSuper */
  }
  return -1;
}
int orc_arith_result_type(int op, int l, int r) { return coerce(op, l, r); }

static void raise_err(const orc_col* a, const orc_col* b, int64_t i, uint8_t* err, uint64_t* cnt) {
  /* EvalContext::set_error ignores NULL rows (function.rs:534-556) */
  if (!col_valid(a, i) || !col_valid(b, i)) return;
  if (err) err[i >> 3] &= (uint8_t)~(1u << (i & 7));
  if (cnt) (*cnt)++;
}

int orc_arith(int op, const orc_col* lhs, const orc_col* rhs, int64_t n, int out_type, void* out,
              uint8_t* err, uint64_t* err_count) {
  if (coerce(op, lhs->type, rhs->type) != out_type) return 1;
  if (err) memset(err, 0xFF, (size_t)((n + 31) / 32) * 4);
  int mtype = coerce(100, lhs->type, rhs->type);
  for (int64_t i = 0; i < n; ++i) {
    val a = load_val(lhs, i), b = load_val(rhs, i), r;
    memset(&r, 0, sizeof r);
    r.cls = t_cls(out_type); r.bits = t_bits(out_type);
    switch (op) {
      case ORC_OP_PLUS: case ORC_OP_MINUS: case ORC_OP_MULTIPLY: {
        val x = cast_val(a, out_type), y = cast_val(b, out_type);
        if (r.cls == 2) r.f = op == ORC_OP_PLUS ? x.f + y.f : (op == ORC_OP_MINUS ? x.f - y.f : x.f * y.f);
        else {
          uint64_t xu = r.cls == 0 ? (uint64_t)x.i : x.u, yu = r.cls == 0 ? (uint64_t)y.i : y.u;
          uint64_t z = op == ORC_OP_PLUS ? xu + yu : (op == ORC_OP_MINUS ? xu - yu : xu * yu);
          if (r.cls == 0) r.i = wrap_i(z, r.bits); else r.u = wrap_u(z, r.bits);
        }
      } break;
      case ORC_OP_DIVIDE: { /* divide_function :410-427 */
        double y = cast_val(b, ORC_T_F64).f;
        if (y == 0.0) { raise_err(lhs, rhs, i, err, err_count); r.f = 0.0; }
        else r.f = cast_val(a, ORC_T_F64).f / y;
      } break;
      case ORC_OP_DIV0: { /* div0_function :441-448: x / 0 = F64::default(), never raises */
        double y = cast_val(b, ORC_T_F64).f;
        r.f = y == 0.0 ? 0.0 : cast_val(a, ORC_T_F64).f / y;
      } break;
      case ORC_OP_DIVNULL: { /* divnull_function :450-457: x / 0 = None — reported through the row bitmap */
        double y = cast_val(b, ORC_T_F64).f;
        if (y == 0.0) { raise_err(lhs, rhs, i, err, err_count); r.f = 0.0; }
        else r.f = cast_val(a, ORC_T_F64).f / y;
      } break;
      case ORC_OP_INTDIV: { /* register_intdiv :459-490 */
        double y = cast_val(b, ORC_T_F64).f;
        if (y == 0.0) raise_err(lhs, rhs, i, err, err_count);
        else {
          val q; memset(&q, 0, sizeof q); q.cls = 2; q.bits = 64; q.f = cast_val(a, ORC_T_F64).f / y;
          r = cast_val(q, out_type);
        }
      } break;
      default: { /* push_modulo_result arithmetic_modulo.rs:70-95 */
        int zero = b.cls == 2 ? (b.f == 0.0) : (b.cls == 0 ? b.i == 0 : b.u == 0);
        if (zero) { raise_err(lhs, rhs, i, err, err_count); break; }
        val x = cast_val(a, mtype), y = cast_val(b, mtype), m;
        memset(&m, 0, sizeof m); m.cls = x.cls; m.bits = x.bits;
        if (x.cls == 2) m.f = x.bits == 32 ? (double)fmodf((float)x.f, (float)y.f) : fmod(x.f, y.f);
        else if (x.cls == 0) {
          int64_t mn = x.bits == 64 ? INT64_MIN : -((int64_t)1 << (x.bits - 1));
          m.i = (x.i == mn && y.i == -1) ? 0 : x.i % y.i;
        } else m.u = x.u % y.u;
        r = cast_val(m, out_type);
      } break;
    }
    store_val(out, out_type, i, r);
  }
  return 0;
}

/* config 1 in the reference's shape: per block of block_rows rows, materialise b*c, then a+(b*c), then
 * NumberSumState::add_batch (aggregate_sum.rs:71-129); all i64 wrapping (Cargo.toml:577). */
int64_t orc_sum_a_plus_b_mul_c_i64(const int64_t* a, const int64_t* b, const int64_t* c, int64_t n,
                                   int64_t block_rows) {
  uint64_t state = 0;
  int64_t* t1 = (int64_t*)malloc(sizeof(int64_t) * (size_t)block_rows);
  int64_t* t2 = (int64_t*)malloc(sizeof(int64_t) * (size_t)block_rows);
  for (int64_t s = 0; s < n; s += block_rows) {
    int64_t m = n - s < block_rows ? n - s : block_rows;
    for (int64_t i = 0; i < m; ++i) t1[i] = (int64_t)((uint64_t)b[s + i] * (uint64_t)c[s + i]);
    for (int64_t i = 0; i < m; ++i) t2[i] = (int64_t)((uint64_t)a[s + i] + (uint64_t)t1[i]);
    uint64_t sum = 0;
    for (int64_t i = 0; i < m; ++i) sum += (uint64_t)t2[i];
    state += sum;
  }
  free(t1); free(t2);
  return (int64_t)state;
}

/* ------------------------------------------------------------------------ */
/* decimal: decimal/src/arithmetic.rs:80-316, types/decimal.rs:759-797,1024-1060 */
/* ------------------------------------------------------------------------ */
static i128 e10(int k) { i128 r = 1; while (k-- > 0) r *= 10; return r; }

/* minimal signed 256-bit integer: 4 little-endian u64 limbs, two's complement */
typedef struct { uint64_t w[4]; } i256;
static i256 i256_from_i128(i128 v) {
  i256 r; r.w[0] = (uint64_t)v; r.w[1] = (uint64_t)((u128)v >> 64);
  r.w[2] = r.w[3] = v < 0 ? ~(uint64_t)0 : 0; return r;
}
static int i256_neg_p(i256 a) { return (a.w[3] >> 63) & 1; }
static i256 i256_add(i256 a, i256 b) {
  i256 r; u128 c = 0;
  for (int i = 0; i < 4; ++i) { c += (u128)a.w[i] + b.w[i]; r.w[i] = (uint64_t)c; c >>= 64; }
  return r;
}
static i256 i256_negate(i256 a) {
  i256 one = {{1, 0, 0, 0}};
  for (int i = 0; i < 4; ++i) a.w[i] = ~a.w[i];
  return i256_add(a, one);
}
static i256 i256_sub(i256 a, i256 b) { return i256_add(a, i256_negate(b)); }
static i256 i256_mul(i256 a, i256 b) { /* wrapping */
  i256 r = {{0, 0, 0, 0}};
  for (int i = 0; i < 4; ++i) {
    u128 c = 0;
    for (int j = 0; i + j < 4; ++j) {
      c += (u128)a.w[i] * b.w[j] + r.w[i + j];
      r.w[i + j] = (uint64_t)c; c >>= 64;
    }
  }
  return r;
}
static int u256_ge(const uint64_t* a, const uint64_t* b) {
  for (int i = 3; i >= 0; --i) { if (a[i] != b[i]) return a[i] > b[i]; }
  return 1;
}
/* truncating signed division */
static i256 i256_div(i256 a, i256 b) {
  int na = i256_neg_p(a), nb = i256_neg_p(b);
  if (na) a = i256_negate(a);
  if (nb) b = i256_negate(b);
  i256 q = {{0, 0, 0, 0}}, r = {{0, 0, 0, 0}};
  for (int bit = 255; bit >= 0; --bit) {
    /* r = (r << 1) | bit(a) */
    for (int i = 3; i > 0; --i) r.w[i] = (r.w[i] << 1) | (r.w[i - 1] >> 63);
    r.w[0] = (r.w[0] << 1) | ((a.w[bit >> 6] >> (bit & 63)) & 1);
    if (u256_ge(r.w, b.w)) { r = i256_sub(r, b); q.w[bit >> 6] |= (uint64_t)1 << (bit & 63); }
  }
  return (na != nb) ? i256_negate(q) : q;
}
static int i256_fits_i128(i256 a) {
  uint64_t ext = (a.w[1] >> 63) ? ~(uint64_t)0 : 0;
  return a.w[2] == ext && a.w[3] == ext;
}
static i128 i256_low128(i256 a) { return (i128)(((u128)a.w[1] << 64) | a.w[0]); }

typedef struct { int p, s; } dsize;
static int dec_props(int type, int p, int s, dsize* o) {
  switch (type) {
    case ORC_T_DEC64: case ORC_T_DEC128: case ORC_T_DEC256: o->p = p; o->s = s; return 1;
    case ORC_T_I8: case ORC_T_U8: o->p = 3; o->s = 0; return 1; /* number.rs:452-465 */
    case ORC_T_I16: case ORC_T_U16: o->p = 5; o->s = 0; return 1;
    case ORC_T_I32: case ORC_T_U32: o->p = 10; o->s = 0; return 1;
    case ORC_T_I64: o->p = 19; o->s = 0; return 1;
    case ORC_T_U64: o->p = 20; o->s = 0; return 1;
  }
  return 0;
}
static int imin(int a, int b) { return a < b ? a : b; }
static int imax(int a, int b) { return a > b ? a : b; }
/* ArithmeticOp::result_size arithmetic.rs:80-139 */
static int result_size(int op, dsize a, dsize b, dsize* l, dsize* r, dsize* ret) {
  int precision, scale, la = a.p - a.s, lb = b.p - b.s;
  switch (op) {
    case ORC_OP_MULTIPLY: scale = imin(a.s + b.s, imax(imax(a.s, b.s), 12)); precision = la + lb + scale; break;
    case ORC_OP_DIVIDE: scale = imax(a.s, imin(a.s + 6, 12)); precision = la + b.s + scale; break;
    case ORC_OP_PLUS: case ORC_OP_MINUS: scale = imax(a.s, b.s); precision = imax(la, lb) + scale + 1; break;
    default: return 0;
  }
  precision = imin(precision, (a.p <= 38 && b.p <= 38) ? 38 : 76); /* :115-121: both Decimal128 or smaller -> clamp to 38 */
  if (precision < 1 || scale > precision) return 0;
  ret->p = precision; ret->s = scale;
  if (op == ORC_OP_MULTIPLY) { l->p = precision; l->s = a.s; r->p = precision; r->s = b.s; }
  else if (op == ORC_OP_DIVIDE) { int pp = imax(precision, imax(a.p, b.p)); l->p = pp; l->s = a.s; r->p = pp; r->s = b.s; }
  else { *l = *ret; *r = *ret; }
  return 1;
}
int orc_decimal_result_size(int op, int lp, int ls, int rp, int rs, int* out_p, int* out_s) {
  /* (the 38 / 76 clamp lives in result_size) */
  dsize a = {lp, ls}, b = {rp, rs}, l, r, ret;
  if (!result_size(op, a, b, &l, &r, &ret)) return 1;
  *out_p = ret.p; *out_s = ret.s; return 0;
}

static i128 load_dec(const orc_col* c, int64_t i) {
  int64_t j = c->is_scalar ? 0 : i;
  if (c->type == ORC_T_DEC128) return ((const i128*)c->data)[j];
  if (c->type == ORC_T_DEC256) return ((const i128*)c->data)[2 * j]; /* as_decimal::<T>() into a narrower T: the low bits */
  val v = load_val(c, i);
  return v.cls == 0 ? (i128)v.i : (i128)v.u;
}
static int checked_mul_t(i128 x, i128 f, int t128, i128* out) {
  if (!t128) { i128 r = x * f; if (r > INT64_MAX || r < INT64_MIN) return 0; *out = r; return 1; }
  i128 r;
  if (__builtin_mul_overflow(x, f, &r)) return 0;
  *out = r; return 1;
}
static i128 wrap_t(i128 v, int t128) { return t128 ? v : (i128)(int64_t)v; }
/* convert_to_decimal arithmetic.rs:141-153 -> integer_to_decimal cast.rs:701-753 / decimal_expand_cast :901-979 */
static int convert_operand(i128 x, int is_dec, int from_s, dsize to, int t128, i128* out) {
  i128 mx = e10(to.p) - 1;
  if (!is_dec) {
    if (to.s == 0) { *out = wrap_t(x, t128); return 1; }
    if (!t128 && (x > INT64_MAX || x < INT64_MIN)) return 0;
    i128 r; if (!checked_mul_t(x, e10(to.s), t128, &r)) return 0;
    if (r > mx || r < -mx) return 0;
    *out = r; return 1;
  }
  if (from_s == to.s) { *out = wrap_t(x, t128); return 1; }
  i128 r; if (!checked_mul_t(wrap_t(x, t128), e10(to.s - from_s), t128, &r)) return 0;
  if (r > mx || r < -mx) return 0;
  *out = r; return 1;
}

int orc_decimal_arith(int op, const orc_col* lhs, const orc_col* rhs, int64_t n, int out_type, int out_p,
                      int out_s, void* out, uint8_t* err, uint64_t* err_count) {
  dsize a, b, l, r, ret;
  if (!dec_props(lhs->type, lhs->precision, lhs->scale, &a) || !dec_props(rhs->type, rhs->precision, rhs->scale, &b)) return 1;
  if (!result_size(op, a, b, &l, &r, &ret)) return 1;
  if (ret.p > 38) /* T = i256: oracle/decimal256.c */
    return out_type == ORC_T_DEC256 ? orc_decimal256_arith(op, lhs, rhs, n, out_p, out_s, out, err, err_count) : 1;
  int t128 = ret.p > 18;
  if (ret.p != out_p || ret.s != out_s || out_type != (t128 ? ORC_T_DEC128 : ORC_T_DEC64)) return 1;
  int a_dec = lhs->type == ORC_T_DEC64 || lhs->type == ORC_T_DEC128 || lhs->type == ORC_T_DEC256;
  int b_dec = rhs->type == ORC_T_DEC64 || rhs->type == ORC_T_DEC128 || rhs->type == ORC_T_DEC256;
  int overflow = ret.p == (t128 ? 38 : 18); /* binary_decimal :203 */
  if (err) memset(err, 0xFF, (size_t)((n + 31) / 32) * 4);
  for (int64_t i = 0; i < n; ++i) {
    i128 x, y, res = 1;
    int ok = convert_operand(load_dec(lhs, i), a_dec, a.s, l, t128, &x);
    ok = convert_operand(load_dec(rhs, i), b_dec, b.s, r, t128, &y) && ok;
    if (ok) switch (op) {
      case ORC_OP_PLUS: case ORC_OP_MINUS: {
        i128 t = wrap_t(op == ORC_OP_PLUS ? (i128)((u128)x + (u128)y) : (i128)((u128)x - (u128)y), t128);
        if (overflow) { i128 mx = e10(ret.p) - 1; if (t < -mx || t > mx) ok = 0; }
        res = t;
      } break;
      case ORC_OP_MULTIPLY: {
        int sm = a.s + b.s - ret.s;
        if (sm == 0) res = wrap_t((i128)((u128)x * (u128)y), t128);
        else if (!t128) { /* i64 do_round_mul decimal.rs:759-786 */
          if (!overflow) {
            int64_t d = (int64_t)e10(sm);
            int64_t pr = (int64_t)((uint64_t)(int64_t)x * (uint64_t)(int64_t)y);
            int64_t num = ((x < 0) == (y < 0)) ? (int64_t)((uint64_t)pr + (uint64_t)(d / 2))
                                               : (int64_t)((uint64_t)pr - (uint64_t)(d / 2));
            res = num / d;
          } else {
            i128 d = e10(sm);
            i128 q = (((x < 0) == (y < 0)) ? x * y + d / 2 : x * y - d / 2) / d;
            i128 mx = e10(18) - 1;
            if (q < -mx || q > mx) ok = 0;
            res = q;
          }
        } else { /* i128 do_round_mul decimal.rs:1024-1054 */
          i128 d = e10(sm);
          if (!overflow) {
            i128 pr = (i128)((u128)x * (u128)y);
            i128 num = ((x < 0) == (y < 0)) ? (i128)((u128)pr + (u128)(d / 2)) : (i128)((u128)pr - (u128)(d / 2));
            res = num / d;
          } else {
            i256 D = i256_from_i128(d), half = i256_from_i128(d / 2);
            i256 pr = i256_mul(i256_from_i128(x), i256_from_i128(y));
            i256 q = i256_div(((x < 0) == (y < 0)) ? i256_add(pr, half) : i256_sub(pr, half), D);
            if (!i256_fits_i128(q)) ok = 0;
            res = i256_low128(q);
          }
        }
      } break;
      default: { /* divide :212-243 */
        int sm = b.s + ret.s - a.s;
        if (y == 0) ok = 0;
        else if (!t128) { /* i64 do_round_div decimal.rs:788-797 */
          i128 am = (i128)((u128)x * (u128)e10(sm));
          i128 num = ((x < 0) == (y < 0)) ? (i128)((u128)am + (u128)(y / 2)) : (i128)((u128)am - (u128)(y / 2));
          res = (i128)(int64_t)(num / y);
        } else { /* i128 do_round_div decimal.rs:1056-1064 */
          i256 Y = i256_from_i128(y), half = i256_from_i128(y / 2);
          i256 am = i256_mul(i256_from_i128(x), i256_from_i128(e10(sm)));
          i256 q = i256_div(((x < 0) == (y < 0)) ? i256_add(am, half) : i256_sub(am, half), Y);
          res = i256_low128(q);
        }
      } break;
    }
    if (!ok) { raise_err(lhs, rhs, i, err, err_count); res = 1; }
    if (t128) ((i128*)out)[i] = res; else ((int64_t*)out)[i] = (int64_t)res;
  }
  return 0;
}

/* ------------------------------------------------------------------------ */
/* comparisons -> Bitmap (register_comparison.rs:52-96, bitmap/immutable.rs:474) */
/* ------------------------------------------------------------------------ */
static const uint8_t* view_bytes(const uint32_t* v, const void* const* buffers, uint32_t* len) {
  *len = v[0];
  if (*len <= 12) return (const uint8_t*)(v + 1);
  return (const uint8_t*)buffers[v[2]] + v[3];
}
static int cmp3_col(const orc_col* a, const orc_col* b, int64_t i) {
  int64_t ja = a->is_scalar ? 0 : i, jb = b->is_scalar ? 0 : i;
  switch (a->type) {
    case ORC_T_DEC128: { i128 x = ((const i128*)a->data)[ja], y = ((const i128*)b->data)[jb]; return (x > y) - (x < y); }
    case ORC_T_BOOL: { int x = bit_get((const uint8_t*)a->data, ja), y = bit_get((const uint8_t*)b->data, jb); return x - y; }
    case ORC_T_STRING: {
      uint32_t la, lb;
      const uint8_t* pa = view_bytes((const uint32_t*)a->data + 4 * ja, a->buffers, &la);
      const uint8_t* pb = view_bytes((const uint32_t*)b->data + 4 * jb, b->buffers, &lb);
      int c = memcmp(pa, pb, la < lb ? la : lb);
      if (c) return c < 0 ? -1 : 1;
      return (la > lb) - (la < lb);
    }
  }
  val x = load_val(a, i), y = load_val(b, i);
  if (x.cls == 0) return (x.i > y.i) - (x.i < y.i);
  if (x.cls == 1) return (x.u > y.u) - (x.u < y.u);
  int xn = x.f != x.f, yn = y.f != y.f; /* OrderedFloat: NaN == NaN, NaN largest */
  if (xn || yn) return xn - yn;
  return (x.f > y.f) - (x.f < y.f);
}
int orc_cmp(int op, const orc_col* lhs, const orc_col* rhs, int64_t n, uint8_t* out) {
  if (lhs->type != rhs->type) return 1;
  memset(out, 0, (size_t)((n + 7) / 8));
  for (int64_t i = 0; i < n; ++i) {
    int c = cmp3_col(lhs, rhs, i), r;
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
/* Decimal comparison of two different DecimalSizes — DecimalCmp::eval + CmpOp::compare
 * (src/query/functions/src/scalars/decimal/src/comparison.rs:326-384 calc_size, :407-441 compare): T = storage class of
 * (max leading digits + max scale, capped at 38); f_a = 10^(s - s_a), f_b = 10^(s - s_b); different signs compare as they
 * are, else each side is multiplied by its factor with checked_mul, whose overflow decides the order. */
static int dec_checked_mul_T(i128 x, i128 f, int t128, i128* out) {
  if (!t128) { /* i64::checked_mul */
    i128 r = x * f;
    if (r > (i128)INT64_MAX || r < (i128)INT64_MIN) return 0;
    *out = r; return 1;
  }
  i128 r;
  if (__builtin_mul_overflow(x, f, &r)) return 0; /* i128::checked_mul */
  *out = r; return 1;
}
int orc_cmp_decimal(int op, const orc_col* lhs, const orc_col* rhs, int64_t n, uint8_t* out) {
  int l_dec = lhs->type == ORC_T_DEC64 || lhs->type == ORC_T_DEC128, r_dec = rhs->type == ORC_T_DEC64 || rhs->type == ORC_T_DEC128;
  if (!l_dec || !r_dec) return 1;
  int scale = lhs->scale > rhs->scale ? lhs->scale : rhs->scale;
  int la = lhs->precision - lhs->scale, lb = rhs->precision - rhs->scale;
  int precision = (la > lb ? la : lb) + scale;
  if (precision > 38) precision = 38;
  int t128 = precision > 18;
  i128 fa = e10(scale - lhs->scale), fb = e10(scale - rhs->scale);
  memset(out, 0, (size_t)((n + 7) / 8));
  for (int64_t i = 0; i < n; ++i) {
    int64_t ja = lhs->is_scalar ? 0 : i, jb = rhs->is_scalar ? 0 : i;
    i128 a = lhs->type == ORC_T_DEC128 ? ((const i128*)lhs->data)[ja] : (i128)((const int64_t*)lhs->data)[ja];
    i128 b = rhs->type == ORC_T_DEC128 ? ((const i128*)rhs->data)[jb] : (i128)((const int64_t*)rhs->data)[jb];
    int c;
    if (fa == fb) c = (a > b) - (a < b);
    else {
      int sa = (a > 0) - (a < 0), sb = (b > 0) - (b < 0);
      if (sa != sb) c = (a > b) - (a < b);
      else {
        i128 x = a, y = b;
        if (fa != 1 && !dec_checked_mul_T(a, fa, t128, &x)) c = sa > 0 ? 1 : -1;
        else if (fb != 1 && !dec_checked_mul_T(b, fb, t128, &y)) c = sb > 0 ? -1 : 1;
        else c = (x > y) - (x < y);
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
int64_t orc_filter_select(const uint8_t* bm, int64_t off, int64_t n, uint32_t* out_sel) {
  int64_t k = 0;
  for (int64_t i = 0; i < n; ++i) if (bit_get(bm, off + i)) out_sel[k++] = (uint32_t)i;
  return k;
}
void orc_take(const void* src, int es, const uint32_t* sel, int64_t n, void* out) {
  for (int64_t i = 0; i < n; ++i) memcpy((uint8_t*)out + i * es, (const uint8_t*)src + (size_t)sel[i] * es, (size_t)es);
}

/* take_ranges (kernels/take_ranges.rs:40) / take_compacted_indices (take_compact.rs:38) as selection vectors, take_blocks
 * (take_chunks.rs:70-190) for one column */
int64_t orc_sel_from_ranges(const uint32_t* ranges, int n, uint32_t* out) {
  int64_t k = 0;
  for (int r = 0; r < n; ++r) for (uint32_t i = ranges[2 * r]; i < ranges[2 * r + 1]; ++i) out[k++] = i;
  return k;
}
int64_t orc_sel_from_repeats(const uint32_t* rep, int n, uint32_t* out) {
  int64_t k = 0;
  for (int r = 0; r < n; ++r) for (uint32_t c = 0; c < rep[2 * r + 1]; ++c) out[k++] = rep[2 * r];
  return k;
}
void orc_take_chunks(const void* const* blocks, int es, const uint32_t* pairs, int64_t n, void* out) {
  for (int64_t i = 0; i < n; ++i)
    memcpy((uint8_t*)out + i * es, (const uint8_t*)blocks[pairs[2 * i]] + (size_t)pairs[2 * i + 1] * es, (size_t)es);
}

/* DataBlock::scatter (kernels/scatter.rs:20-66): divide_indices_by_scatter_size counts the rows per destination, then pushes every
 * row id onto its destination's list in row order (:46-66); each destination is take_with_optimize_size of its list. out_rows = the
 * lists back to back, out_starts[d] .. out_starts[d + 1] = destination d's slice. */
void orc_divide_indices(const uint32_t* indices, int64_t n, int scatter_size, uint32_t* out_rows, int64_t* out_starts) {
  int64_t* cur = (int64_t*)calloc((size_t)scatter_size + 1, sizeof(int64_t));
  for (int64_t i = 0; i < n; ++i) cur[indices[i] + 1]++;              /* scatter_num_rows */
  for (int d = 0; d < scatter_size; ++d) cur[d + 1] += cur[d];
  memcpy(out_starts, cur, ((size_t)scatter_size + 1) * sizeof(int64_t));
  for (int64_t i = 0; i < n; ++i) out_rows[cur[indices[i]]++] = (uint32_t)i;   /* scatter_indices[index].push(i) */
  free(cur);
}
/* Bitmap / Boolean columns under take (kernels/take.rs: take_boolean_types / the validity of a NullableColumn): bit `sel[i]` of the
 * source (read from bit_offset) becomes bit i of the output (written from out_bit_offset; other bits of `out` are left alone) */
void orc_take_bitmap(const uint8_t* src, int64_t bit_offset, const uint32_t* sel, int64_t n_sel, uint8_t* out, int64_t out_bit_offset) {
  for (int64_t i = 0; i < n_sel; ++i) {
    const int64_t sb = bit_offset + sel[i], ob = out_bit_offset + i;
    const int bit = (src[sb >> 3] >> (sb & 7)) & 1;
    out[ob >> 3] = (uint8_t)((out[ob >> 3] & ~(1u << (ob & 7))) | ((unsigned)bit << (ob & 7)));
  }
}
/* DataBlock::concat for one column (kernels/concat.rs:62-110): fixed-width values back to back (concat_primitive_types :262-274);
 * Boolean values and the validity of a NullableColumn bit by bit (concat_boolean_types :307-320; a block without validity counts
 * as all valid). */
void orc_concat_fixed(const void* const* blocks, const int64_t* rows, int nblocks, int elem_size, void* out) {
  uint8_t* o = (uint8_t*)out;
  for (int b = 0; b < nblocks; ++b) { memcpy(o, blocks[b], (size_t)rows[b] * elem_size); o += (size_t)rows[b] * elem_size; }
}
void orc_concat_bitmap(const uint8_t* const* blocks, const int64_t* bit_offsets, const int64_t* rows, int nblocks, uint8_t* out) {
  int64_t ob = 0;
  for (int b = 0; b < nblocks; ++b)
    for (int64_t i = 0; i < rows[b]; ++i, ++ob) {
      const int64_t sb = (bit_offsets ? bit_offsets[b] : 0) + i;
      const int bit = blocks[b] ? (blocks[b][sb >> 3] >> (sb & 7)) & 1 : 1;
      out[ob >> 3] = (uint8_t)((out[ob >> 3] & ~(1u << (ob & 7))) | ((unsigned)bit << (ob & 7)));
    }
}

/* ------------------------------------------------------------------------ */
/* group hash: aggregate/group_hash.rs:38,180-207,267-281,509-632              */
/* ------------------------------------------------------------------------ */
#define NULL_HASH_VAL 0xd1cefa08eb382d69ULL
uint64_t orc_agg_hash_bytes(const uint8_t* p, uint64_t len) { /* :522-553 */
  const uint64_t M = 0xc6a4a7935bd1e995ULL, SEED = 0xe17a1465ULL; const int R = 47;
  uint64_t h = SEED ^ (len * M);
  uint64_t nblocks = len / 8;
  for (uint64_t i = 0; i < nblocks; ++i) {
    uint64_t k; memcpy(&k, p + i * 8, 8);
    k *= M; k ^= k >> R; k *= M;
    h ^= k; h *= M;
  }
  const uint8_t* d = p + nblocks * 8; uint64_t dl = len - nblocks * 8;
  for (uint64_t i = 0; i < dl; ++i) h ^= (uint64_t)d[i] << (8 * (dl - i - 1));
  h ^= h >> R; h *= M; h ^= h >> R;
  return h;
}
uint64_t orc_agg_hash_u64(uint64_t x) { /* :555-570 */
  x ^= x >> 32; x *= 0xd6e8feb86659fd93ULL; x ^= x >> 32; x *= 0xd6e8feb86659fd93ULL; x ^= x >> 32;
  return x;
}
static uint64_t hash_col_row(const orc_col* c, int64_t i) {
  if (!col_valid(c, i)) return NULL_HASH_VAL;
  int64_t j = c->is_scalar ? 0 : i;
  switch (c->type) {
    case ORC_T_BOOL: return (uint64_t)bit_get((const uint8_t*)c->data, j); /* :581-585 */
    case ORC_T_F32: { /* :599-609 */
      float f = ((const float*)c->data)[j]; uint32_t b; memcpy(&b, &f, 4);
      if (f != f) b = 0x7fc00000u; /* f32::NAN.to_bits() */
      return orc_agg_hash_u64(b);
    }
    case ORC_T_F64: { /* :611-620 */
      double f = ((const double*)c->data)[j]; uint64_t b; memcpy(&b, &f, 8);
      if (f != f) b = 0x7ff8000000000000ULL;
      return orc_agg_hash_u64(b);
    }
    case ORC_T_DEC128: return orc_agg_hash_bytes((const uint8_t*)c->data + 16 * j, 16); /* :587-591 */
    case ORC_T_DEC256: return orc_agg_hash_bytes((const uint8_t*)c->data + 32 * j, 32); /* :593-597 */
    case ORC_T_STRING: { uint32_t len; const uint8_t* p = view_bytes((const uint32_t*)c->data + 4 * j, c->buffers, &len);
                         return orc_agg_hash_bytes(p, len); }
  }
  val v = load_val(c, i);
  return orc_agg_hash_u64(v.cls == 0 ? (uint64_t)v.i : v.u); /* `*self as u64` */
}
int orc_group_hash(const orc_col* cols, int ncols, int64_t n, uint64_t* out) { /* group_hash_entries :40-61 */
  for (int k = 0; k < ncols; ++k)
    for (int64_t i = 0; i < n; ++i) {
      uint64_t h = hash_col_row(&cols[k], i);
      out[i] = k == 0 ? h : (out[i] * NULL_HASH_VAL ^ h);
    }
  return 0;
}

/* ------------------------------------------------------------------------ */
/* AggregateHashTable restatement                                             */
/*   HashIndex          hash_index/index.rs:26-236, group.rs:25-52, bitmask.rs */
/*   Payload rows       payload.rs:192-248,361-486 ; payload_row.rs:51-215,324+ */
/*   add_groups         aggregate_hashtable.rs:168-333                          */
/*   states             aggregate_sum.rs, aggregate_count.rs, aggregate_unary.rs */
/* ------------------------------------------------------------------------ */
#define BATCH_SIZE 2048
#define GROUP_WIDTH 8
#define TAG_EMPTY 0xFF
#define LOAD_FACTOR 1.35
#define MAXK 16
#define MAXA 24

struct orc_hashagg {
  int nkeys, naggs;
  int key_type[MAXK], key_nullable[MAXK], key_off[MAXK], key_size[MAXK], validity_off[MAXK];
  orc_agg_desc aggs[MAXA];
  int state_off[MAXA], state_size;
  int hash_off, state_ptr_off, tuple_size;
  /* hash index */
  uint8_t* ctrls; uint8_t** pointers; size_t capacity, mask, count;
  int resize_count;
  /* payload: one growing array of rows + one of states (arena) */
  uint8_t* rows; size_t nrows, rows_cap;
  uint8_t* states; size_t states_len, states_cap;
  /* string arena */
  uint8_t* strs; size_t strs_len, strs_cap;
};

static int rowformat_size(int t) { /* payload_row.rs:51-83 */
  switch (t) {
    case ORC_T_BOOL: return 1;
    case ORC_T_STRING: return 12; /* u32 len + address */
    case ORC_T_DEC128: return 16;
    case ORC_T_DEC256: return 32;
    default: return t_size(t);
  }
}
/* sum over a Nullable argument is wrapped in AggregateNullUnaryAdaptor<true> (adaptors/aggregate_null_adaptor.rs:366-400):
 * the nested state is followed by one flag byte "a non-NULL row was seen" (here: 8 / 16 bytes to keep alignment);
 * merge_result yields NULL while the flag is clear. For min/max the Option's has-value word plays that role. */
static int agg_flag_off(const orc_agg_desc* d) { /* 0 = no flag */
  if (d->kind != ORC_AGG_SUM || !d->arg_nullable) return 0;
  return d->arg_type == ORC_T_DEC256 ? 32 : (d->arg_type == ORC_T_DEC128 ? 16 : 8);
}
static int agg_state_size(const orc_agg_desc* d) {
  if (d->kind == ORC_AGG_SUM && d->arg_type == ORC_T_DEC256) return d->arg_nullable ? 40 : 32; /* DecimalSumState<_, i256>: [u64; 4] */
  if (d->kind == ORC_AGG_SUM && d->arg_type == ORC_T_DEC128) return d->arg_nullable ? 32 : 16;
  if (d->kind == ORC_AGG_SUM && d->arg_nullable) return 16; /* value + flag */
  if ((d->kind == ORC_AGG_MIN || d->kind == ORC_AGG_MAX) && d->arg_type == ORC_T_DEC256) return 48; /* Option<[u64; 4]> (aggregate_min_max_any_decimal.rs:40-43): value + has */
  if (d->kind == ORC_AGG_MIN || d->kind == ORC_AGG_MAX) return (d->arg_type == ORC_T_DEC128 || d->arg_type == ORC_T_STRING) ? 32 : 16; /* value + has flag (Decimal128: 16-byte value; String: offset + length into the table's bytes) */
  return 8;
}
static int mm_has_off(const orc_agg_desc* d) { return d->arg_type == ORC_T_DEC256 ? 32 : (d->arg_type == ORC_T_DEC128 || d->arg_type == ORC_T_STRING) ? 16 : 8; } /* MinMaxAnyState: Option<value> */
/* i256 order on four little-endian u64: the top word signed, the rest unsigned */
static int i256w_cmp(const uint64_t* a, const uint64_t* b) {
  if (a[3] != b[3]) return (int64_t)a[3] < (int64_t)b[3] ? -1 : 1;
  for (int q = 2; q >= 0; --q) if (a[q] != b[q]) return a[q] < b[q] ? -1 : 1;
  return 0;
}

static void index_alloc(orc_hashagg* h, size_t cap) {
  h->capacity = cap; h->mask = cap - 1; h->count = 0;
  h->ctrls = (uint8_t*)malloc(cap + GROUP_WIDTH);
  memset(h->ctrls, TAG_EMPTY, cap + GROUP_WIDTH);
  h->pointers = (uint8_t**)calloc(cap, sizeof(uint8_t*));
}

orc_hashagg* orc_hashagg_create(const int32_t* key_types, const uint8_t* key_nullable, int nkeys,
                                const orc_agg_desc* aggs, int naggs) {
  if (nkeys < 1 || nkeys > MAXK || naggs > MAXA) return NULL;
  orc_hashagg* h = (orc_hashagg*)calloc(1, sizeof(*h));
  h->nkeys = nkeys; h->naggs = naggs;
  int ts = 0;
  for (int k = 0; k < nkeys; ++k) { /* Payload::new payload.rs:192-248 */
    h->key_type[k] = key_types[k]; h->key_nullable[k] = key_nullable ? key_nullable[k] : 0;
    if (h->key_nullable[k]) { h->validity_off[k] = ts; ts += 1; }
  }
  for (int k = 0; k < nkeys; ++k) { h->key_off[k] = ts; h->key_size[k] = rowformat_size(key_types[k]); ts += h->key_size[k]; }
  h->hash_off = ts; ts += 8;
  h->state_ptr_off = ts; if (naggs) ts += 8;
  h->tuple_size = ts;
  int so = 0;
  for (int a = 0; a < naggs; ++a) {
    h->aggs[a] = aggs[a];
    int sz = agg_state_size(&aggs[a]);
    so = (so + sz - 1) / sz * sz; /* natural alignment */
    h->state_off[a] = so; so += sz;
  }
  h->state_size = (so + 15) / 16 * 16;
  index_alloc(h, 32768); /* aggregate_hashtable.rs:492-494 */
  return h;
}

static void set_ctrl(orc_hashagg* h, size_t index, uint8_t tag) { /* index.rs:65-72 */
  size_t index2 = ((index - GROUP_WIDTH) & h->mask) + GROUP_WIDTH;
  h->ctrls[index] = tag; h->ctrls[index2] = tag;
}
static uint8_t tag_full(uint64_t hash) { return (uint8_t)((hash >> 57) & 0x7f); } /* bitmask.rs:90-93 */

/* find_or_insert index.rs:92-110: scan 8-byte groups for the tag, else first empty */
static size_t find_or_insert(orc_hashagg* h, size_t pos, uint64_t hash, int* is_new) {
  uint8_t tag = tag_full(hash);
  for (;;) {
    const uint8_t* g = h->ctrls + pos;
    for (int b = 0; b < GROUP_WIDTH; ++b)
      if (g[b] == tag) { *is_new = 0; return (pos + b) & h->mask; }
    for (int b = 0; b < GROUP_WIDTH; ++b)
      if (g[b] & 0x80) { size_t idx = (pos + b) & h->mask; set_ctrl(h, idx, tag); *is_new = 1; return idx; }
    pos = (pos + GROUP_WIDTH) & h->mask;
  }
}
static size_t probe_empty(orc_hashagg* h, uint64_t hash) { /* index.rs:134-146 */
  size_t pos = hash & h->mask;
  for (;;) { if (h->ctrls[pos] == TAG_EMPTY) { set_ctrl(h, pos, tag_full(hash)); return pos; } pos = (pos + 1) & h->mask; }
}

static void key_field(const orc_hashagg* h, const orc_col* c, int k, int64_t i, uint8_t* dst, orc_hashagg* arena) {
  int64_t j = c->is_scalar ? 0 : i;
  switch (h->key_type[k]) {
    case ORC_T_BOOL: dst[0] = (uint8_t)bit_get((const uint8_t*)c->data, j); break;
    case ORC_T_STRING: {
      uint32_t len; const uint8_t* p = view_bytes((const uint32_t*)c->data + 4 * j, c->buffers, &len);
      memcpy(dst, &len, 4);
      if (arena) { /* copy var-len data into the arena, store its offset (stands in for the address) */
        if (arena->strs_len + len > arena->strs_cap) {
          arena->strs_cap = (arena->strs_len + len) * 2 + 64; arena->strs = (uint8_t*)realloc(arena->strs, arena->strs_cap);
        }
        memcpy(arena->strs + arena->strs_len, p, len);
        uint64_t off = arena->strs_len; memcpy(dst + 4, &off, 8); arena->strs_len += len;
      }
    } break;
    default: memcpy(dst, (const uint8_t*)c->data + (size_t)j * h->key_size[k], (size_t)h->key_size[k]); break;
  }
}
/* row_match_entries payload_row.rs:324+: compare stored row with the probing row */
static int row_match(const orc_hashagg* h, const uint8_t* row, const orc_col* keys, int64_t i) {
  for (int k = 0; k < h->nkeys; ++k) {
    int valid = col_valid(&keys[k], i);
    if (h->key_nullable[k]) { if (row[h->validity_off[k]] != (uint8_t)valid) return 0; if (!valid) continue; }
    const uint8_t* f = row + h->key_off[k];
    if (h->key_type[k] == ORC_T_STRING) {
      int64_t j = keys[k].is_scalar ? 0 : i;
      uint32_t len, slen; const uint8_t* p = view_bytes((const uint32_t*)keys[k].data + 4 * j, keys[k].buffers, &len);
      memcpy(&slen, f, 4); if (slen != len) return 0;
      uint64_t off; memcpy(&off, f + 4, 8);
      if (memcmp(h->strs + off, p, len)) return 0;
    } else {
      uint8_t tmp[32]; key_field(h, &keys[k], k, i, tmp, NULL);
      if (memcmp(tmp, f, (size_t)h->key_size[k])) return 0;
    }
  }
  return 1;
}
static uint8_t* append_row(orc_hashagg* h, const orc_col* keys, int64_t i, uint64_t hash) { /* payload.rs:361-486 */
  if (h->nrows == h->rows_cap) { h->rows_cap = h->rows_cap ? h->rows_cap * 2 : 1024; h->rows = (uint8_t*)realloc(h->rows, h->rows_cap * h->tuple_size); }
  if (h->states_len + h->state_size > h->states_cap) { h->states_cap = h->states_cap ? h->states_cap * 2 : (size_t)h->state_size * 1024; h->states = (uint8_t*)realloc(h->states, h->states_cap); }
  uint8_t* row = h->rows + h->nrows * h->tuple_size;
  memset(row, 0, (size_t)h->tuple_size);
  for (int k = 0; k < h->nkeys; ++k) {
    int valid = col_valid(&keys[k], i);
    if (h->key_nullable[k]) row[h->validity_off[k]] = (uint8_t)valid;
    if (valid) key_field(h, &keys[k], k, i, row + h->key_off[k], h);
  }
  memcpy(row + h->hash_off, &hash, 8);
  uint64_t soff = h->states_len; /* StateAddr as offset into the arena */
  if (h->naggs) {
    memcpy(row + h->state_ptr_off, &soff, 8);
    uint8_t* st = h->states + soff; memset(st, 0, (size_t)h->state_size); /* init_state */
    h->states_len += h->state_size;
  }
  h->nrows++;
  return (uint8_t*)(uintptr_t)(h->nrows); /* row index + 1 (rows may move on realloc) */
}

static void resize_index(orc_hashagg* h, size_t new_cap) { /* aggregate_hashtable.rs:463-490 */
  free(h->ctrls); free(h->pointers);
  index_alloc(h, new_cap);
  for (size_t r = 0; r < h->nrows; ++r) {
    uint64_t hash; memcpy(&hash, h->rows + r * h->tuple_size + h->hash_off, 8);
    size_t idx = probe_empty(h, hash);
    h->pointers[idx] = (uint8_t*)(uintptr_t)(r + 1);
  }
  h->count = h->nrows;
}

/* accumulate one row into one state (accumulate_keys: aggregate_unary.rs:208-222) */
/* i256 as four little-endian u64 (ethnum::i256 through T::U64Array, aggregate_sum.rs:183-216): wrapping add, then the range check
 * of DecimalSumState<true, i256>::add against +-(10^76 - 1) */
static void i256w_add(uint64_t* s, const uint64_t* v) {
  unsigned __int128 c = 0;
  for (int q = 0; q < 4; ++q) { c += (unsigned __int128)s[q] + v[q]; s[q] = (uint64_t)c; c >>= 64; }
}
static int i256w_out_of_range(const uint64_t* s) {
  static const uint64_t mx[4] = {0xFFFFFFFFFFFFFFFFULL, 0x7775A5F171950FFFULL, 0x0764B4ABE8652979ULL, 0x161BCCA7119915B5ULL}; /* 10^76 - 1 */
  uint64_t m[4] = {s[0], s[1], s[2], s[3]};
  if (s[3] >> 63) { unsigned __int128 c = 1; for (int q = 0; q < 4; ++q) { c += (uint64_t)~m[q]; m[q] = (uint64_t)c; c >>= 64; } }
  for (int q = 3; q >= 0; --q) if (m[q] != mx[q]) return m[q] > mx[q];
  return 0;
}
static int state_add(orc_hashagg* h, const orc_agg_desc* d, uint8_t* st, const orc_col* arg, int64_t i) {
  int valid = !arg || !arg->data || col_valid(arg, i);
  switch (d->kind) {
    case ORC_AGG_COUNT: if (valid) { uint64_t c; memcpy(&c, st, 8); c++; memcpy(st, &c, 8); } return 0;
    case ORC_AGG_SUM:
      if (!valid) return 0;
      if (agg_flag_off(d)) st[agg_flag_off(d)] = 1; /* set_flag(place, true), aggregate_null_adaptor.rs:447-452 */
      if (d->arg_type == ORC_T_DEC256) {
        uint64_t sum[4], v[4];
        memcpy(sum, st, 32); memcpy(v, (const uint8_t*)arg->data + 32 * (arg->is_scalar ? 0 : i), 32);
        i256w_add(sum, v);
        memcpy(st, sum, 32);
        return i256w_out_of_range(sum) ? 5 : 0;
      }
      if (d->arg_type == ORC_T_DEC128) { /* DecimalSumState::add aggregate_sum.rs:203-216 */
        i128 s; memcpy(&s, st, 16);
        s = (i128)((u128)s + (u128)((const i128*)arg->data)[arg->is_scalar ? 0 : i]);
        int check = d->arg_precision > 18;
        i128 mx = e10(38) - 1;
        memcpy(st, &s, 16);
        if (check && (s > mx || s < -mx)) return 5;
        return 0;
      } else {
        val v = load_val(arg, i);
        if (v.cls == 2) { double s; memcpy(&s, st, 8); s += v.f; memcpy(st, &s, 8); }
        else { uint64_t s; memcpy(&s, st, 8); s += v.cls == 0 ? (uint64_t)v.i : v.u; memcpy(st, &s, 8); }
        return 0;
      }
    default: { /* MIN / MAX over OrderedFloat / ints / Decimal128 (aggregate_min_max_any.rs: MinMaxAnyState<.., CmpMin / CmpMax>) */
      if (!valid) return 0;
      if (d->arg_type == ORC_T_STRING) {
        /* aggregate_min_max_any.rs:62-110 (StringState): Option<Vec<u8>>, replaced when the new value is smaller / larger in byte order
         * (then length: Rust's Ord on [u8]); the bytes are copied into the table (h->strs), the state keeps (offset, length) */
        uint64_t hs; memcpy(&hs, st + 16, 8);
        uint32_t len; const uint8_t* p = view_bytes((const uint32_t*)arg->data + 4 * (arg->is_scalar ? 0 : i), arg->buffers, &len);
        uint64_t coff, clen; memcpy(&coff, st, 8); memcpy(&clen, st + 8, 8);
        int take = !hs;
        if (!take) {
          size_t m = len < clen ? len : (size_t)clen;
          int c = m ? memcmp(p, h->strs + coff, m) : 0;
          if (c == 0) c = (len > clen) - (len < clen);
          take = d->kind == ORC_AGG_MIN ? c < 0 : c > 0;
        }
        if (take) {
          if (h->strs_len + len > h->strs_cap) { h->strs_cap = (h->strs_len + len) * 2 + 64; h->strs = (uint8_t*)realloc(h->strs, h->strs_cap); }
          if (len) memcpy(h->strs + h->strs_len, p, len);
          coff = h->strs_len; clen = len; h->strs_len += len;
          memcpy(st, &coff, 8); memcpy(st + 8, &clen, 8);
        }
        hs = 1; memcpy(st + 16, &hs, 8);
        return 0;
      }
      if (d->arg_type == ORC_T_DEC256) { /* MinMaxAnyDecimalState<i256>::add (aggregate_min_max_any_decimal.rs:61-76): change_if = new < / > current */
        uint64_t has256; memcpy(&has256, st + 32, 8);
        uint64_t v[4], cur[4];
        memcpy(v, (const uint8_t*)arg->data + 32 * (arg->is_scalar ? 0 : i), 32);
        memcpy(cur, st, 32);
        int c = i256w_cmp(v, cur);
        if (!has256 || (d->kind == ORC_AGG_MIN ? c < 0 : c > 0)) memcpy(st, v, 32);
        has256 = 1; memcpy(st + 32, &has256, 8);
        return 0;
      }
      if (d->arg_type == ORC_T_DEC128) {
        uint64_t has128; memcpy(&has128, st + 16, 8);
        i128 v128, cur128;
        memcpy(&v128, (const uint8_t*)arg->data + 16 * (arg->is_scalar ? 0 : i), 16);
        memcpy(&cur128, st, 16);
        int take128 = !has128 || (d->kind == ORC_AGG_MIN ? v128 < cur128 : v128 > cur128);
        if (take128) memcpy(st, &v128, 16);
        has128 = 1; memcpy(st + 16, &has128, 8);
        return 0;
      }
      uint64_t has; memcpy(&has, st + 8, 8);
      val v = load_val(arg, i);
      val cur; memset(&cur, 0, sizeof cur); cur.cls = v.cls; cur.bits = v.bits;
      if (v.cls == 2) memcpy(&cur.f, st, 8); else if (v.cls == 0) memcpy(&cur.i, st, 8); else memcpy(&cur.u, st, 8);
      int c;
      if (v.cls == 0) c = (v.i > cur.i) - (v.i < cur.i);
      else if (v.cls == 1) c = (v.u > cur.u) - (v.u < cur.u);
      else { int xn = v.f != v.f, yn = cur.f != cur.f; c = (xn || yn) ? xn - yn : (v.f > cur.f) - (v.f < cur.f); }
      int take = !has || (d->kind == ORC_AGG_MIN ? c < 0 : c > 0);
      if (take) { if (v.cls == 2) memcpy(st, &v.f, 8); else if (v.cls == 0) memcpy(st, &v.i, 8); else memcpy(st, &v.u, 8); }
      has = 1; memcpy(st + 8, &has, 8);
      return 0;
    }
  }
}

static int add_groups_inner(orc_hashagg* h, const orc_col* keys, const orc_col* args, int64_t start, int64_t rc) {
  static __thread uint64_t hashes[BATCH_SIZE];
  static __thread size_t slots[BATCH_SIZE], addr[BATCH_SIZE];
  static __thread int no_match[BATCH_SIZE], empty_v[BATCH_SIZE], cmp_v[BATCH_SIZE];
  /* group_hash_entries */
  for (int k = 0; k < h->nkeys; ++k)
    for (int64_t r = 0; r < rc; ++r) {
      uint64_t hv = hash_col_row(&keys[k], start + r);
      hashes[r] = k == 0 ? hv : (hashes[r] * NULL_HASH_VAL ^ hv);
    }
  /* probe_and_create aggregate_hashtable.rs:294-312 */
  if ((size_t)rc + h->count > (size_t)((double)h->capacity / LOAD_FACTOR)) {
    size_t nc = h->resize_count < 4 ? h->capacity * 4 : h->capacity * 2; /* :314-333 */
    h->resize_count++;
    resize_index(h, nc);
  }
  /* HashIndex::probe_and_create index.rs:148-216 */
  for (int64_t r = 0; r < rc; ++r) { no_match[r] = (int)r; slots[r] = hashes[r] & h->mask; }
  int64_t remaining = rc;
  while (remaining > 0) {
    int n_new = 0, n_cmp = 0, n_nomatch = 0;
    for (int64_t t = 0; t < remaining; ++t) {
      int row = no_match[t], is_new;
      slots[row] = find_or_insert(h, slots[row], hashes[row], &is_new);
      if (is_new) empty_v[n_new++] = row; else cmp_v[n_cmp++] = row;
    }
    for (int t = 0; t < n_new; ++t) { /* adapter.append_rows */
      int row = empty_v[t];
      addr[row] = (size_t)(uintptr_t)append_row(h, keys, start + row, hashes[row]);
      h->pointers[slots[row]] = (uint8_t*)(uintptr_t)addr[row];
    }
    for (int t = 0; t < n_cmp; ++t) { /* adapter.compare */
      int row = cmp_v[t];
      addr[row] = (size_t)(uintptr_t)h->pointers[slots[row]];
      const uint8_t* stored = h->rows + (addr[row] - 1) * h->tuple_size;
      if (!row_match(h, stored, keys, start + row)) no_match[n_nomatch++] = row;
    }
    for (int t = 0; t < n_nomatch; ++t) slots[no_match[t]] = (slots[no_match[t]] + 1) & h->mask;
    h->count += (size_t)n_new;
    remaining = n_nomatch;
  }
  /* accumulate_keys per aggregate */
  int rcode = 0;
  for (int a = 0; a < h->naggs; ++a)
    for (int64_t r = 0; r < rc; ++r) {
      uint64_t soff; memcpy(&soff, h->rows + (addr[r] - 1) * h->tuple_size + h->state_ptr_off, 8);
      int e = state_add(h, &h->aggs[a], h->states + soff + h->state_off[a], args ? &args[a] : NULL, start + r);
      if (e) rcode = e;
    }
  return rcode;
}

int orc_hashagg_add_block(orc_hashagg* h, const orc_col* keys, const orc_col* args, int64_t n) { /* add_groups :168-207 */
  int rc = 0;
  for (int64_t s = 0; s < n; s += BATCH_SIZE) {
    int64_t m = n - s < BATCH_SIZE ? n - s : BATCH_SIZE;
    int e = add_groups_inner(h, keys, args, s, m);
    if (e) rc = e;
  }
  return rc;
}
int64_t orc_hashagg_num_groups(orc_hashagg* h) { return (int64_t)h->nrows; }

/* materialise the groups of `h` as columns (payload_flush.rs:50-240) */
typedef struct { void* keys[MAXK]; uint8_t* valid[MAXK]; void* bufs[MAXK]; } flushed;

int orc_hashagg_result_nullable(orc_hashagg* h, void* const* out_keys, uint8_t* const* out_key_valid,
                                void* const* out_aggs, uint8_t* const* out_agg_valid, uint64_t* out_hashes);
int orc_hashagg_result(orc_hashagg* h, void* const* out_keys, uint8_t* const* out_key_valid,
                       void* const* out_aggs, uint64_t* out_hashes) {
  return orc_hashagg_result_nullable(h, out_keys, out_key_valid, out_aggs, NULL, out_hashes);
}
/* out_agg_valid[a] (one byte per group, may be NULL): 0 where sum/min/max over a Nullable argument is NULL */
int orc_hashagg_result_nullable(orc_hashagg* h, void* const* out_keys, uint8_t* const* out_key_valid,
                                void* const* out_aggs, uint8_t* const* out_agg_valid, uint64_t* out_hashes) {
  int rc = 0;
  for (size_t r = 0; r < h->nrows; ++r) {
    const uint8_t* row = h->rows + r * h->tuple_size;
    for (int k = 0; k < h->nkeys; ++k) {
      int valid = h->key_nullable[k] ? row[h->validity_off[k]] : 1;
      if (out_key_valid && out_key_valid[k]) out_key_valid[k][r] = (uint8_t)valid;
      if (!out_keys || !out_keys[k]) continue;
      const uint8_t* f = row + h->key_off[k];
      if (h->key_type[k] == ORC_T_STRING) {
        uint8_t* v = (uint8_t*)out_keys[k] + 16 * r; memset(v, 0, 16);
        uint32_t len; memcpy(&len, f, 4); uint64_t off; memcpy(&off, f + 4, 8);
        memcpy(v, &len, 4);
        if (valid && len <= 12) memcpy(v + 4, h->strs + off, len);
      } else {
        memcpy((uint8_t*)out_keys[k] + r * h->key_size[k], f, (size_t)h->key_size[k]);
      }
    }
    if (out_hashes) memcpy(&out_hashes[r], row + h->hash_off, 8);
    if (!h->naggs) continue;
    uint64_t soff; memcpy(&soff, row + h->state_ptr_off, 8);
    for (int a = 0; a < h->naggs; ++a) {
      const uint8_t* st = h->states + soff + h->state_off[a];
      const orc_agg_desc* d = &h->aggs[a];
      if (out_agg_valid && out_agg_valid[a]) {
        uint8_t ok = 1;
        if (agg_flag_off(d)) ok = st[agg_flag_off(d)];
        else if ((d->kind == ORC_AGG_MIN || d->kind == ORC_AGG_MAX) && d->arg_nullable) { uint64_t hs; memcpy(&hs, st + mm_has_off(d), 8); ok = hs != 0; }
        out_agg_valid[a][r] = ok;
      }
      if (!out_aggs || !out_aggs[a]) continue;
      if (d->kind == ORC_AGG_SUM && d->arg_type == ORC_T_DEC256) memcpy((uint8_t*)out_aggs[a] + 32 * r, st, 32);
      else if (d->kind == ORC_AGG_SUM && d->arg_type == ORC_T_DEC128) memcpy((uint8_t*)out_aggs[a] + 16 * r, st, 16);
      else if ((d->kind == ORC_AGG_MIN || d->kind == ORC_AGG_MAX) && d->arg_type == ORC_T_STRING) {
        /* 16 bytes per group: u32 length, then the bytes when they fit 12, else u32 0, u64 offset into the table's bytes (orc_hashagg_bytes) */
        uint8_t* v = (uint8_t*)out_aggs[a] + 16 * r; memset(v, 0, 16);
        uint64_t hs, coff, clen; memcpy(&hs, st + 16, 8); memcpy(&coff, st, 8); memcpy(&clen, st + 8, 8);
        if (hs) {
          uint32_t l32 = (uint32_t)clen; memcpy(v, &l32, 4);
          if (clen <= 12) memcpy(v + 4, h->strs + coff, (size_t)clen); else memcpy(v + 8, &coff, 8);
        }
      }
      else if ((d->kind == ORC_AGG_MIN || d->kind == ORC_AGG_MAX) && d->arg_type == ORC_T_DEC256) {
        uint64_t hs; memcpy(&hs, st + 32, 8);
        if (hs) memcpy((uint8_t*)out_aggs[a] + 32 * r, st, 32); else memset((uint8_t*)out_aggs[a] + 32 * r, 0, 32); /* push_default() */
      }
      else if ((d->kind == ORC_AGG_MIN || d->kind == ORC_AGG_MAX) && d->arg_type == ORC_T_DEC128) {
        uint64_t hs; memcpy(&hs, st + 16, 8);
        if (hs) memcpy((uint8_t*)out_aggs[a] + 16 * r, st, 16); else memset((uint8_t*)out_aggs[a] + 16 * r, 0, 16); /* push_default() */
      } else if (d->kind == ORC_AGG_MIN || d->kind == ORC_AGG_MAX) {
        int sz = t_size(d->arg_type);
        val v; memset(&v, 0, sizeof v); v.cls = t_cls(d->arg_type);
        if (v.cls == 2) memcpy(&v.f, st, 8); else if (v.cls == 0) memcpy(&v.i, st, 8); else memcpy(&v.u, st, 8);
        store_val(out_aggs[a], d->arg_type, (int64_t)r, v); (void)sz;
      } else memcpy((uint8_t*)out_aggs[a] + 8 * r, st, 8);
    }
  }
  return rc;
}

/* the table's byte store (long string keys, min / max over String values): results refer to it by offset */
const uint8_t* orc_hashagg_bytes(orc_hashagg* h, int64_t* out_len) { if (out_len) *out_len = (int64_t)h->strs_len; return h->strs; }

/* combine_payload aggregate_hashtable.rs:349-380: re-probe the other table's groups and merge states */
int orc_hashagg_combine(orc_hashagg* dst, orc_hashagg* src) {
  int rc = 0;
  size_t n = src->nrows;
  if (!n) return 0;
  /* flush src rows to key columns */
  orc_col keys[MAXK]; memset(keys, 0, sizeof keys);
  void* kb[MAXK]; uint8_t* kv[MAXK]; uint8_t* kbits[MAXK];
  for (int k = 0; k < src->nkeys; ++k) {
    int es = src->key_type[k] == ORC_T_STRING ? 16 : src->key_size[k];
    kb[k] = calloc(n + 8, (size_t)es); kv[k] = (uint8_t*)calloc(n + 8, 1); kbits[k] = (uint8_t*)calloc((n + 7) / 8 + 8, 1);
  }
  orc_hashagg_result(src, kb, kv, NULL, NULL);
  for (int k = 0; k < src->nkeys; ++k) {
    keys[k].type = src->key_type[k]; keys[k].data = kb[k];
    if (src->key_type[k] == ORC_T_BOOL) { /* bytes -> bitmap */
      uint8_t* bm = (uint8_t*)calloc((n + 7) / 8 + 8, 1);
      for (size_t r = 0; r < n; ++r) if (((uint8_t*)kb[k])[r]) bm[r >> 3] |= (uint8_t)(1u << (r & 7));
      free(kb[k]); kb[k] = bm; keys[k].data = bm;
    }
    if (src->key_nullable[k]) {
      for (size_t r = 0; r < n; ++r) if (kv[k][r]) kbits[k][r >> 3] |= (uint8_t)(1u << (r & 7));
      keys[k].validity = kbits[k];
    }
  }
  /* insert groups without accumulating, then merge states (batch_merge_states) */
  int saved = dst->naggs;
  for (size_t s = 0; s < n; s += BATCH_SIZE) {
    size_t m = n - s < BATCH_SIZE ? n - s : BATCH_SIZE;
    /* probe with naggs temporarily 0 would skip state allocation, so probe normally with no-op args */
    static __thread size_t addrs[BATCH_SIZE];
    /* replicate add_groups_inner's probe, capturing addresses */
    dst->naggs = saved;
    {
      static __thread uint64_t hashes[BATCH_SIZE];
      static __thread size_t slots[BATCH_SIZE];
      static __thread int no_match[BATCH_SIZE], empty_v[BATCH_SIZE], cmp_v[BATCH_SIZE];
      for (size_t r = 0; r < m; ++r) memcpy(&hashes[r], src->rows + (s + r) * src->tuple_size + src->hash_off, 8);
      if (m + dst->count > (size_t)((double)dst->capacity / LOAD_FACTOR)) {
        size_t nc = dst->resize_count < 4 ? dst->capacity * 4 : dst->capacity * 2; dst->resize_count++; resize_index(dst, nc);
      }
      for (size_t r = 0; r < m; ++r) { no_match[r] = (int)r; slots[r] = hashes[r] & dst->mask; }
      size_t remaining = m;
      while (remaining > 0) {
        int n_new = 0, n_cmp = 0, n_nomatch = 0;
        for (size_t t = 0; t < remaining; ++t) {
          int row = no_match[t], is_new;
          slots[row] = find_or_insert(dst, slots[row], hashes[row], &is_new);
          if (is_new) empty_v[n_new++] = row; else cmp_v[n_cmp++] = row;
        }
        for (int t = 0; t < n_new; ++t) {
          int row = empty_v[t];
          addrs[row] = (size_t)(uintptr_t)append_row(dst, keys, (int64_t)(s + row), hashes[row]);
          dst->pointers[slots[row]] = (uint8_t*)(uintptr_t)addrs[row];
        }
        for (int t = 0; t < n_cmp; ++t) {
          int row = cmp_v[t];
          addrs[row] = (size_t)(uintptr_t)dst->pointers[slots[row]];
          if (!row_match(dst, dst->rows + (addrs[row] - 1) * dst->tuple_size, keys, (int64_t)(s + row))) no_match[n_nomatch++] = row;
        }
        for (int t = 0; t < n_nomatch; ++t) slots[no_match[t]] = (slots[no_match[t]] + 1) & dst->mask;
        dst->count += (size_t)n_new;
        remaining = (size_t)n_nomatch;
      }
    }
    for (size_t r = 0; r < m; ++r) {
      uint64_t so_s, so_d;
      memcpy(&so_s, src->rows + (s + r) * src->tuple_size + src->state_ptr_off, 8);
      memcpy(&so_d, dst->rows + (addrs[r] - 1) * dst->tuple_size + dst->state_ptr_off, 8);
      for (int a = 0; a < dst->naggs; ++a) {
        uint8_t* d = dst->states + so_d + dst->state_off[a];
        const uint8_t* sp = src->states + so_s + src->state_off[a];
        const orc_agg_desc* ad = &dst->aggs[a];
        if (ad->kind == ORC_AGG_COUNT) { uint64_t x, y; memcpy(&x, d, 8); memcpy(&y, sp, 8); x += y; memcpy(d, &x, 8); }
        else if (ad->kind == ORC_AGG_SUM) {
          if (agg_flag_off(ad)) { /* merge_states of the adaptor (aggregate_null_adaptor.rs:613-622): nothing to do while rhs saw no value */
            if (!sp[agg_flag_off(ad)]) continue;
            d[agg_flag_off(ad)] = 1;
          }
          if (ad->arg_type == ORC_T_DEC256) {
            uint64_t x[4], y[4]; memcpy(x, d, 32); memcpy(y, sp, 32); i256w_add(x, y); memcpy(d, x, 32);
            if (i256w_out_of_range(x)) rc = 5;
          } else
          if (ad->arg_type == ORC_T_DEC128) { /* DecimalSumState::merge -> add (with the overflow check) */
            i128 x, y; memcpy(&x, d, 16); memcpy(&y, sp, 16); x = (i128)((u128)x + (u128)y); memcpy(d, &x, 16);
            i128 mx = e10(38) - 1; if (ad->arg_precision > 18 && (x > mx || x < -mx)) rc = 5;
          } else if (t_cls(ad->arg_type) == 2) { double x, y; memcpy(&x, d, 8); memcpy(&y, sp, 8); x += y; memcpy(d, &x, 8); }
          else { uint64_t x, y; memcpy(&x, d, 8); memcpy(&y, sp, 8); x += y; memcpy(d, &x, 8); }
        } else if (ad->arg_type == ORC_T_STRING) {   /* StringState::merge: the other side's value as one more candidate */
          uint64_t hs, coff, clen; memcpy(&hs, sp + 16, 8); memcpy(&coff, sp, 8); memcpy(&clen, sp + 8, 8);
          if (hs) {
            uint8_t view[16]; memset(view, 0, 16);
            uint32_t l32 = (uint32_t)clen; memcpy(view, &l32, 4);
            const void* bufs[1] = {src->strs + coff};
            if (clen <= 12) memcpy(view + 4, src->strs + coff, (size_t)clen); else memcpy(view + 4, src->strs + coff, 4);
            orc_col tmp; memset(&tmp, 0, sizeof tmp); tmp.type = ORC_T_STRING; tmp.is_scalar = 1; tmp.data = view; tmp.buffers = bufs;
            state_add(dst, ad, d, &tmp, 0);
          }
        } else if (ad->arg_type == ORC_T_DEC128 || ad->arg_type == ORC_T_DEC256) {
          uint64_t hs; memcpy(&hs, sp + mm_has_off(ad), 8);
          if (hs) { orc_col tmp; memset(&tmp, 0, sizeof tmp); tmp.type = ad->arg_type; tmp.is_scalar = 1; tmp.data = sp; state_add(dst, ad, d, &tmp, 0); }
        } else {
          uint64_t hs; memcpy(&hs, sp + 8, 8);
          if (hs) { orc_col tmp; memset(&tmp, 0, sizeof tmp); tmp.type = ad->arg_type; tmp.is_scalar = 1;
            uint8_t buf[8]; val v; memset(&v, 0, sizeof v); v.cls = t_cls(ad->arg_type);
            if (v.cls == 2) memcpy(&v.f, sp, 8); else if (v.cls == 0) memcpy(&v.i, sp, 8); else memcpy(&v.u, sp, 8);
            store_val(buf, ad->arg_type, 0, v); tmp.data = buf; state_add(dst, ad, d, &tmp, 0); }
        }
      }
    }
  }
  for (int k = 0; k < src->nkeys; ++k) { free(kb[k]); free(kv[k]); free(kbits[k]); }
  return rc;
}

/* the strings of key column k of every group, any length (the inline-view result of orc_hashagg_result holds <= 12 bytes):
 * out_offsets[g] .. out_offsets[g + 1] delimit group g's bytes in out_bytes (NULL keys: empty). Returns the total bytes
 * (call with out_bytes == NULL to size the buffer). */
int64_t orc_hashagg_key_strings(orc_hashagg* h, int k, int64_t* out_offsets, uint8_t* out_bytes) {
  int64_t total = 0;
  for (size_t r = 0; r < h->nrows; ++r) {
    const uint8_t* row = h->rows + r * h->tuple_size;
    int valid = h->key_nullable[k] ? row[h->validity_off[k]] : 1;
    uint32_t len = 0; uint64_t off = 0;
    if (valid) { memcpy(&len, row + h->key_off[k], 4); memcpy(&off, row + h->key_off[k] + 4, 8); }
    if (out_offsets) out_offsets[r] = total;
    if (out_bytes && len) memcpy(out_bytes + total, h->strs + off, len);
    total += len;
  }
  if (out_offsets) out_offsets[h->nrows] = total;
  return total;
}

/* ---- serialized-state block: Payload::aggregate_flush (payload_flush.rs:151-181) ----------------------------------
 * entries = per aggregate the columns of its serialize_type() (a Tuple builder per aggregate: flattened here), then the
 * group columns. serialize_type / batch_serialize / batch_merge per function:
 *   count   [UInt64]                         aggregate_count.rs:170-213
 *   sum     [result type]                    aggregate_sum.rs:155-181 (numbers), :281-312 (decimal)
 *   min/max [Boolean, T] = Option<T>         aggregate_min_max_any.rs:315-367 (value = default when None)
 *   Nullable argument (sum/min/max): nested fields + trailing Boolean flag   aggregate_null_adaptor.rs:508-600 */
static int sum_result_type(int t) {
  switch (t) {
    case ORC_T_I8: case ORC_T_I16: case ORC_T_I32: case ORC_T_I64: return ORC_T_I64;
    case ORC_T_U8: case ORC_T_U16: case ORC_T_U32: case ORC_T_U64: return ORC_T_U64;
    case ORC_T_F32: case ORC_T_F64: return ORC_T_F64;
    default: return t; /* decimals keep their storage class */
  }
}
int orc_hashagg_state_fields(orc_hashagg* h, int32_t* types, int32_t* agg_of) {
  int f = 0;
  for (int a = 0; a < h->naggs; ++a) {
    const orc_agg_desc* d = &h->aggs[a];
#define PUT(T) do { if (types) types[f] = (T); if (agg_of) agg_of[f] = a; ++f; } while (0)
    if (d->kind == ORC_AGG_COUNT) PUT(ORC_T_U64);
    else if (d->kind == ORC_AGG_SUM) { PUT(sum_result_type(d->arg_type)); if (d->arg_nullable) PUT(ORC_T_BOOL); }
    else { PUT(ORC_T_BOOL); PUT(d->arg_type); if (d->arg_nullable) PUT(ORC_T_BOOL); }
#undef PUT
  }
  return f;
}
/* out_fields[f]: n elements of the field's type; Boolean fields ONE BYTE per row (0 / 1). Keys / validity / hashes as
 * orc_hashagg_result. Row order = payload order. */
int orc_hashagg_flush_state_block(orc_hashagg* h, void* const* out_keys, uint8_t* const* out_key_valid,
                                  void* const* out_fields, uint64_t* out_hashes) {
  int rc = orc_hashagg_result(h, out_keys, out_key_valid, NULL, out_hashes);
  for (size_t r = 0; r < h->nrows; ++r) {
    uint64_t soff; memcpy(&soff, h->rows + r * h->tuple_size + h->state_ptr_off, 8);
    int f = 0;
    for (int a = 0; a < h->naggs; ++a) {
      const orc_agg_desc* d = &h->aggs[a];
      const uint8_t* st = h->states + soff + h->state_off[a];
      if (d->kind == ORC_AGG_COUNT) { memcpy((uint8_t*)out_fields[f] + 8 * r, st, 8); ++f; }
      else if (d->kind == ORC_AGG_SUM) {
        if (d->arg_type == ORC_T_DEC256) memcpy((uint8_t*)out_fields[f] + 32 * r, st, 32);
        else if (d->arg_type == ORC_T_DEC128) memcpy((uint8_t*)out_fields[f] + 16 * r, st, 16); else memcpy((uint8_t*)out_fields[f] + 8 * r, st, 8);
        ++f;
        if (d->arg_nullable) { ((uint8_t*)out_fields[f])[r] = st[agg_flag_off(d)]; ++f; }
      } else if (d->arg_type == ORC_T_STRING) {
        /* MinMaxStringState: serialize_type = [Nullable(String)] (aggregate_min_max_any.rs:163-181) — here its two buffers as two fields:
         * the validity (has a value) and the values, 16 bytes per group in orc_hashagg_result's form (u32 length, the bytes when they fit
         * 12, else at +8 the u64 offset into orc_hashagg_bytes) */
        uint64_t hs, coff, clen; memcpy(&hs, st + 16, 8); memcpy(&coff, st, 8); memcpy(&clen, st + 8, 8);
        ((uint8_t*)out_fields[f])[r] = hs != 0; ++f;
        uint8_t* v = (uint8_t*)out_fields[f] + 16 * r; memset(v, 0, 16);
        if (hs) {
          uint32_t l32 = (uint32_t)clen; memcpy(v, &l32, 4);
          if (clen <= 12) memcpy(v + 4, h->strs + coff, (size_t)clen); else memcpy(v + 8, &coff, 8);
        }
        ++f;
        if (d->arg_nullable) { ((uint8_t*)out_fields[f])[r] = hs != 0; ++f; }
      } else if (d->arg_type == ORC_T_DEC256) { /* [Nullable(Decimal256)] (aggregate_min_max_any_decimal.rs:140-147): validity, values */
        uint64_t hs; memcpy(&hs, st + 32, 8);
        ((uint8_t*)out_fields[f])[r] = hs != 0; ++f;
        if (hs) memcpy((uint8_t*)out_fields[f] + 32 * r, st, 32); else memset((uint8_t*)out_fields[f] + 32 * r, 0, 32);
        ++f;
        if (d->arg_nullable) { ((uint8_t*)out_fields[f])[r] = hs != 0; ++f; }
      } else if (d->arg_type == ORC_T_DEC128) {
        uint64_t hs; memcpy(&hs, st + 16, 8);
        ((uint8_t*)out_fields[f])[r] = hs != 0; ++f;
        if (hs) memcpy((uint8_t*)out_fields[f] + 16 * r, st, 16); else memset((uint8_t*)out_fields[f] + 16 * r, 0, 16);
        ++f;
        if (d->arg_nullable) { ((uint8_t*)out_fields[f])[r] = hs != 0; ++f; }
      } else {
        uint64_t hs; memcpy(&hs, st + 8, 8);
        ((uint8_t*)out_fields[f])[r] = hs != 0; ++f;
        val v; memset(&v, 0, sizeof v); v.cls = t_cls(d->arg_type);
        if (hs) { if (v.cls == 2) memcpy(&v.f, st, 8); else if (v.cls == 0) memcpy(&v.i, st, 8); else memcpy(&v.u, st, 8); } /* else push_default() */
        store_val(out_fields[f], d->arg_type, (int64_t)r, v); ++f;
        if (d->arg_nullable) { ((uint8_t*)out_fields[f])[r] = hs != 0; ++f; }
      }
    }
  }
  return rc;
}
/* TransformDeserializer + batch_merge: probe / create the block's groups (probe_and_create, no accumulation), then
 * merge every field column into the states. `fields[f]`: orc_col of the field's type; Boolean fields are LSB-first
 * bitmaps (ORC_T_BOOL columns). */
int orc_hashagg_merge_state_block(orc_hashagg* h, const orc_col* keys, const orc_col* fields, int64_t n) {
  int rc = 0;
  for (int64_t s0 = 0; s0 < n; s0 += BATCH_SIZE) {
    size_t m = (size_t)(n - s0 < BATCH_SIZE ? n - s0 : BATCH_SIZE);
    static __thread uint64_t hashes[BATCH_SIZE];
    static __thread size_t slots[BATCH_SIZE], addrs[BATCH_SIZE];
    static __thread int no_match[BATCH_SIZE], empty_v[BATCH_SIZE], cmp_v[BATCH_SIZE];
    for (int k = 0; k < h->nkeys; ++k)
      for (size_t r = 0; r < m; ++r) {
        uint64_t hv = hash_col_row(&keys[k], s0 + (int64_t)r);
        hashes[r] = k == 0 ? hv : (hashes[r] * NULL_HASH_VAL ^ hv);
      }
    if (m + h->count > (size_t)((double)h->capacity / LOAD_FACTOR)) {
      size_t nc = h->resize_count < 4 ? h->capacity * 4 : h->capacity * 2; h->resize_count++; resize_index(h, nc);
    }
    for (size_t r = 0; r < m; ++r) { no_match[r] = (int)r; slots[r] = hashes[r] & h->mask; }
    size_t remaining = m;
    while (remaining > 0) {
      int n_new = 0, n_cmp = 0, n_nomatch = 0;
      for (size_t t = 0; t < remaining; ++t) {
        int row = no_match[t], is_new;
        slots[row] = find_or_insert(h, slots[row], hashes[row], &is_new);
        if (is_new) empty_v[n_new++] = row; else cmp_v[n_cmp++] = row;
      }
      for (int t = 0; t < n_new; ++t) {
        int row = empty_v[t];
        addrs[row] = (size_t)(uintptr_t)append_row(h, keys, s0 + row, hashes[row]);
        h->pointers[slots[row]] = (uint8_t*)(uintptr_t)addrs[row];
      }
      for (int t = 0; t < n_cmp; ++t) {
        int row = cmp_v[t];
        addrs[row] = (size_t)(uintptr_t)h->pointers[slots[row]];
        if (!row_match(h, h->rows + (addrs[row] - 1) * h->tuple_size, keys, s0 + row)) no_match[n_nomatch++] = row;
      }
      for (int t = 0; t < n_nomatch; ++t) slots[no_match[t]] = (slots[no_match[t]] + 1) & h->mask;
      h->count += (size_t)n_new;
      remaining = (size_t)n_nomatch;
    }
    for (size_t r = 0; r < m; ++r) {
      const int64_t i = s0 + (int64_t)r;
      uint64_t so_d; memcpy(&so_d, h->rows + (addrs[r] - 1) * h->tuple_size + h->state_ptr_off, 8);
      int f = 0;
      for (int a = 0; a < h->naggs; ++a) {
        uint8_t* d = h->states + so_d + h->state_off[a];
        const orc_agg_desc* ad = &h->aggs[a];
        if (ad->kind == ORC_AGG_COUNT) { /* aggregate_count.rs:188-213: count += other */
          uint64_t x; memcpy(&x, d, 8); x += ((const uint64_t*)fields[f].data)[fields[f].is_scalar ? 0 : i]; memcpy(d, &x, 8); ++f;
        } else if (ad->kind == ORC_AGG_SUM) {
          const orc_col* vf = &fields[f]; ++f;
          int seen = 1;
          if (ad->arg_nullable) { seen = bit_get((const uint8_t*)fields[f].data, fields[f].is_scalar ? 0 : i); ++f; }
          if (!seen) continue; /* the adaptor's batch_merge filters the nested merge on the flag (:542-575) */
          if (agg_flag_off(ad)) d[agg_flag_off(ad)] = 1;
          int64_t j = vf->is_scalar ? 0 : i;
          if (ad->arg_type == ORC_T_DEC256) {
            uint64_t x[4], y[4]; memcpy(x, d, 32); memcpy(y, (const uint8_t*)vf->data + 32 * j, 32); i256w_add(x, y); memcpy(d, x, 32);
            if (i256w_out_of_range(x)) rc = 5;
          } else
          if (ad->arg_type == ORC_T_DEC128) { /* DecimalSumState batch_merge -> merge -> add with the overflow check */
            i128 x, y; memcpy(&x, d, 16); memcpy(&y, (const uint8_t*)vf->data + 16 * j, 16); x = (i128)((u128)x + (u128)y); memcpy(d, &x, 16);
            i128 mx = e10(38) - 1; if (ad->arg_precision > 18 && (x > mx || x < -mx)) rc = 5;
          } else if (t_cls(ad->arg_type) == 2) { double x, y; memcpy(&x, d, 8); memcpy(&y, (const uint8_t*)vf->data + 8 * j, 8); x += y; memcpy(d, &x, 8); }
          else { uint64_t x, y; memcpy(&x, d, 8); memcpy(&y, (const uint8_t*)vf->data + 8 * j, 8); x += y; memcpy(d, &x, 8); }
        } else { /* MinMaxAnyState batch_merge2: rhs = flag.then_some(value); state.merge(rhs) */
          int has = bit_get((const uint8_t*)fields[f].data, fields[f].is_scalar ? 0 : i); ++f;
          const orc_col* vf = &fields[f]; ++f;
          if (ad->arg_nullable) { has = has && bit_get((const uint8_t*)fields[f].data, fields[f].is_scalar ? 0 : i); ++f; }
          if (has) { orc_col tmp = *vf; tmp.validity = NULL; state_add(h, ad, d, &tmp, i); }
        }
      }
    }
  }
  return rc;
}

/* The reference's own HashIndex test (hash_index/index.rs:252-404) restated: a table of `capacity` slots already holds the
 * payload rows (key, hash, value) (probe_slot_and_set), then probe_and_create runs over the incoming (key, hash) pairs with
 * an adapter that appends (key, hash, key + 20) for new rows and compares keys for tag matches. out_values[i] = the value of
 * the row incoming[i] resolved to; returns the number of new rows. */
int orc_hash_index_case(int capacity, const uint64_t* in_keys, const uint64_t* in_hashes, int n, const uint64_t* pay_keys,
                        const uint64_t* pay_hashes, const uint64_t* pay_values, int m, uint64_t* out_values) {
  int32_t kt[1] = {ORC_T_U64};
  orc_hashagg* h = orc_hashagg_create(kt, NULL, 1, NULL, 0);
  free(h->ctrls); free(h->pointers);
  index_alloc(h, (size_t)capacity);
  uint64_t vals[64];  /* payload values by row (the adapter's payload vector) */
  int nv = 0;
  orc_col kc; memset(&kc, 0, sizeof kc); kc.type = ORC_T_U64;
  for (int r = 0; r < m; ++r) { /* init_hash_index: probe_slot_and_set */
    kc.data = &pay_keys[r];
    uint8_t* addr = append_row(h, &kc, 0, pay_hashes[r]);
    size_t idx = probe_empty(h, pay_hashes[r]);
    h->pointers[idx] = addr; h->count++;
    vals[nv++] = pay_values[r];
  }
  size_t slots[BATCH_SIZE], addr[BATCH_SIZE];
  int no_match[BATCH_SIZE], empty_v[BATCH_SIZE], cmp_v[BATCH_SIZE];
  for (int r = 0; r < n; ++r) { no_match[r] = r; slots[r] = in_hashes[r] & h->mask; }
  int remaining = n, total_new = 0;
  while (remaining > 0) { /* probe_and_create index.rs:148-216 */
    int n_new = 0, n_cmp = 0, n_nomatch = 0;
    for (int t = 0; t < remaining; ++t) {
      int row = no_match[t], is_new;
      slots[row] = find_or_insert(h, slots[row], in_hashes[row], &is_new);
      if (is_new) empty_v[n_new++] = row; else cmp_v[n_cmp++] = row;
    }
    for (int t = 0; t < n_new; ++t) { /* adapter.append_rows: value = key + 20 */
      int row = empty_v[t];
      kc.data = &in_keys[row];
      addr[row] = (size_t)(uintptr_t)append_row(h, &kc, 0, in_hashes[row]);
      h->pointers[slots[row]] = (uint8_t*)(uintptr_t)addr[row];
      vals[nv++] = in_keys[row] + 20;
    }
    for (int t = 0; t < n_cmp; ++t) { /* adapter.compare */
      int row = cmp_v[t];
      addr[row] = (size_t)(uintptr_t)h->pointers[slots[row]];
      uint64_t key; memcpy(&key, h->rows + (addr[row] - 1) * h->tuple_size + h->key_off[0], 8);
      if (key != in_keys[row]) no_match[n_nomatch++] = row;
    }
    for (int t = 0; t < n_nomatch; ++t) slots[no_match[t]] = (slots[no_match[t]] + 1) & h->mask;
    h->count += (size_t)n_new; total_new += n_new;
    remaining = n_nomatch;
  }
  for (int r = 0; r < n; ++r) out_values[r] = vals[addr[r] - 1];
  orc_hashagg_destroy(h);
  return total_new;
}

void orc_hashagg_destroy(orc_hashagg* h) {
  if (!h) return;
  free(h->ctrls); free(h->pointers); free(h->rows); free(h->states); free(h->strs); free(h);
}

/* ------------------------------------------------------------------------ */
/* TPC-H Q1: filter -> take -> maps -> partial agg per thread, final merge     */
/* (physical_aggregate_partial.rs:194-234, transform_aggregate_final.rs:160-175) */
/* ------------------------------------------------------------------------ */
typedef struct {
  const int64_t *qty, *price, *disc, *tax; const uint8_t *rf, *ls; const int32_t* sd;
  int32_t cutoff; int64_t n, block_rows; int tid, nthreads; orc_hashagg* ht; int rc;
} q1_worker;

static orc_hashagg* q1_table(void) {
  int32_t kt[2] = {ORC_T_STRING, ORC_T_STRING};
  orc_agg_desc ag[6]; memset(ag, 0, sizeof ag);
  ag[0].kind = ORC_AGG_SUM; ag[0].arg_type = ORC_T_DEC64; ag[0].arg_precision = 15; ag[0].arg_scale = 2;
  ag[1] = ag[0];
  ag[2].kind = ORC_AGG_SUM; ag[2].arg_type = ORC_T_DEC128; ag[2].arg_precision = 31; ag[2].arg_scale = 4;
  ag[3].kind = ORC_AGG_SUM; ag[3].arg_type = ORC_T_DEC128; ag[3].arg_precision = 38; ag[3].arg_scale = 6;
  ag[4] = ag[0];
  ag[5].kind = ORC_AGG_COUNT;
  return orc_hashagg_create(kt, NULL, 2, ag, 6);
}

static void* q1_work(void* p) {
  q1_worker* w = (q1_worker*)p;
  int64_t B = w->block_rows;
  uint8_t* bm = (uint8_t*)malloc((size_t)(B + 63) / 8 + 8);
  uint32_t* sel = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)B);
  int64_t *tq = malloc(8 * (size_t)B), *tp = malloc(8 * (size_t)B), *td = malloc(8 * (size_t)B), *tt = malloc(8 * (size_t)B);
  uint8_t *trf = malloc(16 * (size_t)B), *tls = malloc(16 * (size_t)B);
  int64_t *omd = malloc(8 * (size_t)B), *opt = malloc(8 * (size_t)B);
  i128 *dp = malloc(16 * (size_t)B), *ch = malloc(16 * (size_t)B);
  int64_t nblocks = (w->n + B - 1) / B;
  for (int64_t b = w->tid; b < nblocks; b += w->nthreads) {
    int64_t s = b * B, m = w->n - s < B ? w->n - s : B;
    /* TransformFilter: l_shipdate <= cutoff */
    orc_col sd = {ORC_T_DATE, 0, w->sd + s, NULL, 0, NULL, 0, 0, 0, {0, 0}};
    orc_col cut = {ORC_T_DATE, 1, &w->cutoff, NULL, 0, NULL, 0, 0, 0, {0, 0}};
    orc_cmp(ORC_CMP_LTE, &sd, &cut, m, bm);
    int64_t k = orc_filter_select(bm, 0, m, sel);
    orc_take(w->qty + s, 8, sel, k, tq); orc_take(w->price + s, 8, sel, k, tp);
    orc_take(w->disc + s, 8, sel, k, td); orc_take(w->tax + s, 8, sel, k, tt);
    orc_take(w->rf + 16 * s, 16, sel, k, trf); orc_take(w->ls + 16 * s, 16, sel, k, tls);
    /* CompoundBlockOperator: one materialised column per call node */
    uint8_t one = 1;
    orc_col c_one = {ORC_T_U8, 1, &one, NULL, 0, NULL, 0, 0, 0, {0, 0}};
    orc_col c_disc = {ORC_T_DEC64, 0, td, NULL, 0, NULL, 0, 15, 2, {0, 0}};
    orc_col c_tax = {ORC_T_DEC64, 0, tt, NULL, 0, NULL, 0, 15, 2, {0, 0}};
    orc_col c_price = {ORC_T_DEC64, 0, tp, NULL, 0, NULL, 0, 15, 2, {0, 0}};
    int e = 0;
    e |= orc_decimal_arith(ORC_OP_MINUS, &c_one, &c_disc, k, ORC_T_DEC64, 16, 2, omd, NULL, NULL);
    orc_col c_omd = {ORC_T_DEC64, 0, omd, NULL, 0, NULL, 0, 16, 2, {0, 0}};
    e |= orc_decimal_arith(ORC_OP_MULTIPLY, &c_price, &c_omd, k, ORC_T_DEC128, 31, 4, dp, NULL, NULL);
    e |= orc_decimal_arith(ORC_OP_PLUS, &c_one, &c_tax, k, ORC_T_DEC64, 16, 2, opt, NULL, NULL);
    orc_col c_opt = {ORC_T_DEC64, 0, opt, NULL, 0, NULL, 0, 16, 2, {0, 0}};
    orc_col c_dp = {ORC_T_DEC128, 0, dp, NULL, 0, NULL, 0, 31, 4, {0, 0}};
    e |= orc_decimal_arith(ORC_OP_MULTIPLY, &c_dp, &c_opt, k, ORC_T_DEC128, 38, 6, ch, NULL, NULL);
    if (e) w->rc = e;
    /* TransformPartialAggregate */
    orc_col keys[2] = {{ORC_T_STRING, 0, trf, NULL, 0, NULL, 0, 0, 0, {0, 0}}, {ORC_T_STRING, 0, tls, NULL, 0, NULL, 0, 0, 0, {0, 0}}};
    orc_col args[6]; memset(args, 0, sizeof args);
    orc_col c_qty = {ORC_T_DEC64, 0, tq, NULL, 0, NULL, 0, 15, 2, {0, 0}};
    orc_col c_ch = {ORC_T_DEC128, 0, ch, NULL, 0, NULL, 0, 38, 6, {0, 0}};
    args[0] = c_qty; args[1] = c_price; args[2] = c_dp; args[3] = c_ch; args[4] = c_disc;
    int e2 = orc_hashagg_add_block(w->ht, keys, args, k);
    if (e2) w->rc = e2;
  }
  free(bm); free(sel); free(tq); free(tp); free(td); free(tt); free(trf); free(tls); free(omd); free(opt); free(dp); free(ch);
  return NULL;
}

int orc_q1_run(const int64_t* qty, const int64_t* price, const int64_t* disc, const int64_t* tax,
               const void* rf_views, const void* ls_views, const int32_t* shipdate, int32_t cutoff,
               int64_t n, int threads, int64_t block_rows, orc_q1_result* out) {
  if (threads < 1) threads = 1;
  q1_worker* ws = (q1_worker*)calloc((size_t)threads, sizeof(q1_worker));
  pthread_t* th = (pthread_t*)calloc((size_t)threads, sizeof(pthread_t));
  for (int t = 0; t < threads; ++t) {
    q1_worker w = {qty, price, disc, tax, (const uint8_t*)rf_views, (const uint8_t*)ls_views, shipdate, cutoff, n, block_rows, t, threads, q1_table(), 0};
    ws[t] = w;
    if (threads == 1) q1_work(&ws[t]); else pthread_create(&th[t], NULL, q1_work, &ws[t]);
  }
  if (threads > 1) for (int t = 0; t < threads; ++t) pthread_join(th[t], NULL);
  /* TransformFinalAggregate: combine all partial tables */
  orc_hashagg* fin = ws[0].ht;
  int rc = ws[0].rc;
  for (int t = 1; t < threads; ++t) { int e = orc_hashagg_combine(fin, ws[t].ht); if (e) rc = e; if (ws[t].rc) rc = ws[t].rc; orc_hashagg_destroy(ws[t].ht); }
  int64_t g = orc_hashagg_num_groups(fin);
  if (g > 64) { orc_hashagg_destroy(fin); free(ws); free(th); return -1; }
  memset(out, 0, sizeof(*out));
  void* keys[2] = {out->returnflag, out->linestatus};
  void* aggs[6] = {out->sum_qty, out->sum_price, out->sum_disc_price, out->sum_charge, out->sum_disc, out->count};
  orc_hashagg_result(fin, keys, NULL, aggs, NULL);
  orc_hashagg_destroy(fin);
  free(ws); free(th);
  return rc ? -rc - 100 : (int)g;
}

/* ------------------------------------------------------------------------ */
/* sort: kernels/sort_compare.rs:33-283 (key sequence; ties unordered there,   */
/* here stable by row id)                                                     */
/* ------------------------------------------------------------------------ */
typedef struct { const orc_col* keys; const uint8_t* desc; const uint8_t* nf; int nkeys; } sort_ctx;
static __thread sort_ctx g_sort;
static int sort_cmp(const void* pa, const void* pb) {
  uint32_t a = *(const uint32_t*)pa, b = *(const uint32_t*)pb;
  for (int k = 0; k < g_sort.nkeys; ++k) {
    const orc_col* c = &g_sort.keys[k];
    int va = col_valid(c, a), vb = col_valid(c, b);
    if (!va || !vb) {
      if (va == vb) continue;
      int a_first = (!va) ? g_sort.nf[k] : !g_sort.nf[k]; /* NULL ordering is independent of asc/desc */
      return a_first ? -1 : 1;
    }
    orc_col ca = *c, cb = *c;
    ca.is_scalar = 1; cb.is_scalar = 1;
    int es = c->type == ORC_T_BOOL ? 0 : t_size(c->type);
    int r;
    if (c->type == ORC_T_BOOL) r = bit_get((const uint8_t*)c->data, a) - bit_get((const uint8_t*)c->data, b);
    else { ca.data = (const uint8_t*)c->data + (size_t)a * es; cb.data = (const uint8_t*)c->data + (size_t)b * es; ca.validity = cb.validity = NULL; r = cmp3_col(&ca, &cb, 0); }
    if (r) return g_sort.desc[k] ? -r : r;
  }
  return (a > b) - (a < b);
}
int orc_sort_perm(const orc_col* keys, const uint8_t* desc, const uint8_t* nulls_first, int nkeys, int64_t n,
                  int64_t limit, uint32_t* out_perm) {
  uint32_t* p = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)(n ? n : 1));
  for (int64_t i = 0; i < n; ++i) p[i] = (uint32_t)i;
  g_sort.keys = keys; g_sort.desc = desc; g_sort.nf = nulls_first; g_sort.nkeys = nkeys;
  qsort(p, (size_t)n, sizeof(uint32_t), sort_cmp);
  int64_t m = (limit > 0 && limit < n) ? limit : n;
  memcpy(out_perm, p, sizeof(uint32_t) * (size_t)m);
  free(p);
  return 0;
}

/* Range partition by Bounds: sort_spill.rs:1008-1040 (block_split_off_position / partition_point = the first row GREATER than the
 * bound, so rows <= bound[i] belong to partition i) applied to every row on its own: partition = number of bounds that sort
 * strictly before the row. `bounds` are ordered by the same keys (core/bounds.rs). */
static int sort_cmp_cross(const orc_col* ka, uint32_t a, const orc_col* kb, uint32_t b, const uint8_t* desc, const uint8_t* nf, int nkeys) {
  for (int k = 0; k < nkeys; ++k) {
    const orc_col *c = &ka[k], *d = &kb[k];
    int va = col_valid(c, a), vb = col_valid(d, b);
    if (!va || !vb) {
      if (va == vb) continue;
      int a_first = (!va) ? (nf ? nf[k] : 0) : !(nf ? nf[k] : 0);
      return a_first ? -1 : 1;
    }
    int r;
    if (c->type == ORC_T_BOOL) r = bit_get((const uint8_t*)c->data, a) - bit_get((const uint8_t*)d->data, b);
    else {
      orc_col ca = *c, cb = *d;
      int es = t_size(c->type);
      ca.is_scalar = cb.is_scalar = 1; ca.validity = cb.validity = NULL;
      ca.data = (const uint8_t*)c->data + (size_t)a * es; cb.data = (const uint8_t*)d->data + (size_t)b * es;
      r = cmp3_col(&ca, &cb, 0);
    }
    if (r) return (desc && desc[k]) ? -r : r;
  }
  return 0;
}
int orc_sort_bound_partition(const orc_col* keys, const orc_col* bounds, const uint8_t* desc, const uint8_t* nulls_first, int nkeys,
                             int64_t n, int64_t nbounds, uint32_t* out_part, uint64_t* out_counts) {
  for (int64_t j = 0; j <= nbounds; ++j) out_counts[j] = 0;
  for (int64_t i = 0; i < n; ++i) {
    int64_t p = 0;   /* linear on purpose: independent of the device's binary search */
    while (p < nbounds && sort_cmp_cross(bounds, (uint32_t)p, keys, (uint32_t)i, desc, nulls_first, nkeys) < 0) ++p;
    out_part[i] = (uint32_t)p;
    out_counts[p]++;
  }
  return 0;
}

/* ------------------------------------------------------------------------ */
/* inner hash join on u64 keys: hashjoin_hashtable.rs:95-137 (chained buckets,  */
/* prepend insert), fixed_keys.rs:209-269 (chain walk, key ==). Output sorted   */
/* by (probe_idx, build_row).                                                   */
/* ------------------------------------------------------------------------ */
int64_t orc_join_inner_u64(const uint64_t* build, const uint8_t* bv, int64_t nb, const uint64_t* probe,
                           const uint8_t* pv, int64_t np, uint32_t* out_p, uint32_t* out_b, int64_t max_pairs) {
  size_t cap = 1024; while (cap < (size_t)nb * 2) cap <<= 1; /* :95-108 */
  int64_t* head = (int64_t*)malloc(sizeof(int64_t) * cap);
  int64_t* next = (int64_t*)malloc(sizeof(int64_t) * (size_t)(nb ? nb : 1));
  for (size_t i = 0; i < cap; ++i) head[i] = -1;
  int shift = 64 - __builtin_ctzll(cap);
  for (int64_t r = 0; r < nb; ++r) {
    if (bv && !bit_get(bv, r)) continue; /* NULL keys never match */
    uint64_t hsh = orc_agg_hash_u64(build[r]) ; size_t idx = (size_t)(hsh >> shift);
    next[r] = head[idx]; head[idx] = r;
  }
  int64_t k = 0;
  for (int64_t i = 0; i < np; ++i) {
    if (pv && !bit_get(pv, i)) continue;
    size_t idx = (size_t)(orc_agg_hash_u64(probe[i]) >> shift);
    int64_t first = k;
    for (int64_t r = head[idx]; r >= 0; r = next[r])
      if (build[r] == probe[i]) { if (k < max_pairs) { out_p[k] = (uint32_t)i; out_b[k] = (uint32_t)r; } k++; }
    /* chain order is newest-first; sort this probe row's matches ascending */
    int64_t hi = k < max_pairs ? k : max_pairs;
    for (int64_t x = first + 1; x < hi; ++x) { uint32_t v = out_b[x]; int64_t y = x - 1; while (y >= first && out_b[y] > v) { out_b[y + 1] = out_b[y]; y--; } out_b[y + 1] = v; }
  }
  free(head); free(next);
  return k;
}

/* ------------------------------------------------------------------------ */
/* vector distances: src/common/vector/src/distance.rs:19-95. (&a*&b).sum() is  */
/* ndarray 0.15.6 (Cargo.lock:12641) numeric_util::unrolled_fold: 8 partial     */
/* sums, combined (p0+p4)+(p1+p5)+(p2+p6)+(p3+p7), then the tail sequentially.  */
/* ------------------------------------------------------------------------ */
static float nd_sum_prod(const float* a, const float* b, int n) {
  float p[8] = {0, 0, 0, 0, 0, 0, 0, 0}, acc = 0.0f;
  int i = 0;
  for (; i + 8 <= n; i += 8) for (int j = 0; j < 8; ++j) p[j] = p[j] + a[i + j] * b[i + j];
  acc = acc + (p[0] + p[4]); acc = acc + (p[1] + p[5]); acc = acc + (p[2] + p[6]); acc = acc + (p[3] + p[7]);
  for (; i < n; ++i) acc = acc + a[i] * b[i];
  return acc;
}
void orc_vec_distance(int metric, const float* base, int64_t n, int dim, const float* queries, int nq, float* out) {
  for (int q = 0; q < nq; ++q) {
    const float* b = queries + (size_t)q * dim;
    for (int64_t i = 0; i < n; ++i) {
      const float* a = base + (size_t)i * dim;
      float r;
      switch (metric) {
        case 0: { float aa = nd_sum_prod(a, a, dim), bb = nd_sum_prod(b, b, dim); r = 1.0f - nd_sum_prod(a, b, dim) / (sqrtf(aa) * sqrtf(bb)); } break;
        case 1: { float s = 0.0f; for (int k = 0; k < dim; ++k) { float d = a[k] - b[k]; s += d * d; } r = sqrtf(s); } break;
        case 2: r = nd_sum_prod(a, b, dim); break;
        default: { float s = 0.0f; for (int k = 0; k < dim; ++k) s += fabsf(a[k] - b[k]); r = s; } break;
      }
      out[(size_t)q * n + i] = r;
    }
  }
}
/* Row by row: out[i] = distance(lhs[i], rhs[i]) — the column-vs-column form of the scalar functions (scalars/vector.rs:59-260 over
 * Array(Float32) / Array(Float64), :490-560 calculate_distance over Vector(Float32 | Int8): Int8 elements are widened to f32 first),
 * a side with `*_scalar` set is one vector for every row. elem: 0 f32, 1 f64 (the *_64 functions of distance.rs:97-165, f64 result),
 * 2 i8. metric 0 cosine, 1 l2, 2 inner product, 3 l1, 4 vector_norm(lhs) (distance.rs:167-170). */
static double nd_sum_prod64(const double* a, const double* b, int n) {
  double p[8] = {0, 0, 0, 0, 0, 0, 0, 0}, acc = 0.0;
  int i = 0;
  for (; i + 8 <= n; i += 8) for (int j = 0; j < 8; ++j) p[j] = p[j] + a[i + j] * b[i + j];
  acc = acc + (p[0] + p[4]); acc = acc + (p[1] + p[5]); acc = acc + (p[2] + p[6]); acc = acc + (p[3] + p[7]);
  for (; i < n; ++i) acc = acc + a[i] * b[i];
  return acc;
}
void orc_vec_distance_rows(int metric, int elem, const void* lhs, int lhs_scalar, const void* rhs, int rhs_scalar, int64_t n, int dim, void* out) {
  float* fa = (float*)malloc(sizeof(float) * (size_t)(dim ? dim : 1));
  float* fb = (float*)malloc(sizeof(float) * (size_t)(dim ? dim : 1));
  for (int64_t i = 0; i < n; ++i) {
    const size_t ia = lhs_scalar ? 0 : (size_t)i * dim, ib = rhs_scalar ? 0 : (size_t)i * dim;
    if (elem == 1) {
      const double* a = (const double*)lhs + ia; const double* b = rhs ? (const double*)rhs + ib : a;
      double r;
      switch (metric) {
        case 0: { double aa = nd_sum_prod64(a, a, dim), bb = nd_sum_prod64(b, b, dim); r = 1.0 - nd_sum_prod64(a, b, dim) / (sqrt(aa) * sqrt(bb)); } break;
        case 1: { double s = 0.0; for (int k = 0; k < dim; ++k) { double d = a[k] - b[k]; s += d * d; } r = sqrt(s); } break;
        case 2: r = nd_sum_prod64(a, b, dim); break;
        case 3: { double s = 0.0; for (int k = 0; k < dim; ++k) s += fabs(a[k] - b[k]); r = s; } break;
        default: r = sqrt(nd_sum_prod64(a, a, dim)); break;
      }
      ((double*)out)[i] = r;
      continue;
    }
    const float *a, *b;
    if (elem == 2) {
      for (int k = 0; k < dim; ++k) { fa[k] = (float)((const int8_t*)lhs)[ia + k]; fb[k] = rhs ? (float)((const int8_t*)rhs)[ib + k] : fa[k]; }
      a = fa; b = fb;
    } else { a = (const float*)lhs + ia; b = rhs ? (const float*)rhs + ib : a; }
    float r;
    switch (metric) {
      case 0: { float aa = nd_sum_prod(a, a, dim), bb = nd_sum_prod(b, b, dim); r = 1.0f - nd_sum_prod(a, b, dim) / (sqrtf(aa) * sqrtf(bb)); } break;
      case 1: { float s = 0.0f; for (int k = 0; k < dim; ++k) { float d = a[k] - b[k]; s += d * d; } r = sqrtf(s); } break;
      case 2: r = nd_sum_prod(a, b, dim); break;
      case 3: { float s = 0.0f; for (int k = 0; k < dim; ++k) s += fabsf(a[k] - b[k]); r = s; } break;
      default: r = sqrtf(nd_sum_prod(a, a, dim)); break;
    }
    ((float*)out)[i] = r;
  }
  free(fa); free(fb);
}
/* scalar statement of cpp/avx2.c:45-139 impl_score_dot_avx / impl_score_l1_avx (exact integer sums) */
void orc_score_u8(int is_l1, const uint8_t* q, const uint8_t* base, int64_t n, int dim, float* out) {
  for (int64_t i = 0; i < n; ++i) {
    const uint8_t* v = base + (size_t)i * dim; int64_t s = 0;
    for (int k = 0; k < dim; ++k) s += is_l1 ? (q[k] > v[k] ? q[k] - v[k] : v[k] - q[k]) : (int64_t)q[k] * v[k];
    out[i] = (float)s;
  }
}

/* ------------------------------------------------------------------------ */
/* TPC-H Q3 (benchmark/tpch/queries/03.sql:1-18), reference-shaped:            */
/*   customer: TransformFilter(c_mktsegment = seg) -> build side of join #1      */
/*   orders:   TransformFilter(o_orderdate < date) -> probe join #1 -> build #2  */
/*   lineitem: per block_rows block: TransformFilter(l_shipdate > date) ->       */
/*             probe join #2 (HashJoinHashTable chains, hashjoin_hashtable.rs:   */
/*             95-137, fixed_keys.rs:209-269) -> take both sides (inner_join.rs: */
/*             248-268) -> decimal maps 1 - l_discount, price * (..)             */
/*             (decimal/arithmetic.rs:190-316) -> TransformPartialAggregate on   */
/*             (l_orderkey, o_orderdate, o_shippriority) with sum(Decimal128)    */
/*   final merge (transform_aggregate_final.rs:160-175) -> sort by revenue desc, */
/*   o_orderdate asc with LIMIT (kernels/sort_compare.rs:197-209).               */
/* ------------------------------------------------------------------------ */
typedef struct { int64_t* head; int64_t* next; const uint64_t* keys; int64_t nb; int shift; } jtab;
static void jtab_build(jtab* t, const uint64_t* keys, int64_t nb) {
  size_t cap = 1024; while (cap < (size_t)nb * 2) cap <<= 1;
  t->head = (int64_t*)malloc(sizeof(int64_t) * cap);
  t->next = (int64_t*)malloc(sizeof(int64_t) * (size_t)(nb ? nb : 1));
  for (size_t i = 0; i < cap; ++i) t->head[i] = -1;
  t->shift = 64 - __builtin_ctzll(cap); t->keys = keys; t->nb = nb;
  for (int64_t r = 0; r < nb; ++r) { size_t idx = (size_t)(orc_agg_hash_u64(keys[r]) >> t->shift); t->next[r] = t->head[idx]; t->head[idx] = r; }
}
/* pairs of one probe block, (probe_idx asc, build_row asc) */
static int64_t jtab_probe(const jtab* t, const uint64_t* probe, int64_t np, uint32_t* out_p, uint32_t* out_b, int64_t cap) {
  int64_t k = 0;
  for (int64_t i = 0; i < np; ++i) {
    size_t idx = (size_t)(orc_agg_hash_u64(probe[i]) >> t->shift);
    int64_t first = k;
    for (int64_t r = t->head[idx]; r >= 0; r = t->next[r])
      if (t->keys[r] == probe[i]) { if (k < cap) { out_p[k] = (uint32_t)i; out_b[k] = (uint32_t)r; } k++; }
    int64_t hi = k < cap ? k : cap;
    for (int64_t x = first + 1; x < hi; ++x) { uint32_t v = out_b[x]; int64_t y = x - 1; while (y >= first && out_b[y] > v) { out_b[y + 1] = out_b[y]; y--; } out_b[y + 1] = v; }
  }
  return k;
}
static void jtab_free(jtab* t) { free(t->head); free(t->next); }

typedef struct {
  const int64_t *lok, *lprice, *ldisc; const int32_t* lship; int64_t n, block_rows; int32_t date;
  const jtab* jt; const int64_t* b_orderkey; const int32_t *b_orderdate, *b_shipprio;
  int tid, nthreads; orc_hashagg* ht; int rc;
} q3_worker;

static orc_hashagg* q3_table(void) {
  int32_t kt[3] = {ORC_T_I64, ORC_T_DATE, ORC_T_I32};
  orc_agg_desc ag[1]; memset(ag, 0, sizeof ag);
  ag[0].kind = ORC_AGG_SUM; ag[0].arg_type = ORC_T_DEC128; ag[0].arg_precision = 31; ag[0].arg_scale = 4;
  return orc_hashagg_create(kt, NULL, 3, ag, 1);
}

static void* q3_work(void* p) {
  q3_worker* w = (q3_worker*)p;
  int64_t B = w->block_rows;
  uint8_t* bm = (uint8_t*)malloc((size_t)(B + 63) / 8 + 8);
  uint32_t *sel = malloc(4 * (size_t)B), *pp = malloc(4 * (size_t)B * 2), *pb = malloc(4 * (size_t)B * 2);
  int64_t *tok = malloc(8 * (size_t)B), *tp = malloc(8 * (size_t)B), *td = malloc(8 * (size_t)B);
  int64_t *jok = malloc(8 * (size_t)B * 2), *jp = malloc(8 * (size_t)B * 2), *jd = malloc(8 * (size_t)B * 2), *omd = malloc(8 * (size_t)B * 2);
  int32_t *jod = malloc(4 * (size_t)B * 2), *jsp = malloc(4 * (size_t)B * 2);
  i128* rev = malloc(16 * (size_t)B * 2);
  int64_t nblocks = (w->n + B - 1) / B;
  for (int64_t b = w->tid; b < nblocks; b += w->nthreads) {
    int64_t s = b * B, m = w->n - s < B ? w->n - s : B;
    orc_col sd = {ORC_T_DATE, 0, w->lship + s, NULL, 0, NULL, 0, 0, 0, {0, 0}};
    orc_col cut = {ORC_T_DATE, 1, &w->date, NULL, 0, NULL, 0, 0, 0, {0, 0}};
    orc_cmp(ORC_CMP_GT, &sd, &cut, m, bm);
    int64_t k = orc_filter_select(bm, 0, m, sel);
    orc_take(w->lok + s, 8, sel, k, tok); orc_take(w->lprice + s, 8, sel, k, tp); orc_take(w->ldisc + s, 8, sel, k, td);
    int64_t np = jtab_probe(w->jt, (const uint64_t*)tok, k, pp, pb, 2 * B);
    if (np > 2 * B) { w->rc = 6; break; } /* o_orderkey is unique: cannot happen */
    orc_take(tok, 8, pp, np, jok); orc_take(tp, 8, pp, np, jp); orc_take(td, 8, pp, np, jd);
    orc_take(w->b_orderdate, 4, pb, np, jod); orc_take(w->b_shipprio, 4, pb, np, jsp);
    uint8_t one = 1;
    orc_col c_one = {ORC_T_U8, 1, &one, NULL, 0, NULL, 0, 0, 0, {0, 0}};
    orc_col c_disc = {ORC_T_DEC64, 0, jd, NULL, 0, NULL, 0, 15, 2, {0, 0}};
    orc_col c_price = {ORC_T_DEC64, 0, jp, NULL, 0, NULL, 0, 15, 2, {0, 0}};
    int e = orc_decimal_arith(ORC_OP_MINUS, &c_one, &c_disc, np, ORC_T_DEC64, 16, 2, omd, NULL, NULL);
    orc_col c_omd = {ORC_T_DEC64, 0, omd, NULL, 0, NULL, 0, 16, 2, {0, 0}};
    e |= orc_decimal_arith(ORC_OP_MULTIPLY, &c_price, &c_omd, np, ORC_T_DEC128, 31, 4, rev, NULL, NULL);
    if (e) w->rc = e;
    orc_col keys[3] = {{ORC_T_I64, 0, jok, NULL, 0, NULL, 0, 0, 0, {0, 0}}, {ORC_T_DATE, 0, jod, NULL, 0, NULL, 0, 0, 0, {0, 0}},
                       {ORC_T_I32, 0, jsp, NULL, 0, NULL, 0, 0, 0, {0, 0}}};
    orc_col args[1] = {{ORC_T_DEC128, 0, rev, NULL, 0, NULL, 0, 31, 4, {0, 0}}};
    int e2 = orc_hashagg_add_block(w->ht, keys, args, np);
    if (e2) w->rc = e2;
  }
  free(bm); free(sel); free(pp); free(pb); free(tok); free(tp); free(td); free(jok); free(jp); free(jd); free(omd); free(jod); free(jsp); free(rev);
  return NULL;
}

int64_t orc_q3_run(const int64_t* c_custkey, const void* c_mktsegment_views, int64_t n_cust,
                   const int64_t* o_orderkey, const int64_t* o_custkey, const int32_t* o_orderdate,
                   const int32_t* o_shippriority, int64_t n_ord,
                   const int64_t* l_orderkey, const int64_t* l_extendedprice, const int64_t* l_discount,
                   const int32_t* l_shipdate, int64_t n_li,
                   const char* segment, int32_t date, int64_t limit, int threads, int64_t block_rows,
                   int64_t* out_orderkey, void* out_revenue_i128, int32_t* out_orderdate, int32_t* out_shippriority,
                   int64_t* out_ngroups, int64_t* out_stage_rows /* [4]: customers kept, orders kept, orders joined, lineitem pairs(groups' input) or NULL */) {
  if (threads < 1) threads = 1;
  /* customer side */
  uint8_t segview[16]; memset(segview, 0, 16);
  uint32_t sl = (uint32_t)strlen(segment); if (sl > 12) return -1;
  memcpy(segview, &sl, 4); memcpy(segview + 4, segment, sl);
  uint8_t* bm = (uint8_t*)calloc((size_t)((n_cust > n_ord ? n_cust : n_ord) + 63) / 8 + 8, 1);
  uint32_t* sel = (uint32_t*)malloc(4 * (size_t)((n_cust > n_ord ? n_cust : n_ord) + 1));
  orc_col cseg = {ORC_T_STRING, 0, c_mktsegment_views, NULL, 0, NULL, 0, 0, 0, {0, 0}};
  orc_col sseg = {ORC_T_STRING, 1, segview, NULL, 0, NULL, 0, 0, 0, {0, 0}};
  orc_cmp(ORC_CMP_EQ, &cseg, &sseg, n_cust, bm);
  int64_t kc = orc_filter_select(bm, 0, n_cust, sel);
  int64_t* ck = (int64_t*)malloc(8 * (size_t)(kc + 1));
  orc_take(c_custkey, 8, sel, kc, ck);
  jtab j1; jtab_build(&j1, (const uint64_t*)ck, kc);
  /* orders side */
  orc_col od = {ORC_T_DATE, 0, o_orderdate, NULL, 0, NULL, 0, 0, 0, {0, 0}};
  orc_col cut = {ORC_T_DATE, 1, &date, NULL, 0, NULL, 0, 0, 0, {0, 0}};
  orc_cmp(ORC_CMP_LT, &od, &cut, n_ord, bm);
  int64_t ko = orc_filter_select(bm, 0, n_ord, sel);
  int64_t *fok = malloc(8 * (size_t)(ko + 1)), *fck = malloc(8 * (size_t)(ko + 1));
  int32_t *fod = malloc(4 * (size_t)(ko + 1)), *fsp = malloc(4 * (size_t)(ko + 1));
  orc_take(o_orderkey, 8, sel, ko, fok); orc_take(o_custkey, 8, sel, ko, fck);
  orc_take(o_orderdate, 4, sel, ko, fod); orc_take(o_shippriority, 4, sel, ko, fsp);
  uint32_t *pp = malloc(4 * (size_t)(ko + 1)), *pb = malloc(4 * (size_t)(ko + 1));
  int64_t kj = jtab_probe(&j1, (const uint64_t*)fck, ko, pp, pb, ko); /* c_custkey unique -> <= ko pairs */
  if (kj > ko) kj = ko;
  int64_t* bok = malloc(8 * (size_t)(kj + 1)); int32_t *bod = malloc(4 * (size_t)(kj + 1)), *bsp = malloc(4 * (size_t)(kj + 1));
  orc_take(fok, 8, pp, kj, bok); orc_take(fod, 4, pp, kj, bod); orc_take(fsp, 4, pp, kj, bsp);
  jtab j2; jtab_build(&j2, (const uint64_t*)bok, kj);
  /* lineitem side, `threads` workers with partial tables */
  q3_worker* ws = (q3_worker*)calloc((size_t)threads, sizeof(q3_worker));
  pthread_t* th = (pthread_t*)calloc((size_t)threads, sizeof(pthread_t));
  for (int t = 0; t < threads; ++t) {
    q3_worker w = {l_orderkey, l_extendedprice, l_discount, l_shipdate, n_li, block_rows, date, &j2, bok, bod, bsp, t, threads, q3_table(), 0};
    ws[t] = w;
    if (threads == 1) q3_work(&ws[t]); else pthread_create(&th[t], NULL, q3_work, &ws[t]);
  }
  if (threads > 1) for (int t = 0; t < threads; ++t) pthread_join(th[t], NULL);
  orc_hashagg* fin = ws[0].ht; int rc = ws[0].rc;
  for (int t = 1; t < threads; ++t) { int e = orc_hashagg_combine(fin, ws[t].ht); if (e) rc = e; if (ws[t].rc) rc = ws[t].rc; orc_hashagg_destroy(ws[t].ht); }
  int64_t g = orc_hashagg_num_groups(fin);
  if (out_ngroups) *out_ngroups = g;
  if (out_stage_rows) { out_stage_rows[0] = kc; out_stage_rows[1] = ko; out_stage_rows[2] = kj; out_stage_rows[3] = g; }
  int64_t m = 0;
  if (g > 0 && !rc) {
    int64_t* rk = malloc(8 * (size_t)g); int32_t *rd = malloc(4 * (size_t)g), *rs = malloc(4 * (size_t)g); i128* rr = malloc(16 * (size_t)g);
    void* keys[3] = {rk, rd, rs}; void* aggs[1] = {rr};
    orc_hashagg_result(fin, keys, NULL, aggs, NULL);
    orc_col skeys[2] = {{ORC_T_DEC128, 0, rr, NULL, 0, NULL, 0, 38, 4, {0, 0}}, {ORC_T_DATE, 0, rd, NULL, 0, NULL, 0, 0, 0, {0, 0}}};
    uint8_t desc[2] = {1, 0}, nf[2] = {0, 0};
    m = (limit > 0 && limit < g) ? limit : g;
    uint32_t* perm = malloc(4 * (size_t)g);
    orc_sort_perm(skeys, desc, nf, 2, g, limit, perm);
    for (int64_t i = 0; i < m; ++i) {
      out_orderkey[i] = rk[perm[i]]; memcpy((uint8_t*)out_revenue_i128 + 16 * i, &rr[perm[i]], 16);
      out_orderdate[i] = rd[perm[i]]; out_shippriority[i] = rs[perm[i]];
    }
    free(rk); free(rd); free(rs); free(rr); free(perm);
  }
  orc_hashagg_destroy(fin);
  jtab_free(&j1); jtab_free(&j2);
  free(bm); free(sel); free(ck); free(fok); free(fck); free(fod); free(fsp); free(pp); free(pb); free(bok); free(bod); free(bsp); free(ws); free(th);
  return rc ? -(int64_t)rc - 100 : m;
}

/* ---------------------------------------------------------------------------------------------
 * packed fixed-width keys — HashMethodFixedKeys (method_fixed_keys.rs:58-78), KeysVec (:310-403),
 * fixed_hash (:405-512), numeric_byte_size (src/query/expression/src/types.rs:606-633),
 * choose_hash_method_with_types (kernels/group_by.rs:40-80).
 * ------------------------------------------------------------------------------------------- */
static int orc_numeric_byte_size(const orc_col* c) {
  switch (c->type) {
    case ORC_T_I8: case ORC_T_U8: return 1;
    case ORC_T_I16: case ORC_T_U16: return 2;
    case ORC_T_I32: case ORC_T_U32: case ORC_T_F32: case ORC_T_DATE: return 4;
    case ORC_T_I64: case ORC_T_U64: case ORC_T_F64: case ORC_T_TIMESTAMP: return 8;
    case ORC_T_DEC64: case ORC_T_DEC128: return c->precision <= 18 ? 8 : 16;  /* can_carried_by_64 / _128 */
    default: return 0;  /* not number / date / decimal -> Serializer */
  }
}

int orc_keys_method(const orc_col* cols, int ncols) {
  int len = 0;
  for (int i = 0; i < ncols; ++i) {
    int b = orc_numeric_byte_size(&cols[i]);
    if (b == 0) return 0;
    len += b;
    if (cols[i].validity) len += 1; /* extra one byte for null flag (group_by.rs:62-65) */
  }
  if (len == 1) return 1;
  if (len == 2) return 2;
  if (len <= 4) return 4;
  if (len <= 8) return 8;
  if (len <= 16) return 16;
  if (len <= 32) return 32;
  return 0;
}

int orc_pack_keys(const orc_col* cols, int ncols, int64_t n, int key_bytes, uint8_t* out) {
  int order[64], size[64], col_off[65], null_off[64];
  if (ncols > 64) return -1;
  for (int i = 0; i < ncols; ++i) {
    order[i] = i;
    size[i] = orc_numeric_byte_size(&cols[i]);
    if (size[i] == 0) return -1;
  }
  /* group_columns.sort_by_key(numeric_byte_size) — stable (method_fixed_keys.rs:63-68) */
  for (int i = 1; i < ncols; ++i) {
    int o = order[i], j = i;
    while (j > 0 && size[order[j - 1]] > size[o]) { order[j] = order[j - 1]; --j; }
    order[j] = o;
  }
  col_off[0] = 0;
  for (int s = 0; s < ncols; ++s) col_off[s + 1] = col_off[s] + size[order[s]];   /* KeysVec::new :326-337 */
  int null_offset = col_off[ncols];
  for (int s = 0; s < ncols; ++s) null_off[s] = cols[order[s]].validity ? null_offset++ : -1;  /* :342-353 */
  if (null_offset > key_bytes) return -1;  /* "size of T too small" */
  memset(out, 0, (size_t)n * key_bytes);
  for (int s = 0; s < ncols; ++s) {
    const orc_col* c = &cols[order[s]];
    const int kb = size[order[s]];
    for (int64_t row = 0; row < n; ++row) {
      int64_t j = c->is_scalar ? 0 : row;
      if (c->validity && !((c->validity[(c->validity_offset + j) >> 3] >> ((c->validity_offset + j) & 7)) & 1)) {
        out[row * key_bytes + null_off[s]] = 1;  /* set_null :389-397 */
        continue;
      }
      uint8_t* dst = out + row * key_bytes + col_off[s];
      if (c->type == ORC_T_DEC128) {
        /* DecimalView<FROM, TO>: marshal the value in the carrier of its precision (:486-499) */
        memcpy(dst, (const uint8_t*)c->data + 16 * j, kb);  /* little endian: low bytes of the i128 */
      } else if (c->type == ORC_T_DEC64 && kb == 16) {
        __int128 v = (__int128)((const int64_t*)c->data)[j];
        memcpy(dst, &v, 16);
      } else {
        memcpy(dst, (const uint8_t*)c->data + (size_t)kb * j, kb);  /* value.marshal(slice): little endian */
      }
    }
  }
  return 0;
}

/* ------------------------------------------------------------------------ */
/* HashMethodSerializer: serialize_group_columns / serialize_column_binary     */
/* (src/query/expression/src/kernels/group_by_hash/utils.rs:33-160,           */
/*  method_serializer.rs:33-52): per row, column after column — numbers /     */
/* decimals / dates / timestamps as their little-endian bytes, Boolean as one  */
/* byte, String as u64 length + bytes, a nullable column as one byte `valid`   */
/* followed by the value only when valid. offsets[n + 1]; data may be NULL to  */
/* size. Returns the total number of bytes, -1 for an unsupported column.      */
/* ------------------------------------------------------------------------ */
static int ser_fixed_width(int type) {
  switch (type) {
    case ORC_T_BOOL: case ORC_T_I8: case ORC_T_U8: return 1;
    case ORC_T_I16: case ORC_T_U16: return 2;
    case ORC_T_I32: case ORC_T_U32: case ORC_T_F32: case ORC_T_DATE: return 4;
    case ORC_T_I64: case ORC_T_U64: case ORC_T_F64: case ORC_T_TIMESTAMP: case ORC_T_DEC64: return 8;
    case ORC_T_DEC128: return 16;
    case ORC_T_DEC256: return 32;
    default: return 0;
  }
}
int64_t orc_serialize_keys(const orc_col* cols, int ncols, int64_t n, uint64_t* offsets, uint8_t* data) {
  uint64_t pos = 0;
  for (int64_t i = 0; i < n; ++i) {
    offsets[i] = pos;
    for (int c = 0; c < ncols; ++c) {
      const orc_col* col = &cols[c];
      int64_t j = col->is_scalar ? 0 : i;
      if (col->validity) { /* Column::Nullable: push valid, then the value only when valid */
        int valid = col_valid(col, i);
        if (data) data[pos] = (uint8_t)valid;
        pos += 1;
        if (!valid) continue;
      }
      if (col->type == ORC_T_STRING) {
        uint32_t len; const uint8_t* p = view_bytes((const uint32_t*)col->data + 4 * j, col->buffers, &len);
        uint64_t l64 = len;
        if (data) { memcpy(data + pos, &l64, 8); memcpy(data + pos + 8, p, len); }
        pos += 8 + len;
      } else if (col->type == ORC_T_BOOL) {
        if (data) data[pos] = (uint8_t)bit_get((const uint8_t*)col->data, j);
        pos += 1;
      } else {
        int w = ser_fixed_width(col->type);
        if (!w) return -1;
        if (data) memcpy(data + pos, (const uint8_t*)col->data + (size_t)j * w, (size_t)w);
        pos += (uint64_t)w;
      }
    }
  }
  offsets[n] = pos;
  return (int64_t)pos;
}


/* ------------------------------------------------------------------------ */
/* siphash64 and the hash-shuffle scatter indices.                             */
/* siphash64: src/query/functions/src/scalars/hash.rs:323-328 — SipHasher13::new_with_keys(0, 0) of the `siphasher` crate      */
/* (Cargo.lock 1.0.1; NOT in /root/reference: the published SipHash-1-3 is restated) fed by DFHash (:436-545): integers / floats  */
/* as native-width little-endian bytes, bool one byte, strings their bytes; decimals scalars/decimal/src/hash.rs:144-160: the  */
/* scale byte, then the i128. Pinned on the reference's golden file (tests/golden/siphash.json) and hash.rs:563-600.            */
/* scatter indices: flight_scatter_hash.rs:133-233 (one key: siphash64 % n, NULL -> default; several: DefaultHasher over them). */
/* ------------------------------------------------------------------------ */
typedef struct { uint64_t v0, v1, v2, v3, tail; int ntail; uint64_t len; } sip13;
static uint64_t rotl64(uint64_t x, int b) { return (x << b) | (x >> (64 - b)); }
static void sip_round(sip13* s) {
  s->v0 += s->v1; s->v1 = rotl64(s->v1, 13); s->v1 ^= s->v0; s->v0 = rotl64(s->v0, 32);
  s->v2 += s->v3; s->v3 = rotl64(s->v3, 16); s->v3 ^= s->v2;
  s->v0 += s->v3; s->v3 = rotl64(s->v3, 21); s->v3 ^= s->v0;
  s->v2 += s->v1; s->v1 = rotl64(s->v1, 17); s->v1 ^= s->v2; s->v2 = rotl64(s->v2, 32);
}
static void sip_init(sip13* s) {
  s->v0 = 0x736f6d6570736575ULL; s->v1 = 0x646f72616e646f6dULL; s->v2 = 0x6c7967656e657261ULL; s->v3 = 0x7465646279746573ULL;
  s->tail = 0; s->ntail = 0; s->len = 0;
}
static void sip_write(sip13* s, const uint8_t* p, size_t n) {   /* a byte stream: Hasher::write calls concatenate */
  for (size_t i = 0; i < n; ++i) {
    s->tail |= (uint64_t)p[i] << (8 * s->ntail);
    s->len++;
    if (++s->ntail == 8) { s->v3 ^= s->tail; sip_round(s); s->v0 ^= s->tail; s->tail = 0; s->ntail = 0; }
  }
}
static uint64_t sip_finish(sip13* s) {
  uint64_t b = ((s->len & 0xff) << 56) | s->tail;
  s->v3 ^= b; sip_round(s); s->v0 ^= b;
  s->v2 ^= 0xff;
  sip_round(s); sip_round(s); sip_round(s);
  return s->v0 ^ s->v1 ^ s->v2 ^ s->v3;
}
static int sip_value(const orc_col* c, int64_t i, uint64_t* out) {
  int64_t j = c->is_scalar ? 0 : i;
  sip13 s; sip_init(&s);
  switch (c->type) {
    case ORC_T_BOOL: { uint8_t b = (uint8_t)bit_get((const uint8_t*)c->data, j); sip_write(&s, &b, 1); } break;
    case ORC_T_STRING: {
      uint32_t len; const uint8_t* p = view_bytes((const uint32_t*)c->data + 4 * j, c->buffers, &len);
      sip_write(&s, p, len);
    } break;
    case ORC_T_DEC64: case ORC_T_DEC128: case ORC_T_DEC256: {
      if (c->precision < 1 || c->precision > 38) return 1;
      uint8_t sc = (uint8_t)c->scale;
      i128 v = c->type == ORC_T_DEC64 ? (i128)((const int64_t*)c->data)[j]
             : c->type == ORC_T_DEC128 ? ((const i128*)c->data)[j] : ((const i128*)c->data)[2 * j];   /* the low half of the i256 */
      sip_write(&s, &sc, 1);
      sip_write(&s, (const uint8_t*)&v, 16);
    } break;
    default: {
      int es = t_size(c->type);
      if (es != 1 && es != 2 && es != 4 && es != 8) return 1;
      sip_write(&s, (const uint8_t*)c->data + (size_t)j * es, (size_t)es);
    }
  }
  *out = sip_finish(&s);
  return 0;
}
int orc_siphash64(const orc_col* col, int64_t n, uint64_t* out) {
  for (int64_t i = 0; i < n; ++i) {
    out[i] = 0;
    if (!col_valid(col, i)) continue;
    if (sip_value(col, i, &out[i])) return 1;
  }
  return 0;
}
int orc_scatter_indices(const orc_col* keys, int nkeys, int64_t n, uint64_t scatter_size, uint64_t default_index, uint32_t* out_index,
                        uint64_t* out_counts) {
  for (uint64_t d = 0; d < scatter_size; ++d) out_counts[d] = 0;
  for (int64_t i = 0; i < n; ++i) {
    uint64_t idx;
    if (nkeys == 1) {
      uint64_t h = 0;
      if (!col_valid(&keys[0], i)) idx = default_index;
      else { if (sip_value(&keys[0], i, &h)) return 1; idx = h % scatter_size; }
    } else {
      sip13 s; sip_init(&s);   /* DefaultHasher::default() */
      for (int k = 0; k < nkeys; ++k) {
        uint64_t h = 0;
        if (col_valid(&keys[k], i) && sip_value(&keys[k], i, &h)) return 1;
        sip_write(&s, (const uint8_t*)&h, 8);   /* write_u64 */
      }
      idx = sip_finish(&s) % scatter_size;
    }
    out_index[i] = (uint32_t)idx;
    out_counts[idx]++;
  }
  return 0;
}

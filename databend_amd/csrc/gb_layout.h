// gb_layout.h — device row layout of the hash-aggregation table (the HBM analogue
// of the reference's Payload row, src/query/expression/src/aggregate/payload.rs:47-72,
// payload_row.rs:51-215): every field is a whole number of u64 words
//
//   [key words ...][validity word (only if a key is nullable)][hash][state words ...]
//
// key encodings (canonical, so that key equality is word equality):
//   ints/date/timestamp/dec64/bool : 1 word, value widened (sign-/zero-extended)
//   f32/f64                        : 1 word, raw bits zero-extended
//   dec128                         : 2 words (lo, hi)
//   dec256                         : 4 words (little endian limbs; r04)
//   string (len <= 12)             : 2 words = the 16-byte inline view, bytes past len zeroed
//   string (len  > 12)             : word 0 = len | first 4 bytes << 32 (the view's prefix), word 1 = where the bytes live:
//                                    in a table row the OFFSET into the table's arena (a device bump allocator, the
//                                    analogue of the Payload's arena, payload.rs:361-486: `(len, ptr)`); in an input /
//                                    partial row the device ADDRESS of the bytes (column data buffer, or a peer's arena)
//   NULL key                       : value words zeroed, bit `k` of the validity word cleared
// state encodings:
//   COUNT                : 1 word u64
//   SUM int/dec64        : 1 word (wrapping i64/u64)        (aggregate_sum.rs:113-129,203-216)
//   SUM f32/f64          : 1 word f64
//   SUM dec128           : 3 words (lo, hi, ext) = exact 192-bit two's complement sum, so that
//                          "left [DECIMAL_MIN, DECIMAL_MAX]" is decided on the exact total
//                          (aggregate_sum.rs:203-216 checks the running sum; for same-signed
//                          inputs the two coincide)
//   SUM dec256           : 5 words = exact 320-bit two's complement sum (r04), row path only
//   MIN/MAX              : 2 words [order-preserving key][has value]; Decimal128 / String: 3 words (gb_device.h)
//   SUM over a NULLABLE argument carries one more word at the end: [seen a non-NULL row] (0 / 1, OR-merged) — the
//   device analogue of the flag byte AggregateNullUnaryAdaptor<true> appends to the nested state
//   (adaptors/aggregate_null_adaptor.rs:366-400,508-540): a group whose argument was NULL in every row yields NULL,
//   not 0. For MIN/MAX the [has value] word already is that flag.
// The same row is the unit of exchange between ranks (serialized partial state).
#pragma once
#ifndef __HIPCC_RTC__
#include <stdint.h>
#endif

#ifndef __HIPCC_RTC__
#include "../../include/dbhip.h"
#else
#include "dbhip.h"
#endif

#define GB_MAX_KEYS 16
#define GB_MAX_AGGS 24
#define GB_MAX_STATE_WORDS 4   // widest state: nullable Decimal128 sum = (lo, hi, ext, flag)

struct GbLayout {
  int32_t nkeys, naggs;
  int32_t key_type[GB_MAX_KEYS];
  int32_t key_off[GB_MAX_KEYS];    // word offset
  int32_t key_words[GB_MAX_KEYS];
  int32_t key_nullable[GB_MAX_KEYS];
  int32_t validity_word;           // -1 when no key is nullable
  int32_t nkey_words;              // key words + validity word (everything compared for equality)
  uint32_t str_w1_mask;            // bit j: word j is the SECOND word of a string key (inline bytes or arena offset / address)
  int32_t hash_word;
  int32_t agg_kind[GB_MAX_AGGS];
  int32_t agg_type[GB_MAX_AGGS];
  int32_t agg_off[GB_MAX_AGGS];
  int32_t agg_words[GB_MAX_AGGS];
  int32_t agg_flag[GB_MAX_AGGS];   // word (inside the state) of the "seen a non-NULL row" flag of a nullable SUM; 0 = none
  int32_t agg_nullable[GB_MAX_AGGS];
  int32_t agg_precision[GB_MAX_AGGS];
  int32_t agg_scale[GB_MAX_AGGS];
  int32_t W;                       // words per row
};

struct GbCol {  // by-value copy of dbhip_col for kernel arguments
  const void* data;
  const uint8_t* validity;
  int64_t voff;
  const void* const* buffers;
  int32_t type;
  int32_t is_scalar;
};

struct GbCols {
  GbCol key[GB_MAX_KEYS];
  GbCol arg[GB_MAX_AGGS];
  // optional predicate Bitmap (LSB-first, read from bit filter_off): rows whose bit is 0 do not take part — the
  // TransformFilter in front of the aggregate pushed down, so that no column is ever compacted (filter_executor.rs:81-118)
  const uint8_t* filter;
  int64_t filter_off;
};

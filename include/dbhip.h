/*
 * dbhip.h — C-ABI of libdbhip.so: MI355X (gfx950) kernels for the Databend
 * column-batch execution hot path.
 *
 * Boundary rules (mirrors the only Rust→C precedent in the reference,
 * src/query/storages/common/index/src/hnsw_index/quantization/encoded_vectors_u8.rs:415-428:
 * plain pointers + lengths, scalar status return, no ownership transfer, callable
 * from any thread):
 *   - every entry point returns int32 status (DBHIP_OK == 0); the message of the
 *     last failure on the calling thread is available from dbhip_last_error();
 *   - no Rust/C++ type, exception or panic crosses the ABI;
 *   - all `const void* / void*` buffer arguments are DEVICE pointers (HBM) unless
 *     the parameter name ends in `_host`; Arrow-layout host buffers
 *     (Buffer<T>::as_ptr(), Bitmap::as_slice()) are moved with
 *     dbhip_memcpy_h2d/d2h — columns are meant to stay resident in HBM between
 *     operators;
 *   - buffers are borrowed for the duration of the call; outputs are written into
 *     caller-provided device buffers (allocated with dbhip_alloc or by any other
 *     allocator of the same HIP context, e.g. a torch tensor's data_ptr());
 *   - `stream` is a hipStream_t passed as void* (NULL = the library's own
 *     per-device stream). Calls are asynchronous on that stream unless they
 *     return a host value, in which case they synchronise the stream.
 *   - threading: every entry point may be called from any host thread, and different threads may be inside the
 *     library at the same time (handles — group-by tables, joins, indexes — belong to one caller at a time, like a
 *     Processor instance). Internal scratch is keyed by (THREAD, STREAM): a thread may keep asynchronous calls in flight on
 *     several streams at once, and several threads may share a stream (their calls are stream-ordered).
 *
 * Each group of functions cites the reference interface it replaces
 * (paths relative to the Databend source tree).
 */
#ifndef DBHIP_H
#define DBHIP_H

#ifndef __HIPCC_RTC__   /* (run-time compiled kernels include this header for its enums; their prelude has the typedefs) */
#include <stddef.h>
#include <stdint.h>
#endif

#ifdef __cplusplus
extern "C" {
#endif

#define DBHIP_ABI_VERSION 6   /* 6 (round 6): pipelined fused aggregation (dbhip_groupby_set_pipelined / dbhip_groupby_checkpoint); 5 (round 5): ZSTD on the device, batched chunk decode (dbhip_pq_chunks_decode_device); 4 (round 4): block scatter / concat, exchange and plan calls, device-mode scan, cancellation, row-wise vector distance */

/* ---- status codes ------------------------------------------------------- */
enum {
  DBHIP_OK = 0,
  DBHIP_ERR_INVALID = 1,      /* bad argument / unsupported type combination     */
  DBHIP_ERR_HIP = 2,          /* a HIP runtime call failed                       */
  DBHIP_ERR_NO_DEVICE = 3,    /* no gfx950 device visible                        */
  DBHIP_ERR_ROW_ERRORS = 4,   /* per-row errors were raised (see err bitmap)     */
  DBHIP_ERR_OVERFLOW = 5,     /* aggregate decimal overflow (aggregate_sum.rs:203-216) */
  DBHIP_ERR_CAPACITY = 6,     /* fixed-capacity fast path overflowed; retry general path */
  DBHIP_ERR_UNSUPPORTED = 7,  /* caller must keep the CPU closure for this case  */
  DBHIP_ERR_CANCELLED = 8     /* dbhip_stream_cancel was called for the stream: the operator stopped between two launches */
};

/* ---- physical types (src/query/expression/src/types.rs:234 DataType,
 *      src/common/column/src/{buffer,bitmap,binview}) ------------------------ */
typedef enum {
  DBHIP_T_BOOL = 1,      /* Bitmap, LSB-first, 1 bit per row                     */
  DBHIP_T_I8 = 2, DBHIP_T_I16 = 3, DBHIP_T_I32 = 4, DBHIP_T_I64 = 5,
  DBHIP_T_U8 = 6, DBHIP_T_U16 = 7, DBHIP_T_U32 = 8, DBHIP_T_U64 = 9,
  DBHIP_T_F32 = 10, DBHIP_T_F64 = 11,
  DBHIP_T_DATE = 12,     /* i32 days (DateType)                                  */
  DBHIP_T_TIMESTAMP = 13,/* i64 micros                                           */
  DBHIP_T_DEC64 = 14,    /* DecimalColumn::Decimal64  (i64)                      */
  DBHIP_T_DEC128 = 15,   /* DecimalColumn::Decimal128 (i128, little endian)      */
  DBHIP_T_STRING = 16,   /* BinaryViewColumn: 16-byte View{len,prefix,buf,off}   */
  DBHIP_T_DEC256 = 17    /* DecimalColumn::Decimal256 (i256: 32 bytes, little endian two's complement; ABI 3) */
} dbhip_type;

/* binary operators (numeric_basic_arithmetic.rs:255-544, decimal/arithmetic.rs:44-49) */
typedef enum {
  DBHIP_OP_PLUS = 0, DBHIP_OP_MINUS = 1, DBHIP_OP_MULTIPLY = 2,
  DBHIP_OP_DIVIDE = 3,   /* '/' : always f64, "divided by zero" row error        */
  DBHIP_OP_INTDIV = 4,   /* 'div'                                                */
  DBHIP_OP_MODULO = 5,
  DBHIP_OP_DIV0 = 6,     /* div0:    f64, x / 0 = 0, never raises (numeric_basic_arithmetic.rs:441-448, 524-531)   */
  DBHIP_OP_DIVNULL = 7   /* divnull: f64, x / 0 = NULL: the row's bit of `err_bitmap` is cleared and counted in
                            *err_count_dev — here the bitmap is the result's validity contribution, not an error
                            (:450-457, 533-543)                                                                    */
} dbhip_arith_op;

/* comparison operators (src/query/functions/src/scalars/comparison.rs:98-112) */
typedef enum {
  DBHIP_CMP_EQ = 0, DBHIP_CMP_NOTEQ = 1, DBHIP_CMP_LT = 2,
  DBHIP_CMP_LTE = 3, DBHIP_CMP_GT = 4, DBHIP_CMP_GTE = 5
} dbhip_cmp_op;

/* A column argument. `data` points at n values (or at ONE value when is_scalar
 * is set — Value::Scalar, src/query/expression/src/values.rs:122). For
 * DBHIP_T_STRING `data` points at 16-byte views and `buffers` at a device array
 * of device pointers to the data buffers (binview/view.rs:30-42).
 * `validity` (may be NULL) is an LSB-first bitmap read from bit `validity_offset`
 * (bitmap/immutable.rs:78-85). */
typedef struct {
  int32_t type;                 /* dbhip_type                                     */
  int32_t is_scalar;            /* 1: `data` holds a single value                 */
  const void* data;
  const uint8_t* validity;
  int64_t validity_offset;
  const void* const* buffers;   /* STRING only                                    */
  int32_t n_buffers;
  uint8_t precision;            /* DEC64/DEC128 only (DecimalSize)                */
  uint8_t scale;
  uint8_t _pad[2];
} dbhip_col;

/* ---- runtime ------------------------------------------------------------- */
int32_t dbhip_abi_version(void);
/* Binds the calling process to `device` and creates the library stream.
 * Fails with DBHIP_ERR_NO_DEVICE when no GPU is visible: there is no CPU fallback. */
int32_t dbhip_init(int32_t device);
int32_t dbhip_device_count(int32_t* out_count_host);
const char* dbhip_last_error(void);
int32_t dbhip_alloc(size_t bytes, void** out_dev_ptr_host);
int32_t dbhip_free(void* dev_ptr);
/* dbhip_alloc/dbhip_free keep freed blocks of >= 1 MiB in size-class free lists (budget:
 * env DBHIP_CACHE_BYTES, default 96 GiB) so that operator outputs do not pay hipMalloc per
 * call; dbhip_trim returns the cached blocks to the driver. */
int32_t dbhip_trim(void);
int32_t dbhip_memcpy_h2d(void* dst_dev, const void* src_host, size_t bytes, void* stream);
int32_t dbhip_memcpy_d2h(void* dst_host, const void* src_dev, size_t bytes, void* stream);
int32_t dbhip_memcpy_d2d(void* dst_dev, const void* src_dev, size_t bytes, void* stream);
int32_t dbhip_memset(void* dst_dev, int32_t byte, size_t bytes, void* stream);
int32_t dbhip_stream_create(void** out_stream_host);
int32_t dbhip_stream_destroy(void* stream);        /* drains the stream, frees the library's scratch of it (every thread's), destroys it */
/* Internal scratch is bounded — a thread keeps the scratch of at most 8 streams (least recently used evicted), a thread that
 * exits frees its own — and can be returned at any time: drains `stream` (NULL = the library stream) and frees every thread's
 * scratch buffers of it. For streams the library did not create (a host's or torch's stream pool) call this before the stream
 * goes away. REQUIREMENT (also of dbhip_stream_destroy): no thread may be inside a dbhip call on `stream` while this runs — the
 * buffers of other threads are freed without their taking part; calls on OTHER streams may run concurrently. */
int32_t dbhip_stream_release_scratch(void* stream);
int32_t dbhip_stream_sync(void* stream);
/* Cancellation (the reference polls check_interrupt() inside its long loops, src/query/pipeline/src/core/processor.rs:36-41,
 * new_hash_join/memory/inner_join.rs:296): dbhip_stream_cancel marks `stream` (NULL = the library stream) — callable from ANY thread,
 * typically the one that kills the query — and every multi-launch operator working on that stream (group-by add_block and its chunk
 * loops, the vector-index search's query batches and ranges, the sort's passes, the partitioned exchange) returns
 * DBHIP_ERR_CANCELLED at its next poll, between two launches: at most one kernel (a few ms) later. What the operator had built so far
 * is left consistent but incomplete (a group-by table holds the chunks merged so far): the caller drops or resets the handle. The mark
 * stays until dbhip_stream_cancel_clear (a killed pipeline's stream is usually destroyed instead, which clears it too). */
int32_t dbhip_stream_cancel(void* stream);
int32_t dbhip_stream_cancel_clear(void* stream);
/* HIP-event timing on `stream` (bench.py's roofline figure): */
int32_t dbhip_event_create(void** out_event_host);
int32_t dbhip_event_record(void* event, void* stream);
int32_t dbhip_event_elapsed_ms(void* start, void* stop, float* out_ms_host);
int32_t dbhip_event_destroy(void* event);
/* Duration (HIP events on the launch stream) of the dominant kernel launched by the
 * most recent dbhip_q1_fused / dbhip_sum_a_plus_b_mul_c_i64 / dbhip_vec_topk call of
 * this thread — the figure bench.py's `roofline.achieved` is computed from. */
int32_t dbhip_last_kernel_ms(float* out_ms_host);

/* ---- a2/a3: numeric arithmetic -------------------------------------------
 * Replaces the closures registered by register_plus/minus/multiply/divide/div/modulo
 * (numeric_basic_arithmetic.rs:255-544) that vectorize_2_arg drives
 * (src/query/expression/src/function/register_vectorize.rs:110-216).
 * lhs/rhs types must be numeric; out_type must be the ResultTypeOfBinary entry
 * (utils/arithmetics_type.rs) — checked. Integer arithmetic wraps
 * (Cargo.toml:577 overflow-checks=false). For DIVIDE/INTDIV/MODULO a zero divisor
 * raises the row error "divided by zero": bit `i` of `err_bitmap` (LSB-first,
 * 1 = row ok; may be NULL) is cleared and `*err_count_dev` incremented
 * (EvalContext::set_error, src/query/expression/src/function.rs:534-556);
 * rows whose validity bit is 0 never raise. Payload is computed for all rows
 * (passthrough_nullable, register_vectorize.rs:447-471). */
int32_t dbhip_arith(int32_t op, const dbhip_col* lhs, const dbhip_col* rhs,
                    int64_t n, int32_t out_type, void* out,
                    uint8_t* err_bitmap, uint64_t* err_count_dev, void* stream);
/* to_<number> / try_to_<number> (CAST / TRY_CAST between the ten number types; register_number_to_number,
 * src/query/functions/src/scalars/arithmetic/src/arithmetic.rs:448-700): lossless pairs are `as`; float -> integer rounds
 * half away from zero first when `rounding_mode` (the numeric_cast_option setting) and truncates otherwise; a value the
 * destination cannot hold (num_traits::cast = None, NaN included) raises the row error "number overflowed" — bit i of
 * `bitmap` (preset to ones here) cleared, *err_count_dev incremented, the row holds 0; NULL input rows never raise —
 * or, for is_try, becomes NULL: `bitmap` is then the result's validity (input validity AND "representable").
 * `bitmap`: LSB-first, whole 64-bit words (ceil(n / 64) * 8 bytes, 8-byte aligned); required for is_try. */
int32_t dbhip_cast(const dbhip_col* src, int32_t dst_type, int32_t is_try, int32_t rounding_mode, int64_t n, void* out,
                   uint8_t* bitmap, uint64_t* err_count_dev, void* stream);
/* Result type table: returns dbhip_type or -1 (arithmetics_type.rs / codegen
 * src/query/codegen/src/writes/arithmetics_type.rs:222-250). */
int32_t dbhip_arith_result_type(int32_t op, int32_t lhs_type, int32_t rhs_type);

/* Fused config-1 twin: out[0] = wrapping sum over rows of a[i] + b[i]*c[i] (Int64).
 * Replaces plus∘multiply (3.1 of SURVEY) followed by NumberSumState::add_batch
 * (aggregate_sum.rs:71-129) without materialising intermediates. `out_sum_dev`
 * must be zeroed by the caller (it is accumulated into). */
int32_t dbhip_sum_a_plus_b_mul_c_i64(const int64_t* a, const int64_t* b, const int64_t* c,
                                     int64_t n, int64_t* out_sum_dev, void* stream);
/* sum(column) for any numeric type into the ResultTypeOfUnary::Sum type
 * (i64/u64 wrapping, f64 pairwise — see DESIGN.md for the float caveat). */
int32_t dbhip_sum(const dbhip_col* col, int64_t n, void* out_sum_dev, void* stream);

/* ---- a1 + §8f-2: fused expression evaluation ---------------------------------
 * Replaces Evaluator::run over a whole Expr tree (src/query/expression/src/evaluator.rs:229-464) /
 * BlockOperator::Map (src/query/sql/src/evaluator/block_operator.rs:42-85): instead of one kernel and one
 * materialised column per call node, the binding flattens the tree (post-order) into a register program
 * that ONE launch interprets; intermediates live in LDS, never in HBM. Registers 0..15; up to 24
 * instructions (LOADs not counted) and 8 input columns (numeric / Date / Timestamp / Decimal64 / Decimal128 /
 * Boolean).
 * Per-node semantics are those of dbhip_arith / dbhip_cmp / dbhip_decimal_arith (same reference lines): `type`
 * is the node's result type and must be the ResultTypeOfBinary entry for the operand registers' types
 * (checked). PLUS / MINUS / MULTIPLY / DIVIDE with a Decimal operand are binary_decimal
 * (decimal/src/arithmetic.rs:190-316): the other operand may be an integer (other_to_decimal), `type` must be
 * the storage class dbhip_decimal_result_size gives and `precision` / `scale` (when non-zero) its DecimalSize;
 * up to 6 decimal nodes per program; row errors "Decimal overflow" / "divided by zero" like
 * dbhip_decimal_arith. Comparisons need equal operand types (decimals: equal scales; the planner inserts
 * CASTs; DBHIP_EX_CAST covers the lossless widenings incl. Decimal64 -> Decimal128 at the same scale, anything
 * that can overflow returns DBHIP_ERR_UNSUPPORTED); DIVIDE raises "divided by zero" like dbhip_arith.
 * DBHIP_EX_IF: dst = a ? b : (register imm) — if(cond, then, else) (evaluator.rs:284-305) for branches that
 * cannot raise and a non-nullable condition (else DBHIP_ERR_UNSUPPORTED: the CPU evaluator's lazy branches stay).
 * NULLs: the result is NULL where a nullable input the RESULT depends on is NULL (passthrough_nullable,
 * register_vectorize.rs:447-471), and a node raises only for rows where the nullable inputs IT depends on are
 * valid. Boolean AND / OR are evaluated strictly (NULL as soon as an operand is NULL) while the reference's are three-valued
 * (FALSE AND NULL = FALSE, TRUE OR NULL = TRUE; and_filters / or_filters, evaluator.rs:284-305): the compiler therefore accepts
 * them over NULLABLE operands only where the two agree — an AND (chain) that ends in the FILTER of a fused aggregation, where
 * NULL and FALSE both drop the row — and returns DBHIP_ERR_UNSUPPORTED for OR over a nullable operand and for an AND over a
 * nullable operand that feeds anything else (a value result, NOT, if, a comparison). DBHIP_EX_IS_TRUE makes an operand
 * non-nullable the way the reference's filters do (NULL -> FALSE): and_filters / or_filters over nullable predicates are
 * AND / OR over IS_TRUE operands, anywhere in a program.
 * Outputs: `out_values` = elements of the out register's type (Decimal128: i128), or for a Boolean result an
 * LSB-first bitmap; `out_validity` likewise a bitmap. Bitmaps are written as whole 64-bit words: both buffers must
 * hold ceil(n/64)*8 bytes and be 8-byte aligned; bits past n are zero. `sum_out_dev` (may be NULL): the
 * wrapping i64/u64 (or f64) sum of the out register over the non-NULL rows is ADDED to *sum_out_dev —
 * `SELECT sum(<expr>)` (BASELINE configs[0]) without materialising <expr>; out_values may then be NULL. */
typedef enum {
  DBHIP_EX_LOAD = 0,     /* dst <- input column `a`                                 */
  DBHIP_EX_CONST = 1,    /* dst <- imm (i64 / u64 / Decimal64 value, or f64 bits for F32/F64) */
  DBHIP_EX_PLUS = 2, DBHIP_EX_MINUS = 3, DBHIP_EX_MULTIPLY = 4, DBHIP_EX_DIVIDE = 5,
  DBHIP_EX_EQ = 6, DBHIP_EX_NOTEQ = 7, DBHIP_EX_LT = 8, DBHIP_EX_LTE = 9, DBHIP_EX_GT = 10, DBHIP_EX_GTE = 11,
  DBHIP_EX_AND = 12, DBHIP_EX_OR = 13, DBHIP_EX_NOT = 14, DBHIP_EX_CAST = 15,
  DBHIP_EX_IF = 16,      /* dst <- a ? b : register (imm & 0xFF)                     */
  DBHIP_EX_IS_TRUE = 17  /* dst <- a is TRUE (a NULL or FALSE operand gives FALSE, the result is never NULL):
                          * FilterHelpers::decode_predicate (utils/filter_helper.rs), what and_filters / or_filters apply to every
                          * argument (evaluator.rs:1815-1880) — or_filters(p, q) = OR(IS_TRUE p, IS_TRUE q) */
} dbhip_expr_op;
typedef struct {
  int32_t op;            /* dbhip_expr_op                                           */
  int32_t dst, a, b;     /* registers (a = input column index for LOAD)             */
  int32_t type;          /* dbhip_type of the result                                */
  uint8_t precision, scale;  /* DecimalSize of a decimal result / constant (0,0: derive) */
  uint8_t _pad[2];
  uint64_t imm;
} dbhip_expr_ins;
int32_t dbhip_expr_eval(const dbhip_expr_ins* prog_host, int32_t n_ins, const dbhip_col* inputs_host,
                        int32_t n_inputs, int64_t n, int32_t out_reg, void* out_values,
                        uint8_t* out_validity, uint8_t* err_bitmap, uint64_t* err_count_dev,
                        void* sum_out_dev, void* stream);

/* ---- a4: decimal arithmetic ------------------------------------------------
 * Replaces binary_decimal (decimal/src/arithmetic.rs:190-316) after the operands
 * were brought to (left_size, right_size) by ArithmeticOp::result_size (:80-139).
 * `lhs`/`rhs` are DEC64/DEC128/DEC256 columns (or integer columns, converted like
 * other_to_decimal) carrying their own precision/scale; the result storage class
 * (precision <= 18: DEC64, <= 38: DEC128, else DEC256) and DecimalSize come from
 * dbhip_decimal_result_size (clamped to 38 digits when both operands have at most 38,
 * to 76 otherwise, arithmetic.rs:115-121). Row errors as dbhip_arith
 * ("Decimal overflow", "Decimal multiply overflow", "divided by zero",
 * "Decimal div overflow"). The Decimal256 class (T = i256: types/decimal.rs:1282-1500 —
 * wrapping ethnum arithmetic on the checked path, exact BigInt fallback when the 256-bit
 * product overflows, from_bigint) is evaluated with 32-bit-limb long division, one row per lane. */
int32_t dbhip_decimal_result_size(int32_t op, uint8_t lp, uint8_t ls, uint8_t rp, uint8_t rs,
                                  uint8_t* out_precision_host, uint8_t* out_scale_host);
int32_t dbhip_decimal_arith(int32_t op, const dbhip_col* lhs, const dbhip_col* rhs,
                            int64_t n, int32_t out_type, uint8_t out_precision,
                            uint8_t out_scale, void* out,
                            uint8_t* err_bitmap, uint64_t* err_count_dev, void* stream);

/* unary minus on a decimal column (register_decimal_minus, decimal/src/arithmetic.rs:514-590): `-t` in the column's own storage
 * class (wrapping), the DecimalSize is unchanged; `out` holds n values of src->type. Validity passes through (the binding reuses it). */
int32_t dbhip_decimal_neg(const dbhip_col* src, int64_t n, void* out, void* stream);
/* to_decimal(p, s) / try_to_decimal(p, s) for DECIMAL and INTEGER sources (CAST(x AS Decimal(p, s)); decimal/src/cast.rs:470-483:
 * decimal_to_decimal :981-1035 — expand / shrink by storage class, scale increase with a checked multiply, scale reduction that
 * truncates or, with `rounding_mode` (the numeric_cast_option setting), rounds half away from zero :790-899 — and
 * integer_to_decimal :701-753). `dst_type` must be the storage class of dst_precision. A value the destination cannot hold
 * raises the row error "Decimal overflow" — bit i of `bitmap` (preset to ones here; LSB-first, whole 64-bit words, 8-byte
 * aligned) cleared, *err_count_dev incremented, the row holds 1 (T::one()); NULL input rows never raise — or, for is_try,
 * becomes NULL: `bitmap` is then the result's validity (input validity AND "representable"). Float / String / Variant sources:
 * DBHIP_ERR_UNSUPPORTED (not on the hot path). */
int32_t dbhip_decimal_cast(const dbhip_col* src, int32_t dst_type, uint8_t dst_precision, uint8_t dst_scale, int32_t is_try,
                           int32_t rounding_mode, int64_t n, void* out, uint8_t* bitmap, uint64_t* err_count_dev, void* stream);

/* ---- a5: comparisons -> Bitmap --------------------------------------------
 * Replaces vectorize_cmp_2_arg + Bitmap::collect_bool
 * (register_comparison.rs:52-96, bitmap/immutable.rs:474). Both sides must have
 * the same physical type (the planner inserts casts) — except two DECIMAL columns (DEC64 / DEC128 / DEC256), which may differ
 * in storage class and DecimalSize (no cast is planned for them: they compare at the larger scale in the storage class of
 * calc_size, decimal/src/comparison.rs:326-441, a side that overflows there ordering by its sign); floats compare as
 * OrderedFloat (NaN largest, types/number.rs:47-48). `out_bitmap` holds
 * ceil(n/8) bytes, LSB-first, trailing bits zero. */
int32_t dbhip_cmp(int32_t op, const dbhip_col* lhs, const dbhip_col* rhs, int64_t n,
                  uint8_t* out_bitmap, void* stream);
/* Bitmap AND (is_or = 0) / OR (1) / AND NOT (2: a & ~b) for and_filters / or_filters / validity merging (evaluator.rs:284-305,
 * 1815-1880: or_filters narrows the validity of its later arguments to the rows that are not TRUE yet). */
int32_t dbhip_bitmap_binary(int32_t is_or, const uint8_t* a, const uint8_t* b, int64_t n,
                            uint8_t* out, void* stream);
int32_t dbhip_bitmap_count(const uint8_t* bitmap, int64_t bit_offset, int64_t n,
                           uint64_t* out_count_dev, void* stream);
/* bitmap[idx[i]] |= 1 for every i (indices >= nbits are ignored; the caller zeroes the bitmap, which must be 4-byte aligned and
 * hold ceil(nbits / 32) words). The "this probe row kept a pair" marks of a join with another conjunct: after the joined rows
 * went through the conjunct's filter the surviving pairs' probe rows are marked (left_join.rs:273-288 conjunct_unmatched[row] = 1,
 * left_join_semi.rs:335-350, left_join_anti.rs:287-300), the unmarked rows are the unmatched ones. */
int32_t dbhip_bitmap_set_indices(const uint32_t* idx, int64_t n_idx, uint8_t* bitmap, int64_t nbits, void* stream);

/* ---- a6: filter -> selection vector, take ---------------------------------
 * Replaces FilterExecutor::filter/select (filter/filter_executor.rs:81-118) for a
 * boolean predicate column: ascending u32 row ids of set bits (Selector output
 * order), count written to *out_count_dev. Scratch is managed internally. */
int32_t dbhip_filter_select(const uint8_t* bitmap, int64_t bit_offset, int64_t n,
                            uint32_t* out_sel, uint64_t* out_count_dev, void* stream);
/* Selector (filter/selector.rs:64-330, filter/select_value/select_column.rs, select_column_scalar.rs): the short-circuit form of a
 * filter. A comparison (or a Boolean column) is evaluated ONLY on the rows of `sel_in` — the true list of the conjunct before
 * (SelectStrategy::True), the false list of the disjunct before (SelectStrategy::False) — or on all `n` rows (sel_in = NULL,
 * SelectStrategy::All); the rows that pass go to `out_true` in order, and, when `out_false` is given (the caller is inside an OR,
 * `has_false`), the others to `out_false` in order. A NULL operand row does not pass (validity && cmp). *out_count_true_dev = number
 * of true rows (the false list holds n - that). Operands: equal fixed-width types (numbers, Date, Timestamp, Decimal64/128 at one
 * scale, Boolean), either may be a scalar; other shapes return DBHIP_ERR_UNSUPPORTED and go through dbhip_cmp + dbhip_filter_select.
 * The lists must not alias `sel_in` (the reference compacts in place; a device kernel cannot). The AND / OR walk over the lists
 * (process_and / process_or) is host logic: databend_amd/host/dbhip_host.hpp Selector, databend_amd/device.py select_and / select_or. */
int32_t dbhip_select_cmp(int32_t op, const dbhip_col* lhs, const dbhip_col* rhs, const uint32_t* sel_in, int64_t n, uint32_t* out_true,
                         uint32_t* out_false, uint64_t* out_count_true_dev, void* stream);
int32_t dbhip_select_bool(const dbhip_col* predicate, const uint32_t* sel_in, int64_t n, uint32_t* out_true, uint32_t* out_false,
                          uint64_t* out_count_true_dev, void* stream);
/* DataBlock::take (kernels/take.rs:43): out[i] = src[sel[i]], elem_size in
 * {1,2,4,8,16} (16 covers i128 and string views). */
int32_t dbhip_take(const void* src, int32_t elem_size, const uint32_t* sel, int64_t n_sel,
                   void* out, void* stream);
/* The same gather over up to 8 columns of one block with ONE selection (DataBlock::take takes every column of the block,
 * kernels/take.rs:43): one launch, the selection read once. elem_sizes_host[k] in {1,2,4,8,16}; Bitmap columns (bool / validity)
 * still go through dbhip_take_bitmap. */
int32_t dbhip_take_block(const void* const* srcs_host, const int32_t* elem_sizes_host, int32_t ncols, const uint32_t* sel, int64_t n_sel,
                         void* const* outs_host, void* stream);
/* take for Bitmap columns (bool / validity). */
int32_t dbhip_take_bitmap(const uint8_t* src, int64_t bit_offset, const uint32_t* sel,
                          int64_t n_sel, uint8_t* out, void* stream);

/* The other gathers of the kernels module, as selection vectors for dbhip_take / dbhip_take_bitmap:
 *   dbhip_sel_from_ranges   DataBlock::take_ranges (kernels/take_ranges.rs:40): `ranges_host` = n_ranges pairs (start, end),
 *                           the selection is the concatenation of the row ranges [start, end) (the block reader's pruned
 *                           row ranges); their lengths must add up to num_rows.
 *   dbhip_sel_from_repeats  DataBlock::take_compacted_indices (kernels/take_compact.rs:38): `repeats_host` = n pairs
 *                           (row, count) = RepeatIndex, row `row` appears `count` times in a row (the hash join's probe
 *                           side); the counts must add up to num_rows.
 *   dbhip_take_chunks       DataBlock::take_blocks / take_column_vec (kernels/take_chunks.rs:70-190): `pairs` = n device
 *                           pairs (block, row) = BlockIndex, out[i] = blocks[block][row] for ONE column of several blocks
 *                           (`blocks_host`: n_blocks device pointers; elem_size 1/2/4/8/16, or 0 for Bitmap columns,
 *                           where a NULL block pointer means "no validity: all valid"). */
int32_t dbhip_sel_from_ranges(const uint32_t* ranges_host, int32_t n_ranges, uint32_t* out_sel, int64_t num_rows,
                              void* stream);
int32_t dbhip_sel_from_repeats(const uint32_t* repeats_host, int32_t n_repeats, uint32_t* out_sel, int64_t num_rows,
                               void* stream);
/* The nullable side of an outer join (new_hash_join/memory/left_join.rs:185-260: build columns of matched rows are wrapped
 * with a true validity, unmatched probe rows get a null block): out[i] = src[idx[i]], validity = the source row's (true when
 * `src_validity` is NULL); idx[i] == 0xFFFFFFFF -> a zero value with validity 0. `out_validity`: LSB-first bits, 8-byte
 * aligned, ceil(n / 8) bytes. elem_size 1 / 2 / 4 / 8 / 16. Left-outer / semi / anti joins are assembled from
 * dbhip_join_probe (pairs), dbhip_join_probe_mark (matched Bitmap), dbhip_bitmap_binary / dbhip_filter_select (the
 * unmatched rows), dbhip_take (probe side) and this (build side): databend_amd/device.py HashJoin.join. */
int32_t dbhip_take_outer(const void* src, const uint8_t* src_validity, int64_t src_validity_offset, int32_t elem_size,
                         const uint32_t* idx, int64_t n, void* out, uint8_t* out_validity, void* stream);
int32_t dbhip_take_chunks(const void* const* blocks_host, int32_t n_blocks, int32_t elem_size, const uint32_t* pairs,
                          int64_t n, void* out, void* stream);

/* ---- a7: group hash ---------------------------------------------------------
 * Replaces group_hash_entries (aggregate/group_hash.rs:40-61): per-row u64 over
 * `ncols` key columns, combined as h = h*NULL_HASH_VAL ^ h_col (:509-511), NULL ->
 * 0xd1cefa08eb382d69 (:38,180-207), ints via the 2-round multiply-xorshift
 * (:555-570), bytes/i128 via the Murmur64A variant (:522-553), floats by bits
 * with canonical NaN (:599-620), bool 0/1 (:581-585). */
int32_t dbhip_group_hash(const dbhip_col* cols, int32_t ncols, int64_t n,
                         uint64_t* out_hashes, void* stream);

/* ---- §8e: scatter indices of the hash-shuffle exchange -----------------------
 * dbhip_siphash64 replaces the `siphash64` / `siphash` scalar function (src/query/functions/src/scalars/hash.rs:50-122,323-328,
 * DFHash :436-545; decimals scalars/decimal/src/hash.rs:144-160): SipHash-1-3 with keys (0, 0) over the value's bytes — integers,
 * Date, Timestamp and float bit patterns in little-endian native width, Boolean as one byte, String as its bytes, Decimal
 * (precision <= 38; `precision` / `scale` of the column must be set) as the scale byte followed by the i128 value. out[i] under a
 * NULL row is 0 (passthrough_nullable: the caller keeps the column's validity).
 * dbhip_scatter_indices replaces HashFlightScatter / OneHashKeyFlightScatter::scatter_indices
 * (src/query/service/src/servers/flight/v1/scatter/flight_scatter_hash.rs:57-330): one key -> siphash64(key) % scatter_size with a
 * NULL key going to `default_index`; several keys -> every key's siphash64 (NULL -> 0) written into a DefaultHasher (SipHash-1-3,
 * zero keys) and finish() % scatter_size. out_index[i] = destination of row i, out_counts (DEVICE, scatter_size u64) = rows per
 * destination. DataBlock::scatter itself = dbhip_scatter_block below.
 * The values equal the reference's bit for bit (its golden file hash.txt), so a GPU node routes rows like the CPU nodes do. */
int32_t dbhip_siphash64(const dbhip_col* col, int64_t n, uint64_t* out, void* stream);
/* DataBlock::scatter(block, indices, scatter_size) (src/query/expression/src/kernels/scatter.rs) for the value buffers of up to any
 * number of columns: outs[c] = the rows of srcs[c] grouped by index[i] (every index < scatter_size), rows keeping their order
 * inside a destination; destination d's rows start at sum(counts[0..d)) (dbhip_scatter_indices / dbhip_sort_bound_partition give
 * the counts). Up to 256 destinations and elements of 1 / 2 / 4 / 8 bytes move in ONE pass per column (per-tile histogram of
 * the indices, scan, stable in-LDS ranking); more destinations or 16-byte elements go through the stable permutation + gather.
 * Bitmap columns: scatter the bytes of an unpacked image or use the permutation path of the caller. */
int32_t dbhip_scatter_block(const void* const* srcs_host, const int32_t* elem_sizes_host, int32_t ncols, const uint32_t* index, int64_t n,
                            uint32_t scatter_size, void* const* outs_host, void* stream);
int32_t dbhip_scatter_indices(const dbhip_col* keys, int32_t nkeys, int64_t n, uint32_t scatter_size, uint64_t default_index,
                              uint32_t* out_index, uint64_t* out_counts, void* stream);
/* DataBlock::scatter over WHOLE columns (kernels/scatter.rs:20-66: divide_indices_by_scatter_size + take of every column) — values,
 * validity Bitmaps, Boolean, String (views; the data buffers are shared with the source, as take.rs does for view columns) and
 * Decimal256 columns: destination d holds the rows whose index is d, in row order. out_row_starts_host[scatter_size + 1]: destination
 * d's rows are [starts[d], starts[d + 1]) of every output value buffer (n elements each). Bitmaps — the values of a Boolean column and
 * the validity of a nullable one (out_validity_host[c] may be NULL for a column without validity) — are written as ONE STAND-ALONE,
 * offset-0 Bitmap per destination: destination d's starts at byte 8 * (starts[d] / 64 + d) of the output buffer, which must hold
 * 8 * (n / 64 + scatter_size + 1) bytes and be 8-byte aligned (destinations never share a 64-bit word, so every destination's columns
 * can go straight into any other entry point). An index >= scatter_size is DBHIP_ERR_INVALID. Checked against the reference's
 * fixture (kernel-pass.txt 'Scatter') and its round-trip property scatter -> concat == take (tests/it/kernel.rs:519-566). */
int32_t dbhip_scatter_columns(const dbhip_col* cols, int32_t ncols, const uint32_t* index, int64_t n, uint32_t scatter_size,
                              void* const* out_data_host, uint8_t* const* out_validity_host, int64_t* out_row_starts_host, void* stream);
/* DataBlock::concat for ONE column of `nblocks` blocks (kernels/concat.rs:62-340): cols[b] (rows_host[b] rows) back to back into
 * out_data; out_validity (needed when any block has a validity; a block without one counts as all valid) and Boolean values are
 * concatenated bit by bit from any bit offset (validity_offset; bool_bit_offsets_host[b] for the values of a Boolean block, NULL = 0)
 * into 8-byte aligned Bitmaps of ceil(total / 64) * 8 bytes. String blocks: the views are copied with the buffer index of long values
 * rebased, and the blocks' buffer tables are written back to back into out_buffers_dev (device array of sum(n_buffers) pointers;
 * *out_n_buffers_host = that sum) — the bytes themselves do not move (the reference's builder copies them; column VALUES are equal).
 * A constant (is_scalar) block of a fixed-width type is expanded. */
int32_t dbhip_concat_columns(const dbhip_col* cols, const int64_t* rows_host, const int64_t* bool_bit_offsets_host, int32_t nblocks,
                             void* out_data, uint8_t* out_validity, const void** out_buffers_dev, int32_t* out_n_buffers_host, void* stream);

/* ---- a8-a13: hash aggregation ---------------------------------------------
 * Replaces AggregateHashTable behind TransformPartialAggregate /
 * TransformFinalAggregate (aggregate_hashtable.rs:168-408,
 * transform_aggregate_partial.rs:179-303, transform_aggregate_final.rs:510+).
 * The table lives in HBM; aggregate states are fixed-width words next to the
 * group's key row (the device analogue of Payload rows, payload.rs:47-72).
 * Output of flush == the serialized-state block of Payload::aggregate_flush
 * (payload_flush.rs:151-181): one column per state + the group key columns +
 * the group hash column. Group order is unspecified (reference tests compare as
 * sorted sets: tests/it/aggregates/agg_hashtable.rs assert_block_value_sort_eq). */
typedef enum {
  DBHIP_AGG_COUNT = 0,       /* count(*) / count(col): u64 (aggregate_count.rs:47-150) */
  DBHIP_AGG_SUM = 1,         /* sum: i64/u64 wrapping, f64, DEC64->i64 wrapping,
                                DEC128 -> i128 with overflow check when arg precision>18
                                (aggregate_sum.rs:51-300,386-441)                    */
  DBHIP_AGG_MIN = 2, DBHIP_AGG_MAX = 3  /* fixed-width arguments incl. Decimal128 (a three-word state merged under a per-state lock: such tables
                                          * aggregate on the row path only); String: DBHIP_ERR_UNSUPPORTED */
} dbhip_agg_kind;

typedef struct {
  int32_t kind;       /* dbhip_agg_kind                                            */
  int32_t arg_type;   /* dbhip_type of the argument (ignored for count(*))         */
  uint8_t arg_precision, arg_scale; /* decimals                                    */
  uint8_t arg_nullable;             /* NULL rows are skipped; sum/min/max then carry a "seen a value" flag and
                                       yield NULL for all-NULL groups (flush_result_nullable)          */
  uint8_t _pad;
} dbhip_agg_desc;

typedef struct dbhip_groupby dbhip_groupby;  /* opaque */

/* key_types: fixed-width types and DBHIP_T_STRING. Strings of up to 12 bytes stay inline in the group's row; longer ones
 * are copied into the table's ARENA (a device bump allocator, the analogue of the Payload's arena: the row holds
 * `(len, prefix, offset)` where the reference holds `(len, ptr)`, payload.rs:361-486, payload_row.rs:85-215) and compared by
 * length, prefix and bytes (row_match_entries, payload_row.rs:324+). A block with long string keys is aggregated on the row
 * path (the LDS / partitioned pre-aggregation kernels hand it over).
 * `initial_capacity`: slots to start with (the table grows), and the caller's ESTIMATE of the number of groups (the
 * reference sizes its tables from the planner's cardinality estimate the same way): a first block of >= 1 M rows with at most
 * twice as many rows as this estimate is taken as "about one group per row" and skips pre-aggregation (a join's output grouped
 * by the join key). Any value is correct; only the path differs. */
int32_t dbhip_groupby_create(const int32_t* key_types_host, const uint8_t* key_nullable_host,
                             int32_t nkeys, const dbhip_agg_desc* aggs_host, int32_t naggs,
                             int64_t initial_capacity, dbhip_groupby** out_host);
/* AggregateHashTable::add_groups: `args[i]` is the argument column of aggs[i]
 * (ignored for count(*)). Grows (rehashes) like resize() (:463-490) when the load
 * factor 1/1.35 (mod.rs:55) would be exceeded. */
int32_t dbhip_groupby_add_block(dbhip_groupby* g, const dbhip_col* keys, const dbhip_col* args,
                                int64_t n, void* stream);
/* The same with the TransformFilter in front of the aggregate pushed down (filter/filter_executor.rs:81-118 produces a
 * selection that DataBlock::take then applies to every column): `filter_bitmap` (LSB-first, read from bit
 * `filter_bit_offset`, NULL = every row) is the predicate's Bitmap over the UNFILTERED block; rows whose bit is 0 do not
 * take part. Keys and arguments are the unfiltered columns: nothing is compacted or copied. Result == add_block on the
 * taken columns. */
int32_t dbhip_groupby_add_block_filtered(dbhip_groupby* g, const dbhip_col* keys, const dbhip_col* args,
                                         int64_t n, const uint8_t* filter_bitmap, int64_t filter_bit_offset,
                                         void* stream);
/* Fused TransformFilter -> BlockOperator::Map -> TransformPartialAggregate for tables with a handful of groups
 * (filter/filter_executor.rs:81-118, sql/src/evaluator/block_operator.rs:42-85, aggregate_hashtable.rs:168-292): ONE
 * pass over the unfiltered input columns. `prog` is a dbhip_expr_eval program over `inputs` whose results are the
 * filter predicate (`filter_reg`, -1 = none; a NULL predicate drops the row) and one argument per aggregate
 * (`arg_regs[i]`: a register, DBHIP_ARG_INPUT(c) = input column c as it is, DBHIP_ARG_NONE = count(*)); maps raise row
 * errors only for rows the filter kept (the filter precedes the maps in the reference's pipeline).
 * `filter_bitmap` (may be NULL) is an additional pushed-down predicate Bitmap as in add_block_filtered. Keys are
 * columns (<= 4 key words; strings up to 12 bytes). Groups resolve through a per-workgroup key table of 8 slots and
 * the states accumulate in per-lane registers: one pass over the inputs, nothing materialised (measured: DESIGN.md 2.2 —
 * the interpreted program is VALU bound, ~7x the hand-written Q1 kernel).
 * Result == add_block over the taken, mapped columns. Returns DBHIP_ERR_CAPACITY when a workgroup met more than 8
 * groups and DBHIP_ERR_ROW_ERRORS when a map raised (in both cases NOTHING was merged: the caller runs the
 * operator-at-a-time kernels on the block), DBHIP_ERR_UNSUPPORTED for layouts / programs outside the fused subset
 * (<= 8 aggregates, <= 12 state words, the dbhip_expr_eval subset). */
#define DBHIP_ARG_NONE INT32_MIN
#define DBHIP_ARG_INPUT(c) (-(1 + (c)))
typedef struct {
  const dbhip_expr_ins* prog;   /* host array, may be NULL when n_ins == 0 (arguments are input columns) */
  int32_t n_ins;
  const dbhip_col* inputs;      /* host array of n_inputs columns (<= 8)                                   */
  int32_t n_inputs;
  int32_t filter_reg;           /* register of the Boolean predicate, -1 = none                            */
  const int32_t* arg_regs;      /* host array, one entry per aggregate of the table                         */
} dbhip_agg_program;
int32_t dbhip_groupby_add_block_program(dbhip_groupby* g, const dbhip_col* keys, const dbhip_agg_program* prog,
                                        int64_t n, const uint8_t* filter_bitmap, int64_t filter_bit_offset,
                                        void* stream);
/* Run-time specialisation. add_block_program interprets the program. prepare_program — the PREPARE of a pipeline, same
 * arguments (column data pointers may be NULL, types must be final) — compiles, through hiprtc, the kernel's own source with
 * THIS program / layout as compile-time constants (what rustc's monomorphisation gives the reference's kernels): ~0.5 s on the
 * calling thread, cached per query shape for the life of the process. Afterwards add_block_program calls of that shape launch
 * the specialised kernel (3.6x faster on TPC-H Q1); shapes that were not prepared are interpreted — a query
 * never waits for a compiler. If the kernel cannot be built the interpreter stays (DBHIP_OK either way).
 * env DBHIP_FAGG_JIT=0 disables, =sync compiles on first use instead. */
int32_t dbhip_groupby_prepare_program(dbhip_groupby* g, const dbhip_col* keys, const dbhip_agg_program* prog);
/* PIPELINED fused aggregation (round 6): the call shape for a host that hands over the reference's own DataBlocks — <= 65,536
 * rows each (max_block_size, src/query/settings/src/settings_default.rs:142-148), one TransformPartialAggregate::transform per
 * block (transform_aggregate_partial.rs:262-270). At that size the kernel is a few microseconds of work and a synchronous call is
 * all round trip. After set_pipelined(g, 1) a dbhip_groupby_add_block_program call queues ONE kernel launch and returns: the
 * partial rows of successive blocks are appended to a buffer the table owns, merged window by window by kernels queued behind
 * them, and nothing is read back (argument / program errors are still returned by the call itself).
 * dbhip_groupby_checkpoint drains the table's stream and reports what a synchronous call would have: DBHIP_OK, or
 * DBHIP_ERR_CAPACITY / DBHIP_ERR_ROW_ERRORS / DBHIP_ERR_UNSUPPORTED with *out_blocks_committed_host = the number of blocks (counted
 * from the previous checkpoint, in call order) that WERE merged. Windows commit in order, and a window in which any block raised a
 * flag — and every block queued behind it — merges nothing: blocks [committed, queued) go to the operator-at-a-time path exactly
 * like a block the synchronous call gave back. (A "fifth group in one workgroup" give-up of the 4-slot kernel is replayed inside the
 * checkpoint with the 8-slot kernel first, as the synchronous call's second pass does.)
 * Contract: the blocks' buffers (columns, validity, filter Bitmaps) stay alive and unchanged until the checkpoint — they are
 * inputs of queued kernels; plain dbhip_groupby_add_block / _add_block_filtered calls on a pipelined table are queued the same way (the
 * block's columns as a program without instructions) while layout and columns qualify for the few-groups kernel — otherwise, and
 * after a checkpoint has returned DBHIP_ERR_CAPACITY, they checkpoint and take the synchronous paths; one stream per pipelined table between checkpoints (the reference's partial tables are per pipeline
 * thread as well). Every other entry point that reads or changes the table's groups (flush_*, num_groups, merge_*, plain
 * add_block, the exchange calls) checkpoints first and returns the checkpoint's error if there is one; dbhip_groupby_reset drops
 * queued blocks with the groups. set_pipelined(g, 0) checkpoints and returns to synchronous calls.
 * Measured (profiles/r06_block_size_sweep.json): DESIGN.md 2.2b. */
int32_t dbhip_groupby_set_pipelined(dbhip_groupby* g, int32_t on, void* stream);
int32_t dbhip_groupby_checkpoint(dbhip_groupby* g, int64_t* out_blocks_committed_host, void* stream);
/* combine_payload (:349-380): merge serialized partial states (as produced by
 * dbhip_groupby_flush_serialized on any rank) into this table. */
int32_t dbhip_groupby_merge_serialized(dbhip_groupby* g, const void* rows_dev, int64_t n_rows,
                                       void* stream);
/* Long string keys. String key columns written by flush_result / flush_state_block are 16-byte views; the long form is
 * {len, prefix, buffer 0, offset}: buffer 0 of such a column is the table's arena (dbhip_groupby_arena: device pointer and
 * bytes in use; valid until the table is reset, destroyed or takes more rows). Serialized rows (flush_serialized) carry the
 * same offsets: ship the arena with them and merge with merge_serialized_arena (`arena_dev` = the SENDER's arena bytes on this
 * device); merge_serialized alone is for tables without long strings. The fixed-block exchange entry points (flush_block,
 * merge_blocks, partition_blocks, replace_with_blocks, flush_partitioned) move rows WITHOUT an arena and return
 * DBHIP_ERR_UNSUPPORTED for a table that holds a string key longer than 12 bytes. */
int32_t dbhip_groupby_arena(dbhip_groupby* g, const void** out_ptr_host, int64_t* out_bytes_host, void* stream);
int32_t dbhip_groupby_merge_serialized_arena(dbhip_groupby* g, const void* rows_dev, int64_t n_rows, const void* arena_dev,
                                             void* stream);
int32_t dbhip_groupby_num_groups(dbhip_groupby* g, int64_t* out_host, void* stream);
/* Bytes per serialized row: [keys (fixed width, strings as 16-B inline views)]
 * [validity byte per nullable key][hash u64][state words]. */
int32_t dbhip_groupby_row_bytes(dbhip_groupby* g, int64_t* out_host);
int32_t dbhip_groupby_flush_serialized(dbhip_groupby* g, void* out_rows_dev, int64_t max_rows,
                                       int64_t* out_n_rows_host, void* stream);
/* Fixed-size exchange of partial states between ranks (the RCCL analogue of the
 * AggregateMeta shuffle, aggregator/aggregate_exchange_injector.rs:57-147, for low cardinality).
 * A block is (max_rows + 1) rows of dbhip_groupby_row_bytes: row 0 is the header (u64 word 0 = rows
 * that follow, ~0 = the table held more than max_rows groups), rows 1.. are serialized rows.
 * flush_block writes this table's block WITHOUT any host synchronisation, so that one
 * all_gather of equal-size blocks can follow on the same stream; merge_blocks merges `n_blocks`
 * gathered blocks except `skip_block` (the caller's own, whose states are still in the table; -1 =
 * none). If a block overflowed it returns DBHIP_ERR_CAPACITY before touching the table. */
int32_t dbhip_groupby_flush_block(dbhip_groupby* g, void* out_block_dev, int64_t max_rows, void* stream);
int32_t dbhip_groupby_merge_blocks(dbhip_groupby* g, const void* blocks_dev, int32_t n_blocks,
                                   int64_t max_rows, int32_t skip_block, void* stream);
/* a12 — hash partitioning of the table's group rows for the exchange / final merge: the device twin of
 * Payload::scan_hash_partition_transfer (payload.rs:548-589: bucket = group hash % bucket count, strength-reduced there)
 * and PartitionedPayload::repartition (partitioned_payload.rs:204-240). The unit is the serialized row
 * (dbhip_groupby_row_bytes).
 *   partition_blocks   n_buckets fixed-size blocks back to back in `out_blocks_dev`, each (max_rows + 1) rows: row 0 =
 *                      header (u64 word 0 = rows that follow, ~0 = more than max_rows; word 1 = 1 when ANY block of
 *                      this table overflowed), rows 1.. = the bucket's rows. No host synchronisation: one
 *                      all_to_all_single with equal splits can follow on the same stream.
 *   replace_with_blocks  the receiving side: checks the n_blocks headers (returns DBHIP_ERR_CAPACITY before touching
 *                      the table if any sender overflowed — every rank sees the same flags), then RESETS the table and
 *                      merges all blocks: the rank now holds exactly the groups with hash % world == rank.
 *   flush_partitioned  variable-length form: all rows grouped by bucket, bucket b = rows [sum(counts[0..b)),
 *                      +counts[b]) of out_rows_dev; out_counts_host[n_buckets] (the all-to-all split sizes). */
int32_t dbhip_groupby_partition_blocks(dbhip_groupby* g, int32_t n_buckets, void* out_blocks_dev, int64_t max_rows,
                                       void* stream);
int32_t dbhip_groupby_replace_with_blocks(dbhip_groupby* g, const void* blocks_dev, int32_t n_blocks, int64_t max_rows,
                                          void* stream);
int32_t dbhip_groupby_flush_partitioned(dbhip_groupby* g, int32_t n_buckets, void* out_rows_dev, int64_t max_rows,
                                        int64_t* out_counts_host, void* stream);
/* merge_result (:382-408): final values as columns. out_keys[i]/out_aggs[i] are
 * device buffers of max_rows elements of the key / result type
 * (dbhip_groupby_result_type). Returns DBHIP_ERR_OVERFLOW if a checked decimal
 * sum left [DECIMAL_MIN, DECIMAL_MAX]. */
int32_t dbhip_groupby_result_type(const dbhip_agg_desc* agg_host, int32_t* out_type_host,
                                  uint8_t* out_precision_host, uint8_t* out_scale_host);
int32_t dbhip_groupby_flush_result(dbhip_groupby* g, void* const* out_keys_host,
                                   uint8_t* const* out_key_validity_host,
                                   void* const* out_aggs_host, uint64_t* out_hashes,
                                   int64_t max_rows, int64_t* out_n_rows_host, void* stream);
/* The same with a validity bitmap per aggregate (out_agg_validity_host[i], LSB-first, ceil(max_rows/64)*8 bytes, may be
 * NULL): sum / min / max over a NULLABLE argument are NULL for a group whose argument was NULL in every row
 * (AggregateNullUnaryAdaptor<true>, adaptors/aggregate_null_adaptor.rs:366-400); count is never NULL. */
int32_t dbhip_groupby_flush_result_nullable(dbhip_groupby* g, void* const* out_keys_host,
                                            uint8_t* const* out_key_validity_host,
                                            void* const* out_aggs_host, uint8_t* const* out_agg_validity_host,
                                            uint64_t* out_hashes, int64_t max_rows, int64_t* out_n_rows_host,
                                            void* stream);
/* §8f-1 — the serialized-state block of Payload::aggregate_flush (payload_flush.rs:151-181): per aggregate the fields
 * of its serialize_type(), flattened in aggregate order, then the group columns:
 *   count                         [UInt64]                                   (aggregate_count.rs:170-186)
 *   sum                           [result type: Int64/UInt64/Float64/Decimal] (aggregate_sum.rs:155-168,281-298)
 *   min / max (numbers, dates)    [Boolean has-value][T value, default if none] (aggregate_min_max_any.rs:315-346)
 *   min / max over Decimal        ONE Nullable(Decimal) column (aggregate_min_max_any_decimal.rs:140-190) — its two buffers as two fields:
 *                                 [Boolean validity = has-value][Decimal values, 0 if none]; Decimal256 values are 32 bytes
 *   min / max over String         ONE Nullable(String) column (aggregate_min_max_any.rs:163-205, NOT borsh) — [Boolean validity]
 *                                 [String values]: 16-byte views, the long form {len, prefix, buffer 0, offset} with buffer 0 = the
 *                                 table's arena (dbhip_groupby_arena); merge_state_block takes such a field as a String dbhip_col with
 *                                 its `buffers`, and copies the winners into the receiving table's arena
 *   sum over Decimal256           [Decimal256 total] (aggregate_sum.rs:281-298 with T = i256; DBHIP_ERR_OVERFLOW outside +-(10^76 - 1))
 *   sum/min/max, NULLABLE argument: the fields above + a trailing [Boolean flag]  (aggregate_null_adaptor.rs:508-540)
 * (A binding that builds the reference's block wraps a [Boolean validity][T values] pair of a Decimal / String min / max into one
 * NullableColumn; for numbers and dates the two fields stay two columns, as the reference's serialize_type says.)
 * Boolean fields are LSB-first bitmaps of ceil(max_rows/64)*8 bytes. dbhip_groupby_state_fields lists the fields of a
 * table (type and owning aggregate). flush_state_block writes the block (a device partial aggregate feeding the
 * UNMODIFIED CPU final stage / Flight exchange); merge_state_block consumes one (TransformDeserializer +
 * AggregateFunction::batch_merge, aggregator/serde/transform_deserializer.rs) — `states` = the flattened fields. */
int32_t dbhip_groupby_state_fields(dbhip_groupby* g, int32_t* out_types_host, int32_t* out_agg_index_host,
                                   int32_t max_fields, int32_t* out_n_fields_host);
int32_t dbhip_groupby_flush_state_block(dbhip_groupby* g, void* const* out_keys_host,
                                        uint8_t* const* out_key_validity_host, void* const* out_state_fields_host,
                                        uint64_t* out_hashes, int64_t max_rows, int64_t* out_n_rows_host, void* stream);
int32_t dbhip_groupby_merge_state_block(dbhip_groupby* g, const dbhip_col* keys, const dbhip_col* states,
                                        int64_t n, void* stream);
int32_t dbhip_groupby_reset(dbhip_groupby* g, void* stream);
int32_t dbhip_groupby_destroy(dbhip_groupby* g);

/* ---- fused TPC-H Q1 pipeline (BASELINE.json configs[1]) --------
 * One pass over lineitem: TransformFilter (l_shipdate <= cutoff,
 * filters/filter_predicate.rs:71-96) -> CompoundBlockOperator decimal maps
 * (1-l_discount, *, 1+l_tax; decimal/arithmetic.rs:155-316) -> TransformPartialAggregate
 * with group keys (l_returnflag, l_linestatus) and states
 * [sum(qty) i64, sum(price) i64, sum(disc_price) i128, sum(charge) i128,
 *  sum(discount) i64, count(*) u64]. Partial states are merged INTO the given
 * group-by table (which must have been created with exactly that key/agg layout:
 * see dbhip_q1_create_groupby), so the same flush/merge/exchange entry points
 * serve the fused and the operator-at-a-time paths. Reads 68 B/row. */
int32_t dbhip_q1_create_groupby(dbhip_groupby** out_host);
int32_t dbhip_q1_fused(dbhip_groupby* g,
                       const int64_t* l_quantity, const int64_t* l_extendedprice,
                       const int64_t* l_discount, const int64_t* l_tax,
                       const void* l_returnflag_views, const void* l_linestatus_views,
                       const int32_t* l_shipdate, int32_t shipdate_cutoff,
                       int64_t n, void* stream);

/* ---- a14: packed fixed-width keys ---------------------------------------------
 * Replaces DataBlock::choose_hash_method_with_types (kernels/group_by.rs:40-80) and
 * HashMethodFixedKeys::build_keys_vec / KeysVec (group_by_hash/method_fixed_keys.rs:58-78,
 * 310-403): numeric / date / timestamp / decimal key columns are stably sorted by byte width
 * (numeric_byte_size, types.rs:606-633), their values written little endian back to back,
 * followed by one null byte per nullable column (1 = NULL, value bytes of a NULL row stay 0);
 * the row's key is that byte string read as ONE integer of 1/2/4/8/16/32 bytes
 * (golden: tests/it/group_by.rs:52-58, three Int8 columns [1,1,1] -> 0x10101).
 * dbhip_keys_method returns the key width the reference would choose (0 = HashMethodSerializer /
 * SingleBinary: dbhip_serialize_keys + dbhip_join_*_binary below). dbhip_pack_keys may be asked for a wider key than needed
 * (zero-extended; the join table takes 8- and 16-byte keys). `out_all_valid` (may be NULL):
 * LSB-first bitmap, bit = no key column is NULL in that row (join keys with a NULL never match). */
int32_t dbhip_keys_method(const dbhip_col* cols, int32_t ncols, int32_t* out_key_bytes_host);
int32_t dbhip_pack_keys(const dbhip_col* cols, int32_t ncols, int64_t n, int32_t key_bytes,
                        void* out_keys, uint8_t* out_all_valid, void* stream);

/* HashMethodSerializer (group_by_hash/method_serializer.rs:33-52, utils.rs:33-160 serialize_group_columns): the key columns of
 * a row serialized back to back into ONE BinaryColumn — the method choose_hash_method_with_types falls to when a key column is
 * not a fixed-width number / date / decimal or the packed key exceeds 32 bytes (dbhip_keys_method returns 0). Per column:
 * numbers / decimals / dates / timestamps = the value's little-endian bytes; Boolean = one byte; String = u64 length + the
 * bytes; a nullable column = one byte `valid` followed by the value only when valid.
 *   dbhip_serialize_keys_offsets  the BinaryColumn's offsets (n + 1 u64 on the device, offsets[0] = 0), the total byte count
 *                                 (host) and, optionally, `out_all_valid` (LSB-first: no key column is NULL in the row — join
 *                                 keys with a NULL never match); synchronises
 *   dbhip_serialize_keys          writes the bytes (out_data holds offsets[n] bytes) */
int32_t dbhip_serialize_keys_offsets(const dbhip_col* cols, int32_t ncols, int64_t n, uint64_t* out_offsets, uint8_t* out_all_valid,
                                     uint64_t* out_total_bytes_host, void* stream);
int32_t dbhip_serialize_keys(const dbhip_col* cols, int32_t ncols, int64_t n, const uint64_t* offsets, uint8_t* out_data, void* stream);

/* ---- a15: hash join ---------------------------------------------------------
 * Replaces HashJoinHashTable<K> build/probe behind trait Join
 * (hash_join_table/hashjoin_hashtable.rs:26-344,
 * new_hash_join/hashtable/fixed_keys.rs:47-269, memory/inner_join.rs:122-271) for
 * KeysU8..U64 (8-byte keys) and KeysU128 (16-byte keys; method_fixed_keys.rs:58-139).
 * `keys` point at n keys of the width the table was created with. Inner join: emits
 * (probe_idx, build_row) pairs. Pair order is unspecified in the reference across threads; this
 * library returns them sorted by (probe_idx, build_row). dbhip_join_probe_mark writes the
 * "probe row has a match" bitmap (LSB-first, ceil(n/8) bytes) that semi / anti joins filter on and
 * left-outer joins use to append unmatched rows (new_hash_join probe_matched,
 * fixed_keys.rs:96-167).
 * Sizing protocol: dbhip_join_probe_count(block) -> allocate -> dbhip_join_probe(same block). The probe that
 * directly follows a count of the same (keys, validity, n) on the same stream reuses the per-row counts (the
 * table is walked once), so the key / validity buffers must not change between the two calls (columns are
 * immutable in the reference; a caller that recycles a staging buffer must count again). */
typedef struct dbhip_join dbhip_join;
int32_t dbhip_join_create(int64_t expected_build_rows, dbhip_join** out_host);  /* 8-byte keys */
int32_t dbhip_join_create_keys(int64_t expected_build_rows, int32_t key_bytes, dbhip_join** out_host);  /* 8 / 16 / 32 (KeysU256) */
int32_t dbhip_join_add_build(dbhip_join* j, const void* keys, const uint8_t* validity,
                             int64_t n, void* stream);
int32_t dbhip_join_finalize(dbhip_join* j, void* stream);
int32_t dbhip_join_probe_count(dbhip_join* j, const void* keys, const uint8_t* validity,
                               int64_t n, uint64_t* out_total_host, void* stream);
int32_t dbhip_join_probe(dbhip_join* j, const void* keys, const uint8_t* validity, int64_t n,
                         uint32_t* out_probe_idx, uint32_t* out_build_row, int64_t max_pairs,
                         uint64_t* out_n_pairs_host, void* stream);
int32_t dbhip_join_probe_mark(dbhip_join* j, const void* keys, const uint8_t* validity, int64_t n,
                              uint8_t* out_matched_bitmap, uint64_t* out_n_matched_host, void* stream);
/* Right / full outer, right semi / right anti joins (new_hash_join/memory/right_join.rs, right_join_semi.rs, right_join_anti.rs,
 * full_join.rs): the table keeps one "matched" bit per BUILD row across all probe blocks (the reference's scan map).
 * dbhip_join_mark_build ORs in the build rows of a probe block's pairs (after whatever conjunct filtering the binding applies);
 * dbhip_join_build_matched copies the bitmap out (LSB-first, ceil(build_rows / 64) * 8 bytes) for final_probe: the unmatched
 * build rows follow with a NULL probe side (right / full), or the matched / unmatched build rows alone (semi / anti). */
int32_t dbhip_join_mark_build(dbhip_join* j, const uint32_t* build_rows, int64_t n_pairs, void* stream);
int32_t dbhip_join_build_matched(dbhip_join* j, uint8_t* out_bitmap, int64_t* out_build_rows_host, void* stream);
int32_t dbhip_join_destroy(dbhip_join* j);

/* Hash join on SERIALIZED keys (HashMethodSerializer: string keys of any length, keys wider than 32 bytes; the reference's
 * BinaryHashJoinHashTable, hash_join_table + new_hash_join/hashtable/serialize_keys.rs): rows are (offsets[n + 1], data) as
 * dbhip_serialize_keys produces them (any BinaryColumn works). Keys are equal iff their bytes are equal: a 128-bit hash of the
 * bytes routes a row through the fixed-key table and every candidate pair is verified byte for byte against the build rows,
 * which the table keeps a copy of. Same pair order as dbhip_join_probe (by probe row, then build row).
 *   probe_count_binary  an UPPER bound of the pairs (candidates by hash) to size the buffers with
 *   probe_binary        the verified pairs, their number, and optionally the "probe row has a match" bitmap (LSB-first,
 *                       ceil(n / 32) * 4 bytes, 4-byte aligned) for semi / anti / left-outer assembly */
typedef struct dbhip_join_binary dbhip_join_binary;
int32_t dbhip_join_create_binary(int64_t expected_build_rows, dbhip_join_binary** out_host);
int32_t dbhip_join_add_build_binary(dbhip_join_binary* j, const uint64_t* offsets, const uint8_t* data, const uint8_t* validity, int64_t n,
                                    void* stream);
int32_t dbhip_join_finalize_binary(dbhip_join_binary* j, void* stream);
int32_t dbhip_join_probe_count_binary(dbhip_join_binary* j, const uint64_t* offsets, const uint8_t* data, const uint8_t* validity, int64_t n,
                                      uint64_t* out_max_pairs_host, void* stream);
int32_t dbhip_join_probe_binary(dbhip_join_binary* j, const uint64_t* offsets, const uint8_t* data, const uint8_t* validity, int64_t n,
                                uint32_t* out_probe_idx, uint32_t* out_build_row, int64_t max_pairs, uint64_t* out_n_pairs_host,
                                uint8_t* out_matched_bitmap, void* stream);
int32_t dbhip_join_destroy_binary(dbhip_join_binary* j);

/* ---- a16: sort / top-k --------------------------------------------------------
 * Replaces DataBlock::sort_with_type / SortCompare
 * (kernels/sort.rs:91-113, kernels/sort_compare.rs:33-283): permutation of row ids
 * ordered by up to 8 key columns — fixed-width types and Strings of any length up to 4096 bytes (memcmp order of the bytes,
 * a proper prefix first: sorts/core/row_convert/variable.rs; values beyond 12 bytes cost one radix key image per 8 bytes of
 * the column's longest value, constant bytes are skipped) — with per-key asc/desc and
 * nulls_first; `limit` (0 = none) keeps the first `limit` rows (LimitType::LimitRows).
 * The reference uses sort_unstable_by, so only the key sequence is part of the
 * contract; this implementation is stable (ties by ascending row id). */
int32_t dbhip_sort_perm(const dbhip_col* keys, const uint8_t* desc_host,
                        const uint8_t* nulls_first_host, int32_t nkeys, int64_t n,
                        int64_t limit, uint32_t* out_perm, void* stream);

/* k-way merge of sorted runs. Replaces Merger / LoserTreeSort / HeapSort
 * (src/query/pipeline/transforms/src/processors/transforms/sorts/core/{merger.rs,algorithm.rs:33-61,
 * loser_tree.rs}, sort_k_way_merge.rs): the runs sit back to back in the key columns, run r = rows
 * [run_offsets[r], run_offsets[r+1]) and each run is ordered by the same keys; out_perm = row ids in
 * merged order (ties: lower run first, then position — equal rows have no defined order between
 * streams in the reference), truncated to `limit` when limit > 0. */
int32_t dbhip_merge_sorted_perm(const dbhip_col* keys, const uint8_t* desc_host,
                                const uint8_t* nulls_first_host, int32_t nkeys,
                                const int64_t* run_offsets_host, int32_t nruns, int64_t limit,
                                uint32_t* out_perm, void* stream);

/* Range partition for the distributed sort (SURVEY §8e "sort": sample -> range-partition -> all-to-all -> local sort). Replaces the
 * cut of sorted streams at Bounds (src/query/pipeline/transforms/src/processors/transforms/sorts/sort_spill.rs:740-1040
 * BoundBlockStream / block_split_off_position / partition_point: rows <= bound[i] in sort order belong to partition i, rows after
 * the last bound to partition nbounds) and SortBoundScatter (src/query/service/src/pipelines/processors/transforms/sort/
 * sort_exchange_injector.rs: partition i goes to node i % n). `bounds` = nbounds rows of the same key columns, ORDERED by the same
 * keys (core/bounds.rs: Bounds::from_column / merge / dedup); out_part[i] = number of bounds that sort strictly before row i;
 * out_counts (DEVICE, nbounds + 1 u64) = rows per partition. The rows need not be sorted. Same key types, desc / nulls_first
 * meaning and string lengths as dbhip_sort_perm. */
int32_t dbhip_sort_bound_partition(const dbhip_col* keys, const dbhip_col* bounds, const uint8_t* desc_host,
                                   const uint8_t* nulls_first_host, int32_t nkeys, int64_t n, int64_t nbounds,
                                   uint32_t* out_part, uint64_t* out_counts, void* stream);

/* ---- a17/a18: vector distance ------------------------------------------------
 * Replaces cosine_distance / l2_distance / inner_product / l1_distance
 * (src/common/vector/src/distance.rs:19-165) driven by
 * functions/src/scalars/vector.rs:497-560, batched over queries:
 * out[q*n + i] = metric(base[i], query[q]) for flat row-major f32
 * (VectorColumn::Float32, types/vector.rs:377-380). The dot products run on
 * v_mfma_f32_32x32x2_f32 (exact f32). */
typedef enum { DBHIP_VEC_COSINE = 0, DBHIP_VEC_L2 = 1, DBHIP_VEC_DOT = 2, DBHIP_VEC_L1 = 3, DBHIP_VEC_NORM = 4 /* vector_norm(lhs): dbhip_vec_distance_rows only */ } dbhip_vec_metric;
/* The scalar functions row by row (scalars/vector.rs:59-260: cosine_distance / l1_distance / l2_distance / inner_product over two
 * Array(Float32) or Array(Float64) COLUMNS; :490-560 calculate_distance / calculate_norm over Vector(Float32 | Int8) columns):
 * out[i] = f(lhs[i], rhs[i]) for dense row-major [n][dim] columns; a side with *_is_scalar set is ONE vector for every row (a constant
 * argument). elem_type DBHIP_T_F32 -> f32 out, DBHIP_T_F64 (the *_64 functions, distance.rs:97-165) -> f64 out, DBHIP_T_I8 (Int8
 * vectors are widened to f32 first) -> f32 out. DBHIP_VEC_NORM ignores rhs. NULL rows / NULL elements are the caller's: the reference
 * raises "Vector contain null values" for an element NULL and passes row NULLs through (the result's validity = the AND of the
 * arguments'). 1e-5 relative to the reference's summation order (north_star). */
int32_t dbhip_vec_distance_rows(int32_t metric, int32_t elem_type, const void* lhs, int32_t lhs_is_scalar, const void* rhs, int32_t rhs_is_scalar,
                                int64_t n, int32_t dim, void* out, void* stream);
int32_t dbhip_vec_distance(int32_t metric, const float* base, int64_t n, int32_t dim,
                           const float* queries, int32_t nq, float* out, void* stream);
/* ORDER BY distance LIMIT k (sort_compare.rs:197-209): per query the k smallest
 * distances, ties by lower row id; never materialises the n*nq matrix. */
int32_t dbhip_vec_topk(int32_t metric, const float* base, int64_t n, int32_t dim,
                       const float* queries, int32_t nq, int32_t k,
                       uint32_t* out_idx, float* out_dist, void* stream);
/* Merge of per-shard top-k lists (the final step of a row-range sharded search): every
 * query row of `dists`/`ids` holds m candidates (id 0xFFFFFFFF = empty slot); writes the k best,
 * ascending distance, ties by lower id, NaN last. */
int32_t dbhip_vec_topk_merge(const float* dists, const uint32_t* ids, int64_t m, int32_t nq, int32_t k,
                             uint32_t* out_idx, float* out_dist, void* stream);
/* Vector index. Replaces HNSWIndex::{build, search} (src/query/storages/common/index/src/hnsw_index/
 * hnsw.rs:62-315: HNSW graph over u8-quantised vectors, ef = 4k, approximate) at the same call sites
 * (one index per block of vectors, searched with a query batch and k). The device index is EXACT
 * (recall 1.0): a bf16 image of the column is scanned on v_mfma_f32_32x32x16_bf16 with a rigorous
 * error bound, so that rows are only ever over-selected, and the survivors are re-scored in f32 from
 * the original column — which is borrowed and must stay valid while the index lives.
 * Metrics: COSINE, DOT, L2 (l2 bounds ||q||^2 + ||b||^2 - 2 q.b from below and re-scores in the
 * reference's difference form, distance.rs:65-80); L1: DBHIP_ERR_UNSUPPORTED, use dbhip_vec_topk.
 * Results as dbhip_vec_topk. */
typedef struct dbhip_vec_index dbhip_vec_index;
int32_t dbhip_vec_index_build(int32_t metric, const float* base, int64_t n, int32_t dim,
                              dbhip_vec_index** out_host, void* stream);
int32_t dbhip_vec_index_search(dbhip_vec_index* ix, const float* queries, int32_t nq, int32_t k,
                               uint32_t* out_idx, float* out_dist, void* stream);
int32_t dbhip_vec_index_destroy(dbhip_vec_index* ix);
/* u8-quantised scoring (cpp/avx2.c:45,87 impl_score_dot_avx / impl_score_l1_avx). */
int32_t dbhip_score_u8(int32_t is_l1, const uint8_t* query, const uint8_t* base, int64_t n,
                       int32_t dim, float* out, void* stream);

/* ---- §8e: the communicator behind the boundary (RCCL over xGMI) ----------------------------------------------------------------
 * One process (rank) per GPU. Rank 0 calls dbhip_comm_unique_id and the host's own control plane ships the 128 bytes to the
 * other ranks; every rank then calls dbhip_comm_create(rank, world, id) on the thread whose current device is its GPU
 * (ncclCommInitRank; librccl.so is loaded with dlopen on first use — a single-GPU binding never loads it). world == 1 with
 * id == NULL gives a local communicator whose exchanges are copies (tests, single-GPU plans).
 *   dbhip_groupby_exchange_allgather  final merge for few groups (TPC-H Q1): the table as ONE fixed-size block of max_rows rows,
 *        ncclAllGather, merge of the other ranks' blocks — flush, collective and merge queued on one stream, no host round trip
 *        in between; afterwards EVERY rank holds the global result. DBHIP_ERR_CAPACITY (before any table is touched, on every
 *        rank alike) when some rank holds more than max_rows groups: take the all-to-all with a larger max_rows.
 *   dbhip_groupby_exchange_alltoall   hash-partitioned final merge (BASELINE configs[3]; the reference scatters partial states by
 *        hash % n into its Flight exchange, payload.rs:548-589 + aggregate_exchange_injector.rs:57-147): rows routed to bucket
 *        hash % world on the device, one grouped ncclSend / ncclRecv all-to-all of equal blocks (every xGMI link busy at once),
 *        the table rebuilt from the received blocks: rank r ends up owning the groups with hash % world == r, merged over all ranks.
 *   dbhip_comm_allgather / _alltoall / _allreduce_sum_u64  the plain collectives on device buffers (equal byte counts per rank /
 *        per peer): ANN shard top-k all-gather (dbhip_vec_topk_merge follows), result checks. */
typedef struct dbhip_comm dbhip_comm;
int32_t dbhip_comm_unique_id(uint8_t* out_id128_host);
int32_t dbhip_comm_create(int32_t rank, int32_t world, const uint8_t* id128_host, dbhip_comm** out_host);
/* An IN-PROCESS world for tests (no RCCL): `world` communicators created by `world` host threads of one process on one GPU under the
 * same group_id; their collectives are device-to-device copies made at a rendezvous. It lets the multi-rank protocols run with
 * world > 1 where only one GPU exists (tests/test_gpu_comm.py); every rank must call the same collective, from its own thread. */
int32_t dbhip_comm_create_loopback(uint64_t group_id, int32_t rank, int32_t world, dbhip_comm** out_host);
int32_t dbhip_comm_destroy(dbhip_comm* c);
/* A rank that gives up (its part of the plan failed, the query was cancelled) tells the others instead of leaving them waiting:
 * loopback — the group is marked failed, every rank waiting in a rendezvous and every later collective of the group returns
 * DBHIP_ERR_INVALID with the aborting rank's message (a failed dbhip_exchange_begin does this by itself; a rendezvous also gives up
 * after DBHIP_COMM_TIMEOUT_S seconds, default 600); RCCL — ncclCommAbort: the handle is marked aborted and every later collective on it returns DBHIP_ERR_INVALID (never a
 * silent single-rank copy); DBHIP_ERR_UNSUPPORTED when this librccl has no ncclCommAbort (the handle is still marked aborted on
 * this rank, peers blocked in a collective stay blocked). Under RCCL a FAILED dbhip_exchange_begin notifies nobody by itself: the
 * host calls dbhip_comm_abort on that rank (the peers' collectives then fail) — there is no timeout inside RCCL collectives. */
int32_t dbhip_comm_abort(dbhip_comm* c);
int32_t dbhip_comm_allgather(dbhip_comm* c, const void* send_dev, void* recv_dev, int64_t bytes_per_rank, void* stream);
int32_t dbhip_comm_alltoall(dbhip_comm* c, const void* send_dev, void* recv_dev, int64_t bytes_per_peer, void* stream);
int32_t dbhip_comm_allreduce_sum_u64(dbhip_comm* c, const uint64_t* send_dev, uint64_t* recv_dev, int64_t count, void* stream);
int32_t dbhip_groupby_exchange_allgather(dbhip_groupby* g, dbhip_comm* c, int64_t max_rows, void* stream);
int32_t dbhip_groupby_exchange_alltoall(dbhip_groupby* g, dbhip_comm* c, int64_t max_rows, void* stream);
/* The exchange of a BLOCK between all ranks, owned by the ABI (round 4; no torch.distributed, no per-column collective in the host):
 * the shuffle of the hash join (HashFlightScatter, flight_scatter_hash.rs:57-330: dest = dbhip_scatter_indices) and the exchange of the
 * distributed sort (sorts/sort_broadcast.rs:150-197 + the range exchange: dest = dbhip_sort_bound_partition) are the same three calls.
 *   dbhip_exchange_begin   DataBlock::scatter of `cols` by dest_index[i] < world (dbhip_scatter_columns: values, validities, Boolean,
 *                          String and Decimal columns) and ONE small all-to-all of what every rank sends every other (rows, and per String
 *                          column with data buffers the bytes of its long values) -> *out_recv_rows_host = rows this rank receives (the
 *                          host sizes the output columns from it). String columns WITH data buffers (round 5): the long (> 12 byte)
 *                          values of every destination's rows are packed back to back, the views re-based onto that piece — the form the
 *                          reference's exchange serializer ships (whole blocks, exchange/serde/exchange_serializer.rs);
 *                          dbhip_exchange_string_bytes then tells how many bytes arrive per column and dbhip_exchange_finish_strings
 *                          takes the buffers for them (plain dbhip_exchange_finish is for blocks whose strings are all inline)
 *   dbhip_exchange_finish  every column of every destination in ONE ncclGroup of send / recv pairs (one launch on the wire whatever
 *                          the width of the block; every xGMI link busy at once); out_data_host[c] = recv_rows elements in source-rank
 *                          order, rows of one source in their order; validity / Boolean Bitmaps arrive as per-source pieces and are
 *                          concatenated bit by bit into out_validity_host[c] / out_data_host[c] (8-byte aligned, ceil(rows / 64) * 8
 *                          bytes). out_src_starts_host[world + 1] (may be NULL): where every source rank's rows start.
 *   dbhip_exchange_destroy frees the scattered image (also after a failed finish).
 * A local communicator (world of one) makes the exchange a copy. dbhip_vec_topk_allgather is the merge of the row-range sharded vector
 * search: per-shard top-k ids (u32, 0xFFFFFFFF = empty, LOCAL row numbers) + distances [nq][k] -> ids made global
 * (+ row_offset), both all-gathered in one group, k-way merged (dbhip_vec_topk_merge) -> the global top-k on every rank. */
typedef struct dbhip_exchange dbhip_exchange;
int32_t dbhip_exchange_begin(dbhip_comm* c, const dbhip_col* cols, int32_t ncols, const uint32_t* dest_index, int64_t n, int64_t* out_recv_rows_host,
                             dbhip_exchange** out_host, void* stream);
/* The two distributed plans that sit on the exchange, as single calls: the destination of every row by the reference's own rule, then
 * dbhip_exchange_begin (finish / destroy as above).
 *   dbhip_shuffle_exchange_begin  hash shuffle (flight_scatter_hash.rs:57-330): destination = siphash64 of the key columns % world
 *                                 (dbhip_scatter_indices: bit-exact on the reference's golden values, so GPU and CPU nodes agree);
 *                                 `cols` = the block to move (usually including the keys). Both sides of a shuffle join call it.
 *   dbhip_sort_exchange_begin     range partition of the distributed sort (sort_spill.rs:740-1040, sort_exchange_injector.rs):
 *                                 partition = number of bounds that sort strictly before the row (dbhip_sort_bound_partition),
 *                                 sent to rank partition % world; the receiver sorts what arrives (dbhip_sort_perm). */
int32_t dbhip_shuffle_exchange_begin(dbhip_comm* c, const dbhip_col* keys, int32_t nkeys, const dbhip_col* cols, int32_t ncols, int64_t n,
                                     int64_t* out_recv_rows_host, dbhip_exchange** out_host, void* stream);
int32_t dbhip_sort_exchange_begin(dbhip_comm* c, const dbhip_col* keys, const dbhip_col* bounds, const uint8_t* desc_host,
                                  const uint8_t* nulls_first_host, int32_t nkeys, int64_t nbounds, const dbhip_col* cols, int32_t ncols,
                                  int64_t n, int64_t* out_recv_rows_host, dbhip_exchange** out_host, void* stream);
int32_t dbhip_exchange_finish(dbhip_exchange* x, void* const* out_data_host, uint8_t* const* out_validity_host, int64_t* out_src_starts_host,
                              void* stream);
/* Long strings (round 5). out_bytes_host[c] = bytes of long-string data column c receives (0 for other columns). finish_strings =
 * finish with out_string_bytes_host[c]: a device buffer of at least that many bytes per such column — it becomes buffer 0 of the received
 * String column (the received views are {len, prefix, 0, offset into it}). */
int32_t dbhip_exchange_string_bytes(dbhip_exchange* x, int64_t* out_bytes_host);
int32_t dbhip_exchange_finish_strings(dbhip_exchange* x, void* const* out_data_host, uint8_t* const* out_validity_host,
                                      uint8_t* const* out_string_bytes_host, int64_t* out_src_starts_host, void* stream);
int32_t dbhip_exchange_destroy(dbhip_exchange* x);
int32_t dbhip_vec_topk_allgather(dbhip_comm* c, const uint32_t* idx_dev, const float* dist_dev, int32_t nq, int32_t k, uint64_t row_offset,
                                 uint32_t* out_idx_dev, float* out_dist_dev, void* stream);

/* ---- §8f-4: vector-cluster KMeans and the f32 VectorDistanceKernel ------------------------------------------------------------
 * Replaces KMeans::compute (src/query/storages/common/index/src/kmeans.rs:93-291: kmeans++ initialisation with the fixed LCG seed,
 * Lloyd iterations until nothing changes or the centroid shift is <= 1e-4, at most 100) behind TransformVectorCluster
 * (fuse/src/operations/common/processors/transform_vector_cluster.rs: batches of <= 262,144 rows, <= 64 clusters), and
 * VectorDistanceKernel::{dot, l2_squared, l1} (vector.rs:45-260). The reference is deterministic — fixed seed, fixed summation
 * orders (its production kernel: Avx, 8 fused lanes + tail) — and so is this: assignments and distances are BIT-IDENTICAL to
 * the CPU path on x86_64 with avx2 + fma.
 *   distance_type       0 = L1, 1 = L2, 2 = Dot (VectorDistanceType); `data` = rows x dim f32, row-major, on the device
 *   normalize_input     != 0: every row is normalised first (vector_samples does this for Dot before KMeans::compute)
 *   out_assignments     u32[rows] cluster ids, out_distances f32[rows] distance to the own centroid (build_result), both on the
 *                       device; *out_k_host = ceil(rows / rows_per_cluster) clamped to [1, rows]; *out_iterations_host
 * dbhip_vec_kernel_f32: out[i] = kernel(a[i], b[i]) over n pairs of dim-vectors; which 0 = dot, 1 = l2_squared, 2 = l1. */
int32_t dbhip_kmeans(int32_t distance_type, const float* data, int64_t rows, int32_t dim, int64_t rows_per_cluster, int32_t normalize_input,
                     uint32_t* out_assignments, float* out_distances, int64_t* out_k_host, int32_t* out_iterations_host, void* stream);
int32_t dbhip_vec_kernel_f32(int32_t which, const float* a, const float* b, int64_t n, int32_t dim, float* out, void* stream);

/* ---- Scan side: one Parquet column chunk -> one device-resident column --------
 * Replaces, per column, column_chunks_to_record_batch + the arrow -> Column conversion
 * (src/query/storages/fuse/src/io/read/block/parquet/deserialize.rs:33-81;
 * block_reader_parquet_deserialize.rs): `DataItem::RawData(bytes)` of one column chunk in, one Column
 * out. Covers what the reference's writer emits (storages/common/blocks/src/parquet_rs.rs:91-160: one
 * row group; DATA_PAGE v1 + PLAIN, or DATA_PAGE_V2 + RLE_DICTIONARY with PLAIN fallback pages; RLE
 * definition levels) for flat columns (max_rep_level 0, max_def_level <= 1), TableCompression
 * None / Zstd (the default) / LZ4 / Snappy (table_compression.rs:38-58). Everything else (nested
 * columns, DELTA_* encodings, other codecs): DBHIP_ERR_UNSUPPORTED — the binding keeps arrow-rs for that
 * chunk. Pages of compressed chunks are decompressed on the HOST inside open() (libzstd / liblz4 of the
 * system, Snappy decoded in place — the reference decompresses on the CPU as well); what the device
 * decodes is the decompressed page stream, the IMAGE (dbhip_pq_chunk_image): for such chunks the caller
 * uploads the image instead of the chunk, and String views point into it.
 *   open   (host)   parses the page headers and the run headers of the hybrid streams from the HOST
 *                   copy of the chunk and plans the decode; nothing touches the device.
 *   decode (device) expands the plan from the DEVICE copy of the same bytes (the caller uploads or
 *                   DMA-s the chunk; it may decode repeatedly). out_values: info.out_bytes bytes
 *                   (rows x element; DBHIP_T_BOOL: LSB-first bitmap); out_validity (nullable columns):
 *                   info.validity_bytes bytes, LSB-first, 1 = valid; NULL rows decode to 0.
 *                   DBHIP_T_STRING: 16-byte views whose long form points INTO chunk_dev — the chunk is
 *                   buffer 0 of the resulting column and must stay resident while the column lives.
 * physical_type / codec are parquet.thrift's Type / CompressionCodec numbers (BOOLEAN 0, INT32 1,
 * INT64 2, FLOAT 4, DOUBLE 5, BYTE_ARRAY 6, FIXED_LEN_BYTE_ARRAY 7; UNCOMPRESSED 0, SNAPPY 1, ZSTD 6, LZ4_RAW 7). out_type: INT32 ->
 * I8/I16/I32/U8/U16/U32/DATE/I64/DEC64; INT64 -> I64/U64/TIMESTAMP/DEC64/DEC128; FLOAT/DOUBLE -> F32/F64;
 * BYTE_ARRAY -> STRING; FIXED_LEN_BYTE_ARRAY(n <= 16, big-endian decimal) -> DEC128 (n <= 8: DEC64). */
typedef struct dbhip_pq_chunk dbhip_pq_chunk;
typedef struct dbhip_pq_info {
  int64_t num_values;     /* rows of the chunk (NULLs included) */
  int64_t num_nulls;
  int32_t out_type;
  int32_t has_validity;   /* max_def_level == 1 */
  int64_t out_bytes;
  int64_t validity_bytes;
  int64_t n_pages;        /* data pages */
  int64_t n_dict_values;
  int64_t image_bytes;    /* compressed chunks: bytes of the decompressed image decode() reads (0: the chunk itself) */
} dbhip_pq_info;
int32_t dbhip_pq_chunk_open(const uint8_t* chunk_host, int64_t chunk_len, int32_t codec,
                            int32_t physical_type, int32_t type_length, int32_t max_def_level,
                            int32_t max_rep_level, int32_t out_type, dbhip_pq_chunk** out_host,
                            dbhip_pq_info* info_host);
/* Nullable columns: the validity bitmap open() decoded from the definition levels (host memory, LSB first, 1 = valid,
 * ceil(rows / 64) * 8 bytes; NULL / 0 for required columns). decode() writes the same bits to out_validity_dev. */
int32_t dbhip_pq_chunk_validity(dbhip_pq_chunk* c, const uint8_t** out_ptr_host, int64_t* out_bytes_host);
/* Host pointer / size of the decompressed image (NULL / 0 for UNCOMPRESSED chunks: upload the chunk). Owned by
 * the handle, valid until close. */
int32_t dbhip_pq_chunk_image(dbhip_pq_chunk* c, const uint8_t** out_ptr_host, int64_t* out_len_host);
/* chunk_dev: device copy of the chunk (UNCOMPRESSED) or of the image (compressed chunks). */
int32_t dbhip_pq_chunk_decode(dbhip_pq_chunk* c, const uint8_t* chunk_dev, void* out_values_dev,
                              uint8_t* out_validity_dev, void* stream);
/* DEVICE mode of the same boundary: the page payload never passes through the host. open_device reads the thrift page headers
 * only (sizes, value counts, encodings: a few dozen bytes per page); decode_device decompresses the pages (one wave per page —
 * ZSTD, the reference's default TableCompression: Huffman literals + FSE sequences walked on the GPU, frames with a dictionary
 * id are DBHIP_ERR_UNSUPPORTED; SNAPPY / LZ4_RAW), walks the run headers of the RLE / bit-packed hybrid streams (definition
 * levels, dictionary indices, RLE booleans), the length prefixes of PLAIN BYTE_ARRAY pages and DELTA_BINARY_PACKED blocks
 * (INT32 / INT64) on the GPU, from the HBM copy of the chunk AS STORED. Same type pairs, same output layout as open / decode.
 *   chunk_dev    the chunk as stored (the bytes given to open_device), 16-byte aligned, readable up to the next 16-byte
 *                boundary past its end
 *   image_dev    compressed chunks: info.image_bytes bytes, 16-byte aligned, caller-owned — receives the decompressed pages;
 *                DBHIP_T_STRING views point into it (it is buffer 0 of the column; UNCOMPRESSED chunks: chunk_dev is, pass NULL)
 *   info.num_nulls is -1 when the page headers do not tell (v1 pages of a nullable column); out_nulls_host (may be NULL) gets
 *   the count. Nothing about the payload is validated on the host, so the device checks every access against the page, the
 *   dictionary and the output; decode_device synchronises `stream` and returns DBHIP_ERR_INVALID for a chunk that fails a check. */
int32_t dbhip_pq_chunk_open_device(const uint8_t* chunk_host, int64_t chunk_len, int32_t codec,
                                   int32_t physical_type, int32_t type_length, int32_t max_def_level,
                                   int32_t max_rep_level, int32_t out_type, dbhip_pq_chunk** out_host,
                                   dbhip_pq_info* info_host);
int32_t dbhip_pq_chunk_decode_device(dbhip_pq_chunk* c, const uint8_t* chunk_dev, uint8_t* image_dev,
                                     void* out_values_dev, uint8_t* out_validity_dev, int64_t* out_nulls_host,
                                     void* stream);
/* MANY chunks, one launch set (what a scan does: the column chunks of a block — or of several blocks — together): the pages of all
 * chunks are decompressed by ONE launch per codec family (ZSTD; SNAPPY + LZ4_RAW), their levels / dictionaries / values by one
 * launch each over all data pages, and the verdicts and null counts come back in one read-back. Arrays of n_chunks entries, as the
 * arguments of dbhip_pq_chunk_decode_device (image_dev[i] / out_validity_dev[i] NULL where that call takes NULL);
 * out_nulls_host / out_status_host may be NULL. A chunk that fails its device checks gets its own status in out_status_host[i]
 * (the others are decoded); the call returns the first such status. Handles must be distinct. */
int32_t dbhip_pq_chunks_decode_device(dbhip_pq_chunk* const* chunks, int32_t n_chunks, const uint8_t* const* chunk_dev,
                                      uint8_t* const* image_dev, void* const* out_values_dev,
                                      uint8_t* const* out_validity_dev, int64_t* out_nulls_host,
                                      int32_t* out_status_host, void* stream);
/* List<primitive> columns (one repeated ancestor: max_rep_level 1, max_def_level = list_nullable + 1 + element_nullable — the
 * three-level LIST of parquet's LogicalTypes.md that the reference writes for Array(T), read there through arrow-rs,
 * storages/common/.../deserialize.rs:33-81). open_device_list reads the page headers like open_device; info.num_values is the
 * number of LEVEL ENTRIES, the bound of both the rows and the elements: size offsets for (num_values + 1) u64, the element values
 * for info.out_bytes and each validity for info.validity_bytes. decode_device_list decodes the repetition / definition levels and
 * the leaf values on the device: out_offsets[r] .. out_offsets[r + 1] are row r's elements (Databend's ArrayColumn offsets),
 * out_list_validity (list_nullable; a NULL list is an empty run), out_values the elements back to back in the output type (a NULL
 * element decodes to 0), out_elem_validity (element_nullable); the row / element / NULL-list counts come back to the host. String
 * elements are views into chunk_dev / the image like a flat String chunk's. Deeper nesting (List<List<..>>, Map, Tuple members):
 * DBHIP_ERR_UNSUPPORTED from the flat open — the binding keeps arrow-rs. */
int32_t dbhip_pq_chunk_open_device_list(const uint8_t* chunk_host, int64_t chunk_len, int32_t codec, int32_t physical_type, int32_t type_length,
                                        int32_t list_nullable, int32_t element_nullable, int32_t out_type, dbhip_pq_chunk** out_host,
                                        dbhip_pq_info* info_host);
int32_t dbhip_pq_chunk_decode_device_list(dbhip_pq_chunk* c, const uint8_t* chunk_dev, uint8_t* image_dev, uint64_t* out_offsets_dev,
                                          uint8_t* out_list_validity_dev, void* out_values_dev, uint8_t* out_elem_validity_dev,
                                          int64_t* out_rows_host, int64_t* out_elems_host, int64_t* out_null_lists_host, void* stream);
int32_t dbhip_pq_chunk_close(dbhip_pq_chunk* c);

/* ---------------------------------------------------------------------------------------------------------------------
 * HNSW vector index with u8 scalar quantisation — the reference's indexed ANN path
 *   HNSWIndex::{build, open, search, generate_scores}        hnsw_index/hnsw.rs:62-315
 *   EncodedVectorsU8::{encode, encode_query, score_point}     hnsw_index/quantization/encoded_vectors_u8.rs:54-215,301-413
 *   GraphLayersBuilder::link_new_point (+ heuristic)          hnsw_index/graph_layers_builder.rs:343-520
 *   GraphLayers::{search_entry, search_on_level, search}      hnsw_index/graph_layers.rs:72-247
 * `distance`: DBHIP_VEC_COSINE (vectors and queries normalised as cosine_preprocess does, score = dot), DBHIP_VEC_L1,
 * DBHIP_VEC_L2 — the three the reference's index option accepts. Vectors: dense row-major f32 [n][dim] on the device.
 * build: levels drawn like get_random_layer (round(-ln(u) / ln(max(m, 2)))) from a generator seeded with `seed` (the
 *   reference uses thread_rng()); m0 = 2 m; the first 256 points are linked one after the other, the rest concurrently (one
 *   wave per point; the reference: one rayon task per point), scoring the ORIGINAL vectors as the reference's build does.
 *   The graph is therefore not reproducible bit for bit in either implementation; the quantiser and the search are.
 * from_graph: the index over a GIVEN graph (HNSWIndex::open): `levels_host[n]`, lists in point-major, level-minor order —
 *   `nlinks_host[list]` entries each, concatenated in `links_host`.
 * search: per query the `limit` nearest by quantised score, ef = 4 * limit (hnsw.rs:108-110), distances post-processed
 *   (cosine |1 - s|, l1 |s|, l2 sqrt|s|, hnsw.rs:317-343). `queries_dev` are raw (preprocess_query is applied inside).
 *   out_ids / out_dist: [nq][limit], missing results (fewer than `limit` reachable points) = 0xFFFFFFFF / NaN.
 *   limit <= 64. Ties between equal scores resolve as std::collections::BinaryHeap resolves them in the reference.
 * scores: generate_scores — the post-processed quantised distance of every row to every query, out [nq][n].
 * encoded: the reference's storage layout of the quantised vectors (per vector: f32 offset, then actual_dim codes,
 *   actual_dim = dim rounded up to 16) into `out_dev` [n][4 + actual_dim]; meta: alpha, offset, multiplier, actual_dim. */
typedef struct dbhip_hnsw dbhip_hnsw;
int32_t dbhip_hnsw_build(const float* vectors_dev, int64_t n, int32_t dim, int32_t distance, int32_t m, int32_t ef_construct,
                         uint64_t seed, dbhip_hnsw** out, void* stream);
/* The DETERMINISTIC build: `levels_host[n]` are given (the reference draws them from thread_rng()), ONE wave links the points
 * one after the other in row order — HNSWIndex::build's insertion loop run sequentially (hnsw.rs:158-235) — and the build
 * scorer sums in the reference's order (calculate_score, point_scorer.rs:133-174: a sequential f32 fold over the
 * pre-processed column). The graph equals the sequential CPU restatement (oracle/hnsw_oracle.c orc_hnsw_build) link for
 * link (tests/test_gpu_hnsw.py); it is the mode that pins the builder, not the fast one. */
int32_t dbhip_hnsw_build_sequential(const float* vectors_dev, int64_t n, int32_t dim, int32_t distance, int32_t m, int32_t ef_construct,
                                    const int32_t* levels_host, dbhip_hnsw** out, void* stream);
int32_t dbhip_hnsw_from_graph(const float* vectors_dev, int64_t n, int32_t dim, int32_t distance, int32_t m,
                              const int32_t* levels_host, const uint32_t* links_host, const int32_t* nlinks_host,
                              uint32_t entry_point, int32_t entry_level, dbhip_hnsw** out, void* stream);
/* HNSWIndex::open (hnsw.rs:62-98) over the STORED form of an index: `encoded_dev` = the `encoded_u8_data` column's bytes (per vector
 * an f32 offset and actual_dim codes, quantization/encoded_storage.rs) with alpha / offset / multiplier from the `encoded_u8_meta`
 * JSON (encoded_vectors_u8.rs:45-52), and the graph as dbhip_hnsw_from_graph takes it. The host-side reader / writer of the four
 * binary columns (graph_links Compressed format, bincode GraphLayerData, JSON metadata) is databend_amd/hnsw_format.py. No
 * original vectors are needed (none are stored): the index only searches. */
int32_t dbhip_hnsw_open(const uint8_t* encoded_dev, float alpha, float offset, float multiplier, int64_t n, int32_t dim, int32_t distance,
                        int32_t m, const int32_t* levels_host, const uint32_t* links_host, const int32_t* nlinks_host, uint32_t entry_point,
                        int32_t entry_level, dbhip_hnsw** out, void* stream);
int32_t dbhip_hnsw_export_graph(dbhip_hnsw* h, int32_t* levels_host, uint32_t* links_host, int32_t* nlinks_host,
                                int64_t* out_n_lists_host, uint32_t* out_entry_point_host, int32_t* out_entry_level_host,
                                void* stream);
int32_t dbhip_hnsw_search(dbhip_hnsw* h, const float* queries_dev, int32_t nq, int32_t limit, uint32_t* out_ids_dev,
                          float* out_dist_dev, void* stream);
int32_t dbhip_hnsw_scores(dbhip_hnsw* h, const float* queries_dev, int32_t nq, float* out_dev, void* stream);
int32_t dbhip_hnsw_encoded(dbhip_hnsw* h, void* out_dev, void* stream);
int32_t dbhip_hnsw_meta(dbhip_hnsw* h, float* alpha_host, float* offset_host, float* multiplier_host, int32_t* actual_dim_host);
int32_t dbhip_hnsw_destroy(dbhip_hnsw* h);

/* ---- diagnostics and test hooks (exported, not part of the drop-in surface) -------------------------------------------------
 * The library exports exactly the functions this header declares (csrc/Makefile builds its export list from it). */
/* restrict the probe hash of an EMPTY table to `mask`, so that distinct keys share a hash word and the collision path runs (the
 * reference tests the same with hand-made tags, hash_index/index.rs:385-404); the binary join's twin is process-wide */
int32_t dbhip_groupby_debug_set_hash_mask(dbhip_groupby* g, uint64_t mask);
int32_t dbhip_join_binary_debug_set_hash_mask(uint64_t mask);
/* force the radix-partitioned path with 2^bits partitions (0: back to adaptive, < 0: never partition) */
int32_t dbhip_groupby_debug_set_partition_bits(dbhip_groupby* g, int32_t bits);
/* keep a table off (0) / on (1, default) the compact-row kernels of the partitioned and LDS aggregation paths (both kernel families
 * are driven through the same parity cases) */
int32_t dbhip_groupby_debug_set_compact(dbhip_groupby* g, int32_t on);
/* launches of the fused-aggregation kernel since the library was loaded: out3_host = {run-time specialised, interpreted, refused
 * because the specialised kernel was still being compiled} — how a bench or a test tells which kernel a call went through */
int32_t dbhip_fagg_stats(uint64_t* out3_host);
/* internal scratch held right now: out2_host = {(thread, stream) entries, bytes} */
int32_t dbhip_scratch_stats(uint64_t* out2_host);
/* Offline compile checks of the run-time specialisation (need no device; the CPU test-suite runs them): the code object of a small
 * fixed query shape / of a table layout + program -> its size in bytes, -1 with the compiler's log in log_out_host, -2 when the shape
 * is outside the fused kernel. */
int64_t dbhip_jit_compile_check(char* log_out_host, int64_t log_cap);
int64_t dbhip_jit_offline(const int32_t* key_types_host, const uint8_t* key_nullable_host, int32_t nkeys, const dbhip_agg_desc* aggs_host,
                          int32_t naggs, const dbhip_col* keys, const dbhip_agg_program* prog, int32_t slots, char* code_out_host,
                          int64_t code_cap, char* log_out_host, int64_t log_cap);

#ifdef __cplusplus
}
#endif
#endif /* DBHIP_H */

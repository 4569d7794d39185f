// dbhip_sys.rs — GENERATED from include/dbhip.h by tools/gen_rust_bindings.py (do not edit; `--check` in tests/test_abi.py).
// The raw FFI surface of libdbhip.so for a Rust host: link with `cargo:rustc-link-lib=dylib=dbhip`. Safe wrappers that
// implement Databend's Function / Processor traits over these calls are sketched in INTEGRATION.md.
#![allow(non_camel_case_types, non_upper_case_globals, dead_code)]
use std::os::raw::{c_char, c_int, c_void};

pub const DBHIP_OK: i32 = 0;
pub const DBHIP_ERR_INVALID: i32 = 1;
pub const DBHIP_ERR_HIP: i32 = 2;
pub const DBHIP_ERR_NO_DEVICE: i32 = 3;
pub const DBHIP_ERR_ROW_ERRORS: i32 = 4;
pub const DBHIP_ERR_OVERFLOW: i32 = 5;
pub const DBHIP_ERR_CAPACITY: i32 = 6;
pub const DBHIP_ERR_UNSUPPORTED: i32 = 7;
pub const DBHIP_ERR_CANCELLED: i32 = 8;
pub const DBHIP_T_BOOL: i32 = 1;   // dbhip_type
pub const DBHIP_T_I8: i32 = 2;   // dbhip_type
pub const DBHIP_T_I16: i32 = 3;   // dbhip_type
pub const DBHIP_T_I32: i32 = 4;   // dbhip_type
pub const DBHIP_T_I64: i32 = 5;   // dbhip_type
pub const DBHIP_T_U8: i32 = 6;   // dbhip_type
pub const DBHIP_T_U16: i32 = 7;   // dbhip_type
pub const DBHIP_T_U32: i32 = 8;   // dbhip_type
pub const DBHIP_T_U64: i32 = 9;   // dbhip_type
pub const DBHIP_T_F32: i32 = 10;   // dbhip_type
pub const DBHIP_T_F64: i32 = 11;   // dbhip_type
pub const DBHIP_T_DATE: i32 = 12;   // dbhip_type
pub const DBHIP_T_TIMESTAMP: i32 = 13;   // dbhip_type
pub const DBHIP_T_DEC64: i32 = 14;   // dbhip_type
pub const DBHIP_T_DEC128: i32 = 15;   // dbhip_type
pub const DBHIP_T_STRING: i32 = 16;   // dbhip_type
pub const DBHIP_T_DEC256: i32 = 17;   // dbhip_type
pub const DBHIP_OP_PLUS: i32 = 0;   // dbhip_arith_op
pub const DBHIP_OP_MINUS: i32 = 1;   // dbhip_arith_op
pub const DBHIP_OP_MULTIPLY: i32 = 2;   // dbhip_arith_op
pub const DBHIP_OP_DIVIDE: i32 = 3;   // dbhip_arith_op
pub const DBHIP_OP_INTDIV: i32 = 4;   // dbhip_arith_op
pub const DBHIP_OP_MODULO: i32 = 5;   // dbhip_arith_op
pub const DBHIP_OP_DIV0: i32 = 6;   // dbhip_arith_op
pub const DBHIP_OP_DIVNULL: i32 = 7;   // dbhip_arith_op
pub const DBHIP_CMP_EQ: i32 = 0;   // dbhip_cmp_op
pub const DBHIP_CMP_NOTEQ: i32 = 1;   // dbhip_cmp_op
pub const DBHIP_CMP_LT: i32 = 2;   // dbhip_cmp_op
pub const DBHIP_CMP_LTE: i32 = 3;   // dbhip_cmp_op
pub const DBHIP_CMP_GT: i32 = 4;   // dbhip_cmp_op
pub const DBHIP_CMP_GTE: i32 = 5;   // dbhip_cmp_op
pub const DBHIP_EX_LOAD: i32 = 0;   // dbhip_expr_op
pub const DBHIP_EX_CONST: i32 = 1;   // dbhip_expr_op
pub const DBHIP_EX_PLUS: i32 = 2;   // dbhip_expr_op
pub const DBHIP_EX_MINUS: i32 = 3;   // dbhip_expr_op
pub const DBHIP_EX_MULTIPLY: i32 = 4;   // dbhip_expr_op
pub const DBHIP_EX_DIVIDE: i32 = 5;   // dbhip_expr_op
pub const DBHIP_EX_EQ: i32 = 6;   // dbhip_expr_op
pub const DBHIP_EX_NOTEQ: i32 = 7;   // dbhip_expr_op
pub const DBHIP_EX_LT: i32 = 8;   // dbhip_expr_op
pub const DBHIP_EX_LTE: i32 = 9;   // dbhip_expr_op
pub const DBHIP_EX_GT: i32 = 10;   // dbhip_expr_op
pub const DBHIP_EX_GTE: i32 = 11;   // dbhip_expr_op
pub const DBHIP_EX_AND: i32 = 12;   // dbhip_expr_op
pub const DBHIP_EX_OR: i32 = 13;   // dbhip_expr_op
pub const DBHIP_EX_NOT: i32 = 14;   // dbhip_expr_op
pub const DBHIP_EX_CAST: i32 = 15;   // dbhip_expr_op
pub const DBHIP_EX_IF: i32 = 16;   // dbhip_expr_op
pub const DBHIP_EX_IS_TRUE: i32 = 17;   // dbhip_expr_op
pub const DBHIP_AGG_COUNT: i32 = 0;   // dbhip_agg_kind
pub const DBHIP_AGG_SUM: i32 = 1;   // dbhip_agg_kind
pub const DBHIP_AGG_MIN: i32 = 2;   // dbhip_agg_kind
pub const DBHIP_AGG_MAX: i32 = 3;   // dbhip_agg_kind
pub const DBHIP_VEC_COSINE: i32 = 0;   // dbhip_vec_metric
pub const DBHIP_VEC_L2: i32 = 1;   // dbhip_vec_metric
pub const DBHIP_VEC_DOT: i32 = 2;   // dbhip_vec_metric
pub const DBHIP_VEC_L1: i32 = 3;   // dbhip_vec_metric
pub const DBHIP_VEC_NORM: i32 = 4;   // dbhip_vec_metric
pub const DBHIP_ABI_VERSION: i32 = 6;

#[repr(C)]
pub struct dbhip_groupby { _private: [u8; 0] }
#[repr(C)]
pub struct dbhip_join { _private: [u8; 0] }
#[repr(C)]
pub struct dbhip_join_binary { _private: [u8; 0] }
#[repr(C)]
pub struct dbhip_vec_index { _private: [u8; 0] }
#[repr(C)]
pub struct dbhip_comm { _private: [u8; 0] }
#[repr(C)]
pub struct dbhip_exchange { _private: [u8; 0] }
#[repr(C)]
pub struct dbhip_pq_chunk { _private: [u8; 0] }
#[repr(C)]
pub struct dbhip_hnsw { _private: [u8; 0] }

#[repr(C)]
#[derive(Clone, Copy)]
pub struct dbhip_col {
    pub r#type: i32,
    pub is_scalar: i32,
    pub data: *const c_void,
    pub validity: *const u8,
    pub validity_offset: i64,
    pub buffers: *const *const c_void,
    pub n_buffers: i32,
    pub precision: u8,
    pub scale: u8,
    pub _pad: [u8; 2],
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct dbhip_expr_ins {
    pub op: i32,
    pub dst: i32,
    pub a: i32,
    pub b: i32,
    pub r#type: i32,
    pub precision: u8,
    pub scale: u8,
    pub _pad: [u8; 2],
    pub imm: u64,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct dbhip_agg_desc {
    pub kind: i32,
    pub arg_type: i32,
    pub arg_precision: u8,
    pub arg_scale: u8,
    pub arg_nullable: u8,
    pub _pad: u8,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct dbhip_agg_program {
    pub prog: *const dbhip_expr_ins,
    pub n_ins: i32,
    pub inputs: *const dbhip_col,
    pub n_inputs: i32,
    pub filter_reg: i32,
    pub arg_regs: *const i32,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct dbhip_pq_info {
    pub num_values: i64,
    pub num_nulls: i64,
    pub out_type: i32,
    pub has_validity: i32,
    pub out_bytes: i64,
    pub validity_bytes: i64,
    pub n_pages: i64,
    pub n_dict_values: i64,
    pub image_bytes: i64,
}

extern "C" {
    pub fn dbhip_abi_version() -> i32;
    pub fn dbhip_init(device: i32) -> i32;
    pub fn dbhip_device_count(out_count_host: *mut i32) -> i32;
    pub fn dbhip_last_error() -> *const c_char;
    pub fn dbhip_alloc(bytes: usize, out_dev_ptr_host: *mut *mut c_void) -> i32;
    pub fn dbhip_free(dev_ptr: *mut c_void) -> i32;
    pub fn dbhip_trim() -> i32;
    pub fn dbhip_memcpy_h2d(dst_dev: *mut c_void, src_host: *const c_void, bytes: usize, stream: *mut c_void) -> i32;
    pub fn dbhip_memcpy_d2h(dst_host: *mut c_void, src_dev: *const c_void, bytes: usize, stream: *mut c_void) -> i32;
    pub fn dbhip_memcpy_d2d(dst_dev: *mut c_void, src_dev: *const c_void, bytes: usize, stream: *mut c_void) -> i32;
    pub fn dbhip_memset(dst_dev: *mut c_void, byte: i32, bytes: usize, stream: *mut c_void) -> i32;
    pub fn dbhip_stream_create(out_stream_host: *mut *mut c_void) -> i32;
    pub fn dbhip_stream_destroy(stream: *mut c_void) -> i32;
    pub fn dbhip_stream_release_scratch(stream: *mut c_void) -> i32;
    pub fn dbhip_stream_sync(stream: *mut c_void) -> i32;
    pub fn dbhip_stream_cancel(stream: *mut c_void) -> i32;
    pub fn dbhip_stream_cancel_clear(stream: *mut c_void) -> i32;
    pub fn dbhip_event_create(out_event_host: *mut *mut c_void) -> i32;
    pub fn dbhip_event_record(event: *mut c_void, stream: *mut c_void) -> i32;
    pub fn dbhip_event_elapsed_ms(start: *mut c_void, stop: *mut c_void, out_ms_host: *mut f32) -> i32;
    pub fn dbhip_event_destroy(event: *mut c_void) -> i32;
    pub fn dbhip_last_kernel_ms(out_ms_host: *mut f32) -> i32;
    pub fn dbhip_arith(op: i32, lhs: *const dbhip_col, rhs: *const dbhip_col, n: i64, out_type: i32, out: *mut c_void, err_bitmap: *mut u8, err_count_dev: *mut u64, stream: *mut c_void) -> i32;
    pub fn dbhip_cast(src: *const dbhip_col, dst_type: i32, is_try: i32, rounding_mode: i32, n: i64, out: *mut c_void, bitmap: *mut u8, err_count_dev: *mut u64, stream: *mut c_void) -> i32;
    pub fn dbhip_arith_result_type(op: i32, lhs_type: i32, rhs_type: i32) -> i32;
    pub fn dbhip_sum_a_plus_b_mul_c_i64(a: *const i64, b: *const i64, c: *const i64, n: i64, out_sum_dev: *mut i64, stream: *mut c_void) -> i32;
    pub fn dbhip_sum(col: *const dbhip_col, n: i64, out_sum_dev: *mut c_void, stream: *mut c_void) -> i32;
    pub fn dbhip_expr_eval(prog_host: *const dbhip_expr_ins, n_ins: i32, inputs_host: *const dbhip_col, n_inputs: i32, n: i64, out_reg: i32, out_values: *mut c_void, out_validity: *mut u8, err_bitmap: *mut u8, err_count_dev: *mut u64, sum_out_dev: *mut c_void, stream: *mut c_void) -> i32;
    pub fn dbhip_decimal_result_size(op: i32, lp: u8, ls: u8, rp: u8, rs: u8, out_precision_host: *mut u8, out_scale_host: *mut u8) -> i32;
    pub fn dbhip_decimal_arith(op: i32, lhs: *const dbhip_col, rhs: *const dbhip_col, n: i64, out_type: i32, out_precision: u8, out_scale: u8, out: *mut c_void, err_bitmap: *mut u8, err_count_dev: *mut u64, stream: *mut c_void) -> i32;
    pub fn dbhip_decimal_neg(src: *const dbhip_col, n: i64, out: *mut c_void, stream: *mut c_void) -> i32;
    pub fn dbhip_decimal_cast(src: *const dbhip_col, dst_type: i32, dst_precision: u8, dst_scale: u8, is_try: i32, rounding_mode: i32, n: i64, out: *mut c_void, bitmap: *mut u8, err_count_dev: *mut u64, stream: *mut c_void) -> i32;
    pub fn dbhip_cmp(op: i32, lhs: *const dbhip_col, rhs: *const dbhip_col, n: i64, out_bitmap: *mut u8, stream: *mut c_void) -> i32;
    pub fn dbhip_bitmap_binary(is_or: i32, a: *const u8, b: *const u8, n: i64, out: *mut u8, stream: *mut c_void) -> i32;
    pub fn dbhip_bitmap_count(bitmap: *const u8, bit_offset: i64, n: i64, out_count_dev: *mut u64, stream: *mut c_void) -> i32;
    pub fn dbhip_bitmap_set_indices(idx: *const u32, n_idx: i64, bitmap: *mut u8, nbits: i64, stream: *mut c_void) -> i32;
    pub fn dbhip_filter_select(bitmap: *const u8, bit_offset: i64, n: i64, out_sel: *mut u32, out_count_dev: *mut u64, stream: *mut c_void) -> i32;
    pub fn dbhip_select_cmp(op: i32, lhs: *const dbhip_col, rhs: *const dbhip_col, sel_in: *const u32, n: i64, out_true: *mut u32, out_false: *mut u32, out_count_true_dev: *mut u64, stream: *mut c_void) -> i32;
    pub fn dbhip_select_bool(predicate: *const dbhip_col, sel_in: *const u32, n: i64, out_true: *mut u32, out_false: *mut u32, out_count_true_dev: *mut u64, stream: *mut c_void) -> i32;
    pub fn dbhip_take(src: *const c_void, elem_size: i32, sel: *const u32, n_sel: i64, out: *mut c_void, stream: *mut c_void) -> i32;
    pub fn dbhip_take_block(srcs_host: *const *const c_void, elem_sizes_host: *const i32, ncols: i32, sel: *const u32, n_sel: i64, outs_host: *const *mut c_void, stream: *mut c_void) -> i32;
    pub fn dbhip_take_bitmap(src: *const u8, bit_offset: i64, sel: *const u32, n_sel: i64, out: *mut u8, stream: *mut c_void) -> i32;
    pub fn dbhip_sel_from_ranges(ranges_host: *const u32, n_ranges: i32, out_sel: *mut u32, num_rows: i64, stream: *mut c_void) -> i32;
    pub fn dbhip_sel_from_repeats(repeats_host: *const u32, n_repeats: i32, out_sel: *mut u32, num_rows: i64, stream: *mut c_void) -> i32;
    pub fn dbhip_take_outer(src: *const c_void, src_validity: *const u8, src_validity_offset: i64, elem_size: i32, idx: *const u32, n: i64, out: *mut c_void, out_validity: *mut u8, stream: *mut c_void) -> i32;
    pub fn dbhip_take_chunks(blocks_host: *const *const c_void, n_blocks: i32, elem_size: i32, pairs: *const u32, n: i64, out: *mut c_void, stream: *mut c_void) -> i32;
    pub fn dbhip_group_hash(cols: *const dbhip_col, ncols: i32, n: i64, out_hashes: *mut u64, stream: *mut c_void) -> i32;
    pub fn dbhip_siphash64(col: *const dbhip_col, n: i64, out: *mut u64, stream: *mut c_void) -> i32;
    pub fn dbhip_scatter_block(srcs_host: *const *const c_void, elem_sizes_host: *const i32, ncols: i32, index: *const u32, n: i64, scatter_size: u32, outs_host: *const *mut c_void, stream: *mut c_void) -> i32;
    pub fn dbhip_scatter_indices(keys: *const dbhip_col, nkeys: i32, n: i64, scatter_size: u32, default_index: u64, out_index: *mut u32, out_counts: *mut u64, stream: *mut c_void) -> i32;
    pub fn dbhip_scatter_columns(cols: *const dbhip_col, ncols: i32, index: *const u32, n: i64, scatter_size: u32, out_data_host: *const *mut c_void, out_validity_host: *const *mut u8, out_row_starts_host: *mut i64, stream: *mut c_void) -> i32;
    pub fn dbhip_concat_columns(cols: *const dbhip_col, rows_host: *const i64, bool_bit_offsets_host: *const i64, nblocks: i32, out_data: *mut c_void, out_validity: *mut u8, out_buffers_dev: *mut *const c_void, out_n_buffers_host: *mut i32, stream: *mut c_void) -> i32;
    pub fn dbhip_groupby_create(key_types_host: *const i32, key_nullable_host: *const u8, nkeys: i32, aggs_host: *const dbhip_agg_desc, naggs: i32, initial_capacity: i64, out_host: *mut *mut dbhip_groupby) -> i32;
    pub fn dbhip_groupby_add_block(g: *mut dbhip_groupby, keys: *const dbhip_col, args: *const dbhip_col, n: i64, stream: *mut c_void) -> i32;
    pub fn dbhip_groupby_add_block_filtered(g: *mut dbhip_groupby, keys: *const dbhip_col, args: *const dbhip_col, n: i64, filter_bitmap: *const u8, filter_bit_offset: i64, stream: *mut c_void) -> i32;
    pub fn dbhip_groupby_add_block_program(g: *mut dbhip_groupby, keys: *const dbhip_col, prog: *const dbhip_agg_program, n: i64, filter_bitmap: *const u8, filter_bit_offset: i64, stream: *mut c_void) -> i32;
    pub fn dbhip_groupby_prepare_program(g: *mut dbhip_groupby, keys: *const dbhip_col, prog: *const dbhip_agg_program) -> i32;
    pub fn dbhip_groupby_set_pipelined(g: *mut dbhip_groupby, on: i32, stream: *mut c_void) -> i32;
    pub fn dbhip_groupby_checkpoint(g: *mut dbhip_groupby, out_blocks_committed_host: *mut i64, stream: *mut c_void) -> i32;
    pub fn dbhip_groupby_merge_serialized(g: *mut dbhip_groupby, rows_dev: *const c_void, n_rows: i64, stream: *mut c_void) -> i32;
    pub fn dbhip_groupby_arena(g: *mut dbhip_groupby, out_ptr_host: *mut *const c_void, out_bytes_host: *mut i64, stream: *mut c_void) -> i32;
    pub fn dbhip_groupby_merge_serialized_arena(g: *mut dbhip_groupby, rows_dev: *const c_void, n_rows: i64, arena_dev: *const c_void, stream: *mut c_void) -> i32;
    pub fn dbhip_groupby_num_groups(g: *mut dbhip_groupby, out_host: *mut i64, stream: *mut c_void) -> i32;
    pub fn dbhip_groupby_row_bytes(g: *mut dbhip_groupby, out_host: *mut i64) -> i32;
    pub fn dbhip_groupby_flush_serialized(g: *mut dbhip_groupby, out_rows_dev: *mut c_void, max_rows: i64, out_n_rows_host: *mut i64, stream: *mut c_void) -> i32;
    pub fn dbhip_groupby_flush_block(g: *mut dbhip_groupby, out_block_dev: *mut c_void, max_rows: i64, stream: *mut c_void) -> i32;
    pub fn dbhip_groupby_merge_blocks(g: *mut dbhip_groupby, blocks_dev: *const c_void, n_blocks: i32, max_rows: i64, skip_block: i32, stream: *mut c_void) -> i32;
    pub fn dbhip_groupby_partition_blocks(g: *mut dbhip_groupby, n_buckets: i32, out_blocks_dev: *mut c_void, max_rows: i64, stream: *mut c_void) -> i32;
    pub fn dbhip_groupby_replace_with_blocks(g: *mut dbhip_groupby, blocks_dev: *const c_void, n_blocks: i32, max_rows: i64, stream: *mut c_void) -> i32;
    pub fn dbhip_groupby_flush_partitioned(g: *mut dbhip_groupby, n_buckets: i32, out_rows_dev: *mut c_void, max_rows: i64, out_counts_host: *mut i64, stream: *mut c_void) -> i32;
    pub fn dbhip_groupby_result_type(agg_host: *const dbhip_agg_desc, out_type_host: *mut i32, out_precision_host: *mut u8, out_scale_host: *mut u8) -> i32;
    pub fn dbhip_groupby_flush_result(g: *mut dbhip_groupby, out_keys_host: *const *mut c_void, out_key_validity_host: *const *mut u8, out_aggs_host: *const *mut c_void, out_hashes: *mut u64, max_rows: i64, out_n_rows_host: *mut i64, stream: *mut c_void) -> i32;
    pub fn dbhip_groupby_flush_result_nullable(g: *mut dbhip_groupby, out_keys_host: *const *mut c_void, out_key_validity_host: *const *mut u8, out_aggs_host: *const *mut c_void, out_agg_validity_host: *const *mut u8, out_hashes: *mut u64, max_rows: i64, out_n_rows_host: *mut i64, stream: *mut c_void) -> i32;
    pub fn dbhip_groupby_state_fields(g: *mut dbhip_groupby, out_types_host: *mut i32, out_agg_index_host: *mut i32, max_fields: i32, out_n_fields_host: *mut i32) -> i32;
    pub fn dbhip_groupby_flush_state_block(g: *mut dbhip_groupby, out_keys_host: *const *mut c_void, out_key_validity_host: *const *mut u8, out_state_fields_host: *const *mut c_void, out_hashes: *mut u64, max_rows: i64, out_n_rows_host: *mut i64, stream: *mut c_void) -> i32;
    pub fn dbhip_groupby_merge_state_block(g: *mut dbhip_groupby, keys: *const dbhip_col, states: *const dbhip_col, n: i64, stream: *mut c_void) -> i32;
    pub fn dbhip_groupby_reset(g: *mut dbhip_groupby, stream: *mut c_void) -> i32;
    pub fn dbhip_groupby_destroy(g: *mut dbhip_groupby) -> i32;
    pub fn dbhip_q1_create_groupby(out_host: *mut *mut dbhip_groupby) -> i32;
    pub fn dbhip_q1_fused(g: *mut dbhip_groupby, l_quantity: *const i64, l_extendedprice: *const i64, l_discount: *const i64, l_tax: *const i64, l_returnflag_views: *const c_void, l_linestatus_views: *const c_void, l_shipdate: *const i32, shipdate_cutoff: i32, n: i64, stream: *mut c_void) -> i32;
    pub fn dbhip_keys_method(cols: *const dbhip_col, ncols: i32, out_key_bytes_host: *mut i32) -> i32;
    pub fn dbhip_pack_keys(cols: *const dbhip_col, ncols: i32, n: i64, key_bytes: i32, out_keys: *mut c_void, out_all_valid: *mut u8, stream: *mut c_void) -> i32;
    pub fn dbhip_serialize_keys_offsets(cols: *const dbhip_col, ncols: i32, n: i64, out_offsets: *mut u64, out_all_valid: *mut u8, out_total_bytes_host: *mut u64, stream: *mut c_void) -> i32;
    pub fn dbhip_serialize_keys(cols: *const dbhip_col, ncols: i32, n: i64, offsets: *const u64, out_data: *mut u8, stream: *mut c_void) -> i32;
    pub fn dbhip_join_create(expected_build_rows: i64, out_host: *mut *mut dbhip_join) -> i32;
    pub fn dbhip_join_create_keys(expected_build_rows: i64, key_bytes: i32, out_host: *mut *mut dbhip_join) -> i32;
    pub fn dbhip_join_add_build(j: *mut dbhip_join, keys: *const c_void, validity: *const u8, n: i64, stream: *mut c_void) -> i32;
    pub fn dbhip_join_finalize(j: *mut dbhip_join, stream: *mut c_void) -> i32;
    pub fn dbhip_join_probe_count(j: *mut dbhip_join, keys: *const c_void, validity: *const u8, n: i64, out_total_host: *mut u64, stream: *mut c_void) -> i32;
    pub fn dbhip_join_probe(j: *mut dbhip_join, keys: *const c_void, validity: *const u8, n: i64, out_probe_idx: *mut u32, out_build_row: *mut u32, max_pairs: i64, out_n_pairs_host: *mut u64, stream: *mut c_void) -> i32;
    pub fn dbhip_join_probe_mark(j: *mut dbhip_join, keys: *const c_void, validity: *const u8, n: i64, out_matched_bitmap: *mut u8, out_n_matched_host: *mut u64, stream: *mut c_void) -> i32;
    pub fn dbhip_join_mark_build(j: *mut dbhip_join, build_rows: *const u32, n_pairs: i64, stream: *mut c_void) -> i32;
    pub fn dbhip_join_build_matched(j: *mut dbhip_join, out_bitmap: *mut u8, out_build_rows_host: *mut i64, stream: *mut c_void) -> i32;
    pub fn dbhip_join_destroy(j: *mut dbhip_join) -> i32;
    pub fn dbhip_join_create_binary(expected_build_rows: i64, out_host: *mut *mut dbhip_join_binary) -> i32;
    pub fn dbhip_join_add_build_binary(j: *mut dbhip_join_binary, offsets: *const u64, data: *const u8, validity: *const u8, n: i64, stream: *mut c_void) -> i32;
    pub fn dbhip_join_finalize_binary(j: *mut dbhip_join_binary, stream: *mut c_void) -> i32;
    pub fn dbhip_join_probe_count_binary(j: *mut dbhip_join_binary, offsets: *const u64, data: *const u8, validity: *const u8, n: i64, out_max_pairs_host: *mut u64, stream: *mut c_void) -> i32;
    pub fn dbhip_join_probe_binary(j: *mut dbhip_join_binary, offsets: *const u64, data: *const u8, validity: *const u8, n: i64, out_probe_idx: *mut u32, out_build_row: *mut u32, max_pairs: i64, out_n_pairs_host: *mut u64, out_matched_bitmap: *mut u8, stream: *mut c_void) -> i32;
    pub fn dbhip_join_destroy_binary(j: *mut dbhip_join_binary) -> i32;
    pub fn dbhip_sort_perm(keys: *const dbhip_col, desc_host: *const u8, nulls_first_host: *const u8, nkeys: i32, n: i64, limit: i64, out_perm: *mut u32, stream: *mut c_void) -> i32;
    pub fn dbhip_merge_sorted_perm(keys: *const dbhip_col, desc_host: *const u8, nulls_first_host: *const u8, nkeys: i32, run_offsets_host: *const i64, nruns: i32, limit: i64, out_perm: *mut u32, stream: *mut c_void) -> i32;
    pub fn dbhip_sort_bound_partition(keys: *const dbhip_col, bounds: *const dbhip_col, desc_host: *const u8, nulls_first_host: *const u8, nkeys: i32, n: i64, nbounds: i64, out_part: *mut u32, out_counts: *mut u64, stream: *mut c_void) -> i32;
    pub fn dbhip_vec_distance_rows(metric: i32, elem_type: i32, lhs: *const c_void, lhs_is_scalar: i32, rhs: *const c_void, rhs_is_scalar: i32, n: i64, dim: i32, out: *mut c_void, stream: *mut c_void) -> i32;
    pub fn dbhip_vec_distance(metric: i32, base: *const f32, n: i64, dim: i32, queries: *const f32, nq: i32, out: *mut f32, stream: *mut c_void) -> i32;
    pub fn dbhip_vec_topk(metric: i32, base: *const f32, n: i64, dim: i32, queries: *const f32, nq: i32, k: i32, out_idx: *mut u32, out_dist: *mut f32, stream: *mut c_void) -> i32;
    pub fn dbhip_vec_topk_merge(dists: *const f32, ids: *const u32, m: i64, nq: i32, k: i32, out_idx: *mut u32, out_dist: *mut f32, stream: *mut c_void) -> i32;
    pub fn dbhip_vec_index_build(metric: i32, base: *const f32, n: i64, dim: i32, out_host: *mut *mut dbhip_vec_index, stream: *mut c_void) -> i32;
    pub fn dbhip_vec_index_search(ix: *mut dbhip_vec_index, queries: *const f32, nq: i32, k: i32, out_idx: *mut u32, out_dist: *mut f32, stream: *mut c_void) -> i32;
    pub fn dbhip_vec_index_destroy(ix: *mut dbhip_vec_index) -> i32;
    pub fn dbhip_score_u8(is_l1: i32, query: *const u8, base: *const u8, n: i64, dim: i32, out: *mut f32, stream: *mut c_void) -> i32;
    pub fn dbhip_comm_unique_id(out_id128_host: *mut u8) -> i32;
    pub fn dbhip_comm_create(rank: i32, world: i32, id128_host: *const u8, out_host: *mut *mut dbhip_comm) -> i32;
    pub fn dbhip_comm_create_loopback(group_id: u64, rank: i32, world: i32, out_host: *mut *mut dbhip_comm) -> i32;
    pub fn dbhip_comm_destroy(c: *mut dbhip_comm) -> i32;
    pub fn dbhip_comm_abort(c: *mut dbhip_comm) -> i32;
    pub fn dbhip_comm_allgather(c: *mut dbhip_comm, send_dev: *const c_void, recv_dev: *mut c_void, bytes_per_rank: i64, stream: *mut c_void) -> i32;
    pub fn dbhip_comm_alltoall(c: *mut dbhip_comm, send_dev: *const c_void, recv_dev: *mut c_void, bytes_per_peer: i64, stream: *mut c_void) -> i32;
    pub fn dbhip_comm_allreduce_sum_u64(c: *mut dbhip_comm, send_dev: *const u64, recv_dev: *mut u64, count: i64, stream: *mut c_void) -> i32;
    pub fn dbhip_groupby_exchange_allgather(g: *mut dbhip_groupby, c: *mut dbhip_comm, max_rows: i64, stream: *mut c_void) -> i32;
    pub fn dbhip_groupby_exchange_alltoall(g: *mut dbhip_groupby, c: *mut dbhip_comm, max_rows: i64, stream: *mut c_void) -> i32;
    pub fn dbhip_exchange_begin(c: *mut dbhip_comm, cols: *const dbhip_col, ncols: i32, dest_index: *const u32, n: i64, out_recv_rows_host: *mut i64, out_host: *mut *mut dbhip_exchange, stream: *mut c_void) -> i32;
    pub fn dbhip_shuffle_exchange_begin(c: *mut dbhip_comm, keys: *const dbhip_col, nkeys: i32, cols: *const dbhip_col, ncols: i32, n: i64, out_recv_rows_host: *mut i64, out_host: *mut *mut dbhip_exchange, stream: *mut c_void) -> i32;
    pub fn dbhip_sort_exchange_begin(c: *mut dbhip_comm, keys: *const dbhip_col, bounds: *const dbhip_col, desc_host: *const u8, nulls_first_host: *const u8, nkeys: i32, nbounds: i64, cols: *const dbhip_col, ncols: i32, n: i64, out_recv_rows_host: *mut i64, out_host: *mut *mut dbhip_exchange, stream: *mut c_void) -> i32;
    pub fn dbhip_exchange_finish(x: *mut dbhip_exchange, out_data_host: *const *mut c_void, out_validity_host: *const *mut u8, out_src_starts_host: *mut i64, stream: *mut c_void) -> i32;
    pub fn dbhip_exchange_string_bytes(x: *mut dbhip_exchange, out_bytes_host: *mut i64) -> i32;
    pub fn dbhip_exchange_finish_strings(x: *mut dbhip_exchange, out_data_host: *const *mut c_void, out_validity_host: *const *mut u8, out_string_bytes_host: *const *mut u8, out_src_starts_host: *mut i64, stream: *mut c_void) -> i32;
    pub fn dbhip_exchange_destroy(x: *mut dbhip_exchange) -> i32;
    pub fn dbhip_vec_topk_allgather(c: *mut dbhip_comm, idx_dev: *const u32, dist_dev: *const f32, nq: i32, k: i32, row_offset: u64, out_idx_dev: *mut u32, out_dist_dev: *mut f32, stream: *mut c_void) -> i32;
    pub fn dbhip_kmeans(distance_type: i32, data: *const f32, rows: i64, dim: i32, rows_per_cluster: i64, normalize_input: i32, out_assignments: *mut u32, out_distances: *mut f32, out_k_host: *mut i64, out_iterations_host: *mut i32, stream: *mut c_void) -> i32;
    pub fn dbhip_vec_kernel_f32(which: i32, a: *const f32, b: *const f32, n: i64, dim: i32, out: *mut f32, stream: *mut c_void) -> i32;
    pub fn dbhip_pq_chunk_open(chunk_host: *const u8, chunk_len: i64, codec: i32, physical_type: i32, type_length: i32, max_def_level: i32, max_rep_level: i32, out_type: i32, out_host: *mut *mut dbhip_pq_chunk, info_host: *mut dbhip_pq_info) -> i32;
    pub fn dbhip_pq_chunk_validity(c: *mut dbhip_pq_chunk, out_ptr_host: *mut *const u8, out_bytes_host: *mut i64) -> i32;
    pub fn dbhip_pq_chunk_image(c: *mut dbhip_pq_chunk, out_ptr_host: *mut *const u8, out_len_host: *mut i64) -> i32;
    pub fn dbhip_pq_chunk_decode(c: *mut dbhip_pq_chunk, chunk_dev: *const u8, out_values_dev: *mut c_void, out_validity_dev: *mut u8, stream: *mut c_void) -> i32;
    pub fn dbhip_pq_chunk_open_device(chunk_host: *const u8, chunk_len: i64, codec: i32, physical_type: i32, type_length: i32, max_def_level: i32, max_rep_level: i32, out_type: i32, out_host: *mut *mut dbhip_pq_chunk, info_host: *mut dbhip_pq_info) -> i32;
    pub fn dbhip_pq_chunk_decode_device(c: *mut dbhip_pq_chunk, chunk_dev: *const u8, image_dev: *mut u8, out_values_dev: *mut c_void, out_validity_dev: *mut u8, out_nulls_host: *mut i64, stream: *mut c_void) -> i32;
    pub fn dbhip_pq_chunks_decode_device(chunks: *const *mut dbhip_pq_chunk, n_chunks: i32, chunk_dev: *const *const u8, image_dev: *const *mut u8, out_values_dev: *const *mut c_void, out_validity_dev: *const *mut u8, out_nulls_host: *mut i64, out_status_host: *mut i32, stream: *mut c_void) -> i32;
    pub fn dbhip_pq_chunk_open_device_list(chunk_host: *const u8, chunk_len: i64, codec: i32, physical_type: i32, type_length: i32, list_nullable: i32, element_nullable: i32, out_type: i32, out_host: *mut *mut dbhip_pq_chunk, info_host: *mut dbhip_pq_info) -> i32;
    pub fn dbhip_pq_chunk_decode_device_list(c: *mut dbhip_pq_chunk, chunk_dev: *const u8, image_dev: *mut u8, out_offsets_dev: *mut u64, out_list_validity_dev: *mut u8, out_values_dev: *mut c_void, out_elem_validity_dev: *mut u8, out_rows_host: *mut i64, out_elems_host: *mut i64, out_null_lists_host: *mut i64, stream: *mut c_void) -> i32;
    pub fn dbhip_pq_chunk_close(c: *mut dbhip_pq_chunk) -> i32;
    pub fn dbhip_hnsw_build(vectors_dev: *const f32, n: i64, dim: i32, distance: i32, m: i32, ef_construct: i32, seed: u64, out: *mut *mut dbhip_hnsw, stream: *mut c_void) -> i32;
    pub fn dbhip_hnsw_build_sequential(vectors_dev: *const f32, n: i64, dim: i32, distance: i32, m: i32, ef_construct: i32, levels_host: *const i32, out: *mut *mut dbhip_hnsw, stream: *mut c_void) -> i32;
    pub fn dbhip_hnsw_from_graph(vectors_dev: *const f32, n: i64, dim: i32, distance: i32, m: i32, levels_host: *const i32, links_host: *const u32, nlinks_host: *const i32, entry_point: u32, entry_level: i32, out: *mut *mut dbhip_hnsw, stream: *mut c_void) -> i32;
    pub fn dbhip_hnsw_open(encoded_dev: *const u8, alpha: f32, offset: f32, multiplier: f32, n: i64, dim: i32, distance: i32, m: i32, levels_host: *const i32, links_host: *const u32, nlinks_host: *const i32, entry_point: u32, entry_level: i32, out: *mut *mut dbhip_hnsw, stream: *mut c_void) -> i32;
    pub fn dbhip_hnsw_export_graph(h: *mut dbhip_hnsw, levels_host: *mut i32, links_host: *mut u32, nlinks_host: *mut i32, out_n_lists_host: *mut i64, out_entry_point_host: *mut u32, out_entry_level_host: *mut i32, stream: *mut c_void) -> i32;
    pub fn dbhip_hnsw_search(h: *mut dbhip_hnsw, queries_dev: *const f32, nq: i32, limit: i32, out_ids_dev: *mut u32, out_dist_dev: *mut f32, stream: *mut c_void) -> i32;
    pub fn dbhip_hnsw_scores(h: *mut dbhip_hnsw, queries_dev: *const f32, nq: i32, out_dev: *mut f32, stream: *mut c_void) -> i32;
    pub fn dbhip_hnsw_encoded(h: *mut dbhip_hnsw, out_dev: *mut c_void, stream: *mut c_void) -> i32;
    pub fn dbhip_hnsw_meta(h: *mut dbhip_hnsw, alpha_host: *mut f32, offset_host: *mut f32, multiplier_host: *mut f32, actual_dim_host: *mut i32) -> i32;
    pub fn dbhip_hnsw_destroy(h: *mut dbhip_hnsw) -> i32;
    pub fn dbhip_groupby_debug_set_hash_mask(g: *mut dbhip_groupby, mask: u64) -> i32;
    pub fn dbhip_join_binary_debug_set_hash_mask(mask: u64) -> i32;
    pub fn dbhip_groupby_debug_set_partition_bits(g: *mut dbhip_groupby, bits: i32) -> i32;
    pub fn dbhip_groupby_debug_set_compact(g: *mut dbhip_groupby, on: i32) -> i32;
    pub fn dbhip_fagg_stats(out3_host: *mut u64) -> i32;
    pub fn dbhip_scratch_stats(out2_host: *mut u64) -> i32;
    pub fn dbhip_jit_compile_check(log_out_host: *mut c_char, log_cap: i64) -> i64;
    pub fn dbhip_jit_offline(key_types_host: *const i32, key_nullable_host: *const u8, nkeys: i32, aggs_host: *const dbhip_agg_desc, naggs: i32, keys: *const dbhip_col, prog: *const dbhip_agg_program, slots: i32, code_out_host: *mut c_char, code_cap: i64, log_out_host: *mut c_char, log_cap: i64) -> i64;
}

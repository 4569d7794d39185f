"""ctypes binding of include/dbhip.h (the drop-in C-ABI).  Fails loudly."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class DbhipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"dbhip error {code}: {msg}")
        self.code = code


# status codes (include/dbhip.h)
OK, ERR_INVALID, ERR_HIP, ERR_NO_DEVICE, ERR_ROW_ERRORS, ERR_OVERFLOW, ERR_CAPACITY, ERR_UNSUPPORTED, ERR_CANCELLED = range(9)

# dbhip_type
T_BOOL, T_I8, T_I16, T_I32, T_I64, T_U8, T_U16, T_U32, T_U64, T_F32, T_F64, T_DATE, T_TIMESTAMP, T_DEC64, T_DEC128, T_STRING, T_DEC256 = range(1, 18)
OP_PLUS, OP_MINUS, OP_MULTIPLY, OP_DIVIDE, OP_INTDIV, OP_MODULO, OP_DIV0, OP_DIVNULL = range(8)
CMP_EQ, CMP_NOTEQ, CMP_LT, CMP_LTE, CMP_GT, CMP_GTE = range(6)
AGG_COUNT, AGG_SUM, AGG_MIN, AGG_MAX = range(4)
VEC_COSINE, VEC_L2, VEC_DOT, VEC_L1, VEC_NORM = range(5)


class Col(C.Structure):
    """dbhip_col"""
    _fields_ = [
        ("type", C.c_int32), ("is_scalar", C.c_int32), ("data", C.c_void_p),
        ("validity", C.c_void_p), ("validity_offset", C.c_int64),
        ("buffers", C.c_void_p), ("n_buffers", C.c_int32),
        ("precision", C.c_uint8), ("scale", C.c_uint8), ("_pad", C.c_uint8 * 2),
    ]


EX_LOAD, EX_CONST, EX_PLUS, EX_MINUS, EX_MULTIPLY, EX_DIVIDE, EX_EQ, EX_NOTEQ, EX_LT, EX_LTE, EX_GT, EX_GTE, EX_AND, EX_OR, EX_NOT, EX_CAST, EX_IF, EX_IS_TRUE = range(18)


class ExprIns(C.Structure):
    """dbhip_expr_ins"""
    _fields_ = [("op", C.c_int32), ("dst", C.c_int32), ("a", C.c_int32), ("b", C.c_int32), ("type", C.c_int32),
                ("precision", C.c_uint8), ("scale", C.c_uint8), ("_pad", C.c_uint8 * 2), ("imm", C.c_uint64)]


class AggProgram(C.Structure):
    """dbhip_agg_program"""
    _fields_ = [("prog", C.c_void_p), ("n_ins", C.c_int32), ("inputs", C.c_void_p), ("n_inputs", C.c_int32), ("filter_reg", C.c_int32),
                ("arg_regs", C.c_void_p)]


class PqInfo(C.Structure):
    """dbhip_pq_info"""
    _fields_ = [("num_values", C.c_int64), ("num_nulls", C.c_int64), ("out_type", C.c_int32), ("has_validity", C.c_int32),
                ("out_bytes", C.c_int64), ("validity_bytes", C.c_int64), ("n_pages", C.c_int64), ("n_dict_values", C.c_int64),
                ("image_bytes", C.c_int64)]


# parquet.thrift Type numbers
PQ_BOOLEAN, PQ_INT32, PQ_INT64, PQ_INT96, PQ_FLOAT, PQ_DOUBLE, PQ_BYTE_ARRAY, PQ_FLBA = range(8)
# parquet.thrift CompressionCodec numbers the library decodes
PQ_UNCOMPRESSED, PQ_SNAPPY, PQ_ZSTD, PQ_LZ4_RAW = 0, 1, 6, 7


class AggDesc(C.Structure):
    """dbhip_agg_desc"""
    _fields_ = [("kind", C.c_int32), ("arg_type", C.c_int32), ("arg_precision", C.c_uint8),
                ("arg_scale", C.c_uint8), ("arg_nullable", C.c_uint8), ("_pad", C.c_uint8)]


def library_path():
    # DBHIP_LIBRARY: another build of the same library (tools/probes A/B runs: an experiments build beside the shipped one)
    return os.environ.get("DBHIP_LIBRARY") or os.path.join(_HERE, "libdbhip.so")


# every symbol include/dbhip.h declares (tests check that the built library exports all of them)
SYMBOLS = [
    "dbhip_abi_version", "dbhip_init", "dbhip_device_count", "dbhip_last_error", "dbhip_alloc", "dbhip_free", "dbhip_trim",
    "dbhip_memcpy_h2d", "dbhip_memcpy_d2h", "dbhip_memset", "dbhip_stream_create", "dbhip_stream_destroy", "dbhip_stream_release_scratch", "dbhip_stream_cancel", "dbhip_stream_cancel_clear",
    "dbhip_stream_sync", "dbhip_event_create", "dbhip_event_record", "dbhip_event_elapsed_ms",
    "dbhip_event_destroy", "dbhip_last_kernel_ms", "dbhip_arith", "dbhip_arith_result_type", "dbhip_sum_a_plus_b_mul_c_i64",
    "dbhip_sum", "dbhip_expr_eval", "dbhip_decimal_result_size", "dbhip_decimal_arith", "dbhip_decimal_neg", "dbhip_decimal_cast", "dbhip_cmp", "dbhip_bitmap_binary",
    "dbhip_bitmap_count", "dbhip_filter_select", "dbhip_select_cmp", "dbhip_select_bool", "dbhip_take", "dbhip_take_block", "dbhip_take_bitmap", "dbhip_group_hash",
    "dbhip_groupby_create", "dbhip_groupby_add_block", "dbhip_groupby_merge_serialized", "dbhip_groupby_merge_state_block",
    "dbhip_groupby_num_groups", "dbhip_groupby_row_bytes", "dbhip_groupby_flush_serialized", "dbhip_groupby_flush_block",
    "dbhip_groupby_merge_blocks", "dbhip_groupby_add_block_filtered", "dbhip_groupby_partition_blocks",
    "dbhip_groupby_replace_with_blocks", "dbhip_groupby_flush_partitioned", "dbhip_groupby_flush_result_nullable",
    "dbhip_groupby_state_fields", "dbhip_groupby_flush_state_block", "dbhip_groupby_add_block_program", "dbhip_groupby_prepare_program", "dbhip_groupby_set_pipelined", "dbhip_groupby_checkpoint", "dbhip_groupby_arena", "dbhip_groupby_merge_serialized_arena", "dbhip_sel_from_ranges", "dbhip_sel_from_repeats", "dbhip_take_chunks", "dbhip_take_outer", "dbhip_cast", "dbhip_memcpy_d2d",
    "dbhip_groupby_result_type", "dbhip_groupby_flush_result", "dbhip_groupby_reset",
    "dbhip_groupby_destroy", "dbhip_q1_create_groupby", "dbhip_q1_fused", "dbhip_keys_method", "dbhip_pack_keys", "dbhip_serialize_keys_offsets", "dbhip_serialize_keys", "dbhip_join_create_binary",
    "dbhip_join_add_build_binary", "dbhip_join_finalize_binary", "dbhip_join_probe_count_binary", "dbhip_join_probe_binary", "dbhip_join_destroy_binary",
    "dbhip_join_create", "dbhip_join_create_keys", "dbhip_join_probe_mark",
    "dbhip_join_add_build", "dbhip_join_finalize", "dbhip_join_probe_count", "dbhip_join_probe",
    "dbhip_join_destroy", "dbhip_join_mark_build", "dbhip_join_build_matched", "dbhip_sort_perm", "dbhip_merge_sorted_perm", "dbhip_sort_bound_partition", "dbhip_bitmap_set_indices", "dbhip_siphash64", "dbhip_scatter_indices", "dbhip_scatter_block", "dbhip_vec_distance", "dbhip_vec_distance_rows", "dbhip_vec_topk", "dbhip_score_u8",
    "dbhip_vec_topk_merge", "dbhip_vec_index_build", "dbhip_vec_index_search", "dbhip_vec_index_destroy",
    "dbhip_comm_unique_id", "dbhip_comm_create", "dbhip_comm_destroy", "dbhip_comm_abort", "dbhip_comm_allgather", "dbhip_comm_alltoall",
    "dbhip_comm_allreduce_sum_u64", "dbhip_groupby_exchange_allgather", "dbhip_groupby_exchange_alltoall", "dbhip_kmeans", "dbhip_vec_kernel_f32", "dbhip_hnsw_build", "dbhip_hnsw_build_sequential", "dbhip_hnsw_from_graph", "dbhip_hnsw_open", "dbhip_hnsw_export_graph", "dbhip_hnsw_search", "dbhip_hnsw_scores",
    "dbhip_hnsw_encoded", "dbhip_hnsw_meta", "dbhip_hnsw_destroy",
    "dbhip_pq_chunk_open", "dbhip_pq_chunk_validity", "dbhip_pq_chunk_image", "dbhip_pq_chunk_decode", "dbhip_pq_chunk_close",
    "dbhip_pq_chunk_open_device", "dbhip_pq_chunk_decode_device", "dbhip_pq_chunks_decode_device", "dbhip_pq_chunk_open_device_list", "dbhip_pq_chunk_decode_device_list",
    "dbhip_scatter_columns", "dbhip_concat_columns", "dbhip_comm_create_loopback", "dbhip_exchange_begin", "dbhip_shuffle_exchange_begin", "dbhip_sort_exchange_begin", "dbhip_exchange_finish", "dbhip_exchange_string_bytes", "dbhip_exchange_finish_strings", "dbhip_exchange_destroy", "dbhip_vec_topk_allgather",
    # diagnostics and test hooks (declared in the header's last section)
    "dbhip_groupby_debug_set_hash_mask", "dbhip_groupby_debug_set_partition_bits", "dbhip_groupby_debug_set_compact", "dbhip_join_binary_debug_set_hash_mask",
    "dbhip_fagg_stats", "dbhip_scratch_stats", "dbhip_jit_compile_check", "dbhip_jit_offline",
]
_RESTYPE_I64 = {"dbhip_jit_compile_check", "dbhip_jit_offline"}


def load_library():
    """dlopen libdbhip.so (no GPU needed to load; compute entry points need one)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = library_path()
    if not os.path.exists(path):
        raise DbhipError(ERR_NO_DEVICE, f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                                       "(there is no CPU fallback)")
    L = C.CDLL(path, mode=C.RTLD_GLOBAL)
    L.dbhip_last_error.restype = C.c_char_p
    for name in SYMBOLS:
        fn = getattr(L, name)
        if name != "dbhip_last_error":
            fn.restype = C.c_int64 if name in _RESTYPE_I64 else C.c_int32
    _LIB = L
    return L


def lib():
    return load_library()


def check(rc):
    if rc != OK:
        raise DbhipError(rc, load_library().dbhip_last_error().decode("utf-8", "replace"))
    return rc

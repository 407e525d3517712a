"""ctypes mirror of include/tsq.h (struct layouts, enums).  Pure declarations — no library loading.

Shared by the product binding (tinysql_amd._lib) and by the test-only oracle binding
(tests/oracle_binding.py), so both sides are driven by byte-identical structs.
"""
import ctypes as C

TSQ_ABI_VERSION = 7
RADIX_AUTO, RADIX_OFF, RADIX_FORCE = -1, 0, 1
AGGFAST_AUTO, AGGFAST_OFF, AGGFAST_FORCE = -1, 0, 1
JIT_AUTO, JIT_OFF, JIT_FORCE = -1, 0, 1

# status codes
OK, ERR_INVALID, ERR_UNSUPPORTED, ERR_OOM_DEVICE, ERR_HIP = 0, 1, 2, 3, 4
ERR_OVERFLOW_BIGINT, ERR_OVERFLOW_BIGINT_UNSIGNED, ERR_OVERFLOW_DOUBLE = 5, 6, 7
ERR_CANCELLED, ERR_NO_DEVICE, ERR_DIV_BY_ZERO = 8, 9, 10
STATUS_NAMES = {
    0: "OK", 1: "INVALID", 2: "UNSUPPORTED", 3: "OOM_DEVICE", 4: "HIP", 5: "OVERFLOW_BIGINT",
    6: "OVERFLOW_BIGINT_UNSIGNED", 7: "OVERFLOW_DOUBLE", 8: "CANCELLED", 9: "NO_DEVICE", 10: "DIV_BY_ZERO",
}

# column types
I64, U64, F32, F64, BYTES = 0, 1, 2, 3, 4
COL_DEVICE = 1
COL_BORROW = 2
COL_RETAIN = 4

# generator kinds
GEN_SEQ, GEN_AFFINE, GEN_RAND_MOD, GEN_RAND_F64, GEN_HASH_OF_COL, GEN_ZIPF_OCT = 0, 1, 2, 3, 4, 5

# join types / agg funcs / modes
JOIN_INNER, JOIN_LEFT_OUTER, JOIN_RIGHT_OUTER = 0, 1, 2
AGG_COUNT, AGG_SUM, AGG_AVG, AGG_MAX, AGG_MIN, AGG_FIRSTROW = 0, 1, 2, 3, 4, 5
MODE_COMPLETE, MODE_FINAL, MODE_PARTIAL1, MODE_PARTIAL2 = 0, 1, 2, 3

MAX_KEYS, MAX_COLS, MAX_AGGS, MAX_GROUP_KEYS = 4, 16, 16, 4
EXPR_MAX_OPS, EXPR_MAX_STACK, EXPR_MAX_CONSTS, EXPR_STR_POOL = 64, 12, 32, 256

# opcodes
OP_COL_INT, OP_COL_REAL, OP_CONST_INT, OP_CONST_REAL, OP_CONST_NULL_INT, OP_CONST_NULL_REAL = 1, 2, 3, 4, 5, 6
OP_PLUS_REAL, OP_MINUS_REAL, OP_MUL_REAL, OP_DIV_REAL = 10, 11, 12, 13
OP_PLUS_INT, OP_MINUS_INT, OP_MUL_INT, OP_MUL_INT_UNSIGNED = 14, 15, 16, 17
OP_LT_INT, OP_LE_INT, OP_GT_INT, OP_GE_INT, OP_EQ_INT, OP_NE_INT = 20, 21, 22, 23, 24, 25
OP_LT_REAL, OP_LE_REAL, OP_GT_REAL, OP_GE_REAL, OP_EQ_REAL, OP_NE_REAL = 26, 27, 28, 29, 30, 31
OP_LOGIC_AND, OP_LOGIC_OR, OP_NOT_INT, OP_NOT_REAL, OP_NEG_INT, OP_NEG_REAL = 40, 41, 42, 43, 44, 45
OP_ISNULL_INT, OP_ISNULL_REAL = 46, 47
OP_IFNULL_INT, OP_IFNULL_REAL, OP_IF_INT, OP_IF_REAL = 50, 51, 52, 53
OP_IN_INT, OP_IN_REAL = 60, 61
OP_COL_STR, OP_CONST_STR, OP_CONST_NULL_STR = 70, 71, 72
OP_LT_STR, OP_LE_STR, OP_GT_STR, OP_GE_STR, OP_EQ_STR, OP_NE_STR = 73, 74, 75, 76, 77, 78
OP_STRCMP, OP_LENGTH, OP_ISNULL_STR, OP_IFNULL_STR, OP_IF_STR, OP_IN_STR = 79, 80, 81, 82, 83, 84
F_LHS_UNSIGNED, F_RHS_UNSIGNED, F_FORCE_SIGNED = 1, 2, 4


RC_HANDLE, RC_HAS_DEFAULT, RC_BIT = 1, 2, 4  # tsq_rowcodec_col.flags (a bit column: its byte size in bits 8..11)
ENC_COMPARABLE = 1  # tsq_rows_encode col_flags


class RowcodecCol(C.Structure):
    """tsq_rowcodec_col — one requested column of a stored-row scan (rowcodec.ColInfo, util/rowcodec/decoder.go:45-55)."""
    _fields_ = [
        ("col_id", C.c_int64),
        ("type", C.c_int32),
        ("flags", C.c_uint32),
        ("def_bits", C.c_uint64),
        ("def_bytes", C.c_void_p),
        ("def_len", C.c_int64),
    ]


class Col(C.Structure):
    """tsq_col — mirrors util/chunk/column.go:28-34."""
    _fields_ = [
        ("data", C.c_void_p),
        ("null_bitmap", C.c_void_p),
        ("offsets", C.c_void_p),
        ("length", C.c_int64),
        ("elem_size", C.c_int32),
        ("type", C.c_int32),
        ("flags", C.c_uint32),
        ("reserved", C.c_uint32),
    ]


class GenSpec(C.Structure):
    _fields_ = [
        ("kind", C.c_int32), ("table", C.c_int32), ("col", C.c_int32), ("null_pct", C.c_int32),
        ("seed", C.c_uint64), ("start", C.c_int64), ("a", C.c_uint64), ("b", C.c_uint64), ("m", C.c_uint64),
    ]


class ExprOp(C.Structure):
    _fields_ = [("opcode", C.c_uint8), ("flags", C.c_uint8), ("arg", C.c_uint16), ("aux", C.c_uint32)]


class ExprProg(C.Structure):
    _fields_ = [
        ("n_ops", C.c_int32), ("n_consts", C.c_int32), ("result_type", C.c_int32), ("result_unsigned", C.c_int32),
        ("ops", ExprOp * EXPR_MAX_OPS), ("consts", C.c_int64 * EXPR_MAX_CONSTS),
        ("n_str_bytes", C.c_int32), ("reserved", C.c_int32), ("str_pool", C.c_uint8 * EXPR_STR_POOL),
    ]


class JoinCfg(C.Structure):
    _fields_ = [
        ("join_type", C.c_int32), ("build_is_right", C.c_int32), ("n_keys", C.c_int32),
        ("build_key_idx", C.c_int32 * MAX_KEYS), ("probe_key_idx", C.c_int32 * MAX_KEYS),
        ("n_build_cols", C.c_int32), ("n_probe_cols", C.c_int32),
        ("build_types", C.c_int32 * MAX_COLS), ("probe_types", C.c_int32 * MAX_COLS),
        ("est_build_rows", C.c_int64), ("max_chunk_size", C.c_int32), ("concurrency", C.c_int32),
        ("probe_batch_rows", C.c_int64),
        ("other_conds", C.POINTER(ExprProg)), ("n_other_conds", C.c_int32),
        ("outer_filters", C.POINTER(ExprProg)), ("n_outer_filters", C.c_int32),
    ]


class AggFunc(C.Structure):
    _fields_ = [("func", C.c_int32), ("mode", C.c_int32), ("arg_col", C.c_int32), ("arg_col2", C.c_int32),
                ("arg_type", C.c_int32), ("reserved", C.c_int32)]


class AggCfg(C.Structure):
    _fields_ = [
        ("n_group_keys", C.c_int32), ("group_key_col", C.c_int32 * MAX_GROUP_KEYS),
        ("group_key_type", C.c_int32 * MAX_GROUP_KEYS), ("n_aggs", C.c_int32), ("aggs", AggFunc * MAX_AGGS),
        ("n_input_cols", C.c_int32), ("input_types", C.c_int32 * MAX_COLS), ("est_groups", C.c_int64),
        ("max_chunk_size", C.c_int32), ("reserved", C.c_int32),
    ]


class SortCfg(C.Structure):
    _fields_ = [
        ("n_cols", C.c_int32), ("col_types", C.c_int32 * MAX_COLS), ("n_keys", C.c_int32), ("key_col", C.c_int32 * MAX_KEYS),
        ("key_desc", C.c_int32 * MAX_KEYS), ("limit_offset", C.c_int64), ("limit_count", C.c_int64), ("max_chunk_size", C.c_int32),
        ("reserved", C.c_int32),
    ]


class Stats(C.Structure):
    _fields_ = [
        ("build_rows", C.c_int64), ("build_rows_inserted", C.c_int64), ("probe_rows", C.c_int64),
        ("out_rows", C.c_int64), ("table_bytes", C.c_int64), ("table_buckets", C.c_int64),
        ("build_kernel_ms", C.c_double), ("probe_kernel_ms", C.c_double), ("h2d_bytes", C.c_int64),
        ("d2h_bytes", C.c_int64), ("kernel_launches", C.c_int64),
        ("partition_kernel_ms", C.c_double), ("radix_probe_kernel_ms", C.c_double), ("partition_kernel_ms_sum", C.c_double),
        ("radix_probe_kernel_ms_sum", C.c_double), ("radix_timed_batches", C.c_int64), ("radix_batches", C.c_int64), ("radix_overflow_rows", C.c_int64),
        ("radix_bits", C.c_int32), ("build_partitioned", C.c_int32), ("build_handed_back_rows", C.c_int64),
        ("table_slice_bits", C.c_int32), ("build_slice_retries", C.c_int32),
        ("probe_route", C.c_int32), ("packed_key_bits", C.c_int32), ("packed_build_ms", C.c_double),
        ("heap_bytes", C.c_int64), ("heap_compactions", C.c_int64),
        ("shared_build", C.c_int32), ("dense_flushes", C.c_int32), ("shared_image_bytes", C.c_int64), ("shared_allreduce_ms", C.c_double),
        ("div_by_zero_warnings", C.c_int64),
        ("packed_lds_bits", C.c_int32), ("keyrec_digests", C.c_int32), ("side_stream_batches", C.c_int32), ("reserved0", C.c_int32),
    ]


ROUTE_DIRECT, ROUTE_RADIX_L2, ROUTE_RADIX_LDS, ROUTE_PACKED, ROUTE_KEYREC = 0, 1, 2, 3, 4


COMM_ID_BYTES = 128
KEYMODE_JOIN, KEYMODE_GROUP, KEYMODE_BROADCAST = 0, 1, 2

# tsq_ctx_set_knob (test / measurement knobs, include/tsq.h)
KNOB_DEFAULT = -(1 << 63)
(KNOB_PACKED_KEYS, KNOB_DA_MIN_BUILD_ROWS, KNOB_DA_PBITS, KNOB_PACKED_EMIT_PAIRS, KNOB_RADIX_KERNEL_L2, KNOB_LDS_NF_MAX, KNOB_RADIX_PB_MAX,
 KNOB_TABLE_LF_PERMILLE, KNOB_LDS_PROF, KNOB_DA_TRACE, KNOB_BUILD_IMAGES_CAS, KNOB_DAAGG_SIG, KNOB_DAAGG_LOG2C, KNOB_AGG_HEAP_GC_BYTES,
 KNOB_AGG_TAG_BITS, KNOB_AGG_BATCH_ROWS, KNOB_ROWCODEC_LDS_KB, KNOB_ROWCODEC_FAST_LAYOUT, KNOB_ROWCODEC_PIPELINE, KNOB_DA_PARTITION,
 KNOB_DA_NT_LOADS, KNOB_LAZY_TABLE, KNOB_DA_PAIRS_BELOW_PERMILLE, KNOB_AGG_WIDE_KEYS, KNOB_AGG_DENSE, KNOB_AGG_NARROW_CELLS, KNOB_DAAGG_PART2, KNOB_DAAGG_HOT, KNOB_KEYREC, KNOB_STREAMAGG_LANES, KNOB_XCD_ATOMICS, KNOB_DENSE_DIRECT, KNOB_DA_LDS_BUILD, KNOB_AGG_PG, KNOB_AGG_OVERLAP, KNOB_JIT_VARIANT, KNOB_HOST_OVERLAP, KNOB_HOST_NT_COPY, KNOB_KR_WG) = range(39)


# every symbol include/tsq.h declares: name -> (restype, argtypes)
P = C.c_void_p
PP = C.POINTER(C.c_void_p)
SIGNATURES = {
    "tsq_abi_version": (C.c_int32, []),
    "tsq_device_count": (C.c_int32, []),
    "tsq_last_error": (C.c_char_p, [P]),
    "tsq_ctx_create": (C.c_int32, [C.c_int32, PP]),
    "tsq_ctx_set_stream": (C.c_int32, [P, P]),
    "tsq_ctx_sync": (C.c_int32, [P]),
    "tsq_ctx_destroy": (None, [P]),
    "tsq_ctx_reserve": (C.c_int32, [P, C.c_int64]),
    "tsq_ctx_set_knob": (C.c_int32, [P, C.c_int32, C.c_int64]),
    "tsq_ctx_arena_stats": (C.c_int32, [P, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "tsq_dev_alloc": (C.c_int32, [P, C.c_int64, PP]),
    "tsq_dev_free": (C.c_int32, [P, P]),
    "tsq_host_alloc": (C.c_int32, [P, C.c_int64, PP]),
    "tsq_host_free": (C.c_int32, [P, P]),
    "tsq_dev_memset": (C.c_int32, [P, P, C.c_int32, C.c_int64]),
    "tsq_copy_h2d": (C.c_int32, [P, P, P, C.c_int64]),
    "tsq_copy_d2h": (C.c_int32, [P, P, P, C.c_int64]),
    "tsq_copy_d2d": (C.c_int32, [P, P, P, C.c_int64]),
    "tsq_bitmap_append": (C.c_int32, [P, P, C.c_int64, P, C.c_int64]),
    "tsq_timer_start": (C.c_int32, [P]),
    "tsq_timer_stop_ms": (C.c_int32, [P, C.POINTER(C.c_double)]),
    "tsq_gen_column": (C.c_int32, [P, C.POINTER(GenSpec), C.c_int64, P, P, P]),
    "tsq_expr_compile": (C.c_int32, [P, C.POINTER(ExprProg), C.c_int32, PP]),
    "tsq_expr_eval": (C.c_int32, [P, C.POINTER(Col), C.c_int32, C.c_int64, P, C.POINTER(Col), C.POINTER(C.c_int64)]),
    "tsq_expr_eval_str": (C.c_int32, [P, C.POINTER(Col), C.c_int32, C.c_int64, P, C.POINTER(Col), C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "tsq_filter_eval": (C.c_int32, [P, C.POINTER(Col), C.c_int32, C.c_int64, P, P, P, C.POINTER(C.c_int64)]),
    "tsq_expr_set_jit": (C.c_int32, [P, C.c_int32]),
    "tsq_expr_jit_launches": (C.c_int64, [P]),
    "tsq_expr_jit_compile_ms": (C.c_double, [P]),
    "tsq_expr_destroy": (None, [P]),
    "tsq_join_create": (C.c_int32, [P, C.POINTER(JoinCfg), PP]),
    "tsq_join_build_push": (C.c_int32, [P, C.POINTER(Col), C.c_int32, C.c_int64]),
    "tsq_join_build_finish_shared": (C.c_int32, [P, P, C.POINTER(C.c_int32)]),
    "tsq_join_set_used_columns": (C.c_int32, [P, P, C.c_int32]),
    "tsq_join_build_finish": (C.c_int32, [P]),
    "tsq_join_probe_push": (C.c_int32, [P, C.POINTER(Col), C.c_int32, C.c_int64, P]),
    "tsq_join_probe_finish": (C.c_int32, [P]),
    "tsq_join_pull": (C.c_int32, [P, C.POINTER(Col), C.c_int32, C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_int32)]),
    "tsq_join_set_count_only": (C.c_int32, [P, C.c_int32]),
    "tsq_join_count": (C.c_int32, [P, C.POINTER(C.c_int64)]),
    "tsq_join_checksum": (C.c_int32, [P, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "tsq_join_set_checksum": (C.c_int32, [P, C.c_int32]),
    "tsq_join_set_ordered": (C.c_int32, [P, C.c_int32]),
    "tsq_join_set_radix": (C.c_int32, [P, C.c_int32]),
    "tsq_join_set_key_packing": (C.c_int32, [P, C.c_int32]),
    "tsq_join_cancel": (C.c_int32, [P]),
    "tsq_join_destroy": (None, [P]),
    "tsq_agg_create": (C.c_int32, [P, C.POINTER(AggCfg), PP]),
    "tsq_agg_push": (C.c_int32, [P, C.POINTER(Col), C.c_int32, C.c_int64]),
    "tsq_agg_finish": (C.c_int32, [P]),
    "tsq_agg_set_fast": (C.c_int32, [P, C.c_int32]),
    "tsq_agg_set_stream": (C.c_int32, [P, C.c_int32]),
    "tsq_agg_num_groups": (C.c_int32, [P, C.POINTER(C.c_int64)]),
    "tsq_agg_pull": (C.c_int32, [P, C.POINTER(Col), C.c_int32, C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_int32)]),
    "tsq_agg_cancel": (C.c_int32, [P]),
    "tsq_agg_destroy": (None, [P]),
    "tsq_sort_create": (C.c_int32, [P, C.POINTER(SortCfg), PP]),
    "tsq_sort_push": (C.c_int32, [P, C.POINTER(Col), C.c_int32, C.c_int64]),
    "tsq_sort_finish": (C.c_int32, [P]),
    "tsq_sort_pull": (C.c_int32, [P, C.POINTER(Col), C.c_int32, C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_int32)]),
    "tsq_sort_peek": (C.c_int32, [P, C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_int32]),
    "tsq_sort_stats": (C.c_int32, [P, C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_double)]),
    "tsq_sort_cancel": (C.c_int32, [P]),
    "tsq_sort_destroy": (None, [P]),
    "tsq_chunk_compact": (C.c_int32, [P, C.POINTER(Col), C.c_int32, C.c_int64, P, C.POINTER(Col), C.POINTER(C.c_int64)]),
    "tsq_rows_decode": (C.c_int32, [P, P, C.c_int64, C.c_uint32, C.c_int32, C.POINTER(C.c_int32), C.POINTER(Col), C.c_int64, C.POINTER(C.c_int64),
                                    C.POINTER(C.c_int64)]),
    "tsq_rows_decode_chunks": (C.c_int32, [P, P, C.c_int64, P, C.c_int64, C.c_uint32, C.c_int32, C.POINTER(C.c_int32), C.POINTER(Col), C.c_int64,
                                           C.POINTER(C.c_int64)]),
    "tsq_indexkeys_decode": (C.c_int32, [P, P, C.c_int64, P, C.c_int64, P, C.c_int64, P, C.c_uint32, C.c_int32, C.POINTER(C.c_int32), C.c_int32, C.POINTER(Col),
                                         C.POINTER(C.c_int64)]),
    "tsq_chunk_encode": (C.c_int32, [P, C.POINTER(Col), C.c_int32, C.c_int64, P, C.c_int64, C.c_uint32, C.POINTER(C.c_int64)]),
    "tsq_chunk_decode_peek": (C.c_int32, [P, P, C.c_int64, C.c_uint32, C.POINTER(C.c_int32), C.c_int32, C.c_int64, C.c_int64, C.POINTER(C.c_int64),
                                          C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "tsq_chunk_decode": (C.c_int32, [P, P, C.c_int64, C.c_uint32, C.POINTER(C.c_int32), C.c_int32, C.c_int64, C.c_int64, C.POINTER(Col),
                                     C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "tsq_rowcodec_decode": (C.c_int32, [P, P, C.c_int64, P, P, C.c_int64, C.c_uint32, C.c_int32, C.POINTER(RowcodecCol), C.POINTER(Col),
                                        C.POINTER(C.c_int64)]),
    "tsq_rows_encode": (C.c_int32, [P, C.POINTER(Col), C.c_int32, C.POINTER(C.c_uint32), C.c_int64, P, C.c_int64, C.c_uint32, P, C.POINTER(C.c_int64)]),
    "tsq_rowkeys_decode": (C.c_int32, [P, P, C.c_int64, P, C.c_int64, C.c_uint32, P, P, C.POINTER(C.c_int64)]),
    "tsq_rowkeys_encode": (C.c_int32, [P, C.c_int64, P, C.c_int64, C.c_uint32, P]),
    "tsq_radix_split": (C.c_int32, [P, C.POINTER(Col), C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.c_int32,
                                    C.POINTER(Col), C.POINTER(C.c_int64)]),
    "tsq_comm_unique_id": (C.c_int32, [P]),
    "tsq_comm_create": (C.c_int32, [P, C.c_int32, C.c_int32, P, PP]),
    "tsq_comm_destroy": (None, [P]),
    "tsq_comm_allreduce_i64": (C.c_int32, [P, C.POINTER(C.c_int64), C.c_int32, C.c_int32]),
    "tsq_comm_allreduce_f64": (C.c_int32, [P, C.POINTER(C.c_double), C.c_int32, C.c_int32]),
    "tsq_comm_info": (C.c_int32, [P, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "tsq_comm_barrier": (C.c_int32, [P]),
    "tsq_redistribute": (C.c_int32, [P, C.POINTER(Col), C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.c_int32, C.POINTER(Col), C.POINTER(C.c_int64)]),
    "tsq_redistribute_wait": (C.c_int32, [P, C.c_int32]),
    "tsq_redistribute_prepare": (C.c_int32, [P, C.POINTER(Col), C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.c_int32]),
    "tsq_redistribute_counts": (C.c_int32, [P, C.POINTER(C.c_int32), C.c_int32]),
    "tsq_redistribute_issue": (C.c_int32, [P, C.c_int32, C.POINTER(Col), C.c_int32, C.POINTER(C.c_int64)]),
    "tsq_join_peek": (C.c_int32, [P, C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_int32]),
    "tsq_agg_peek": (C.c_int32, [P, C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_int32]),
    "tsq_join_stats": (C.c_int32, [P, C.POINTER(Stats)]),
    "tsq_agg_stats": (C.c_int32, [P, C.POINTER(Stats)]),
}

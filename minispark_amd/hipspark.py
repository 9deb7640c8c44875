"""ctypes binding of libhipspark.so (the C ABI declared in include/hipspark.h).

Thin by design: structures mirror the header field for field, every function gets its argtypes, and
``check()`` turns a nonzero return into :class:`HipSparkError`.  There is no fallback: if the shared
object cannot be built or loaded, importing the engine fails loudly.
"""

from __future__ import annotations

import ctypes as C
from pathlib import Path

HS_MAX_INS = 96
HS_MAX_LIT = 32
HS_MAX_POOL = 256
HS_MAX_COLS = 12
HS_MAX_ACC = 16
HS_MAX_STACK = 8
HS_MAX_OUTS = 16
HS_MAX_PARTS = 8
HS_FUSED_COLS = 8

# storage kinds
I32, F32, I64, F64, STR, U8 = 0, 1, 2, 3, 4, 5
JOIN8_CODE, JOIN8_UNIT = 16, 17  # virtual columns of the fused join + aggregate (hs_agg_shared_join8)
JOIN8_WINDOW = 65536
KIND_BYTES = {I32: 4, F32: 4, I64: 8, F64: 8, U8: 1}

# flags
FLAG_DIV_ZERO = 0x1
FLAG_INT_OVERFLOW = 0x2
FLAG_FLT_OVERFLOW = 0x4
FLAG_DICT_FULL = 0x8
FLAG_MERGE_FULL = 0x100
FLAG_MERGE_ROWS = 0x200
FLAG_BAD_PROGRAM = 0x10
FLAG_STR_TOO_LONG = 0x20
FLAG_TYPE_ASSERT = 0x40
FLAG_JOIN_DUP = 0x80
FLAG_PEER_TIMEOUT = 0x400
FLAG_ROUTE_STALE = 0x800
FLAG_KNOWN = 0xFFF  # every HS_FLAG_* bit include/hipspark.h defines

AGG_SUM, AGG_MIN, AGG_MAX = 0, 1, 2

# opcodes
OP_END, OP_LD, OP_LIT = 0, 1, 2
OP_ADD_F, OP_SUB_F, OP_MUL_F, OP_DIV_F, OP_FLOORDIV_F, OP_MOD_F = 3, 4, 5, 6, 7, 8
OP_ADD_I, OP_SUB_I, OP_MUL_I, OP_FLOORDIV_I, OP_MOD_I = 9, 10, 11, 12, 13
OP_LT_F, OP_LE_F, OP_GT_F, OP_GE_F, OP_EQ_F, OP_NE_F = 14, 15, 16, 17, 18, 19
OP_LT_I, OP_LE_I, OP_GT_I, OP_GE_I, OP_EQ_I, OP_NE_I = 20, 21, 22, 23, 24, 25
OP_AND, OP_OR, OP_I2F = 26, 27, 28
OP_STRCMP_LIT, OP_STRCMP_COL, OP_LIKE = 29, 30, 31
OP_FILTER, OP_AGG, OP_OUT, OP_KEY, OP_DICTBIT = 32, 33, 34, 35, 36


class HipSparkError(RuntimeError):
    pass


class HipSparkLimit(HipSparkError):
    """An entry point returned HS_E_LIMIT: the call exceeds one of its documented limits (the caller may take another
    path; nothing ran wrong)."""


class hs_col(C.Structure):
    _fields_ = [
        ("kind", C.c_int32),
        ("fixed_len", C.c_int32),
        ("data", C.c_void_p),
        ("lens", C.c_void_p),
        ("offs", C.c_void_p),
    ]


class hs_program(C.Structure):
    _fields_ = [
        ("n_ins", C.c_uint32),
        ("n_lit", C.c_uint32),
        ("ins", C.c_uint64 * HS_MAX_INS),
        ("lit", C.c_uint64 * HS_MAX_LIT),
        ("pool", C.c_uint8 * HS_MAX_POOL),
    ]


class hs_agg_spec(C.Structure):
    _fields_ = [
        ("n_acc", C.c_int32),
        ("op", C.c_uint8 * HS_MAX_ACC),
        ("is_int", C.c_uint8 * HS_MAX_ACC),
    ]


class hs_join8(C.Structure):
    _fields_ = [("table", C.c_void_p), ("slots", C.c_int64), ("key_min", C.c_int32), ("n_parts", C.c_int32)]


class hs_span(C.Structure):
    _fields_ = [("file_offset", C.c_int64), ("bytes", C.c_int64), ("dst", C.c_void_p)]


class hs_segment(C.Structure):
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("bytes", C.c_int64)]


class hs_chunk(C.Structure):
    _fields_ = [("row_begin", C.c_int64), ("row_end", C.c_int64), ("unit_begin", C.c_int64), ("unit", C.c_int64)]


class hs_agg_geom(C.Structure):
    _fields_ = [
        ("group_cap", C.c_int32),
        ("chunk_rows", C.c_int32),
        ("wg_threads", C.c_int32),
        ("pad", C.c_int32),
        ("n_chunks", C.c_int64),
        ("lds_bytes", C.c_size_t),
        ("ws_bytes", C.c_size_t),
    ]


HS_FINISH_MAX_OUT = 24


class hs_slab_desc(C.Structure):
    _fields_ = [
        ("slab_rows", C.c_int64),
        ("stride", C.c_int64),
        ("order_off", C.c_int64),
        ("key_off", C.c_int64),
        ("acc_off", C.c_int64 * HS_MAX_ACC),
        ("key_kind", C.c_int32),
        ("key_len", C.c_int32),
        ("n_acc", C.c_int32),
        ("pad", C.c_int32),
        ("acc_kind", C.c_int32 * HS_MAX_ACC),
    ]


class hs_finish_out(C.Structure):
    _fields_ = [("src", C.c_int32), ("index", C.c_int32), ("kind", C.c_int32), ("pad", C.c_int32), ("offset", C.c_int64)]


class hs_finish_spec(C.Structure):
    _fields_ = [
        ("n_fold", C.c_int32),
        ("fold_src", C.c_int32 * HS_MAX_ACC),
        ("fold_op", C.c_int32 * HS_MAX_ACC),
        ("n_out", C.c_int32),
        ("outs", hs_finish_out * HS_FINISH_MAX_OUT),
        ("prog_src", C.c_int32 * HS_MAX_COLS),
        ("prog_out", C.c_int32 * HS_MAX_OUTS),
    ]


class hs_stage_plan(C.Structure):
    _fields_ = [
        ("version", C.c_int32),
        ("n_cols", C.c_int32),
        ("col_ids", C.c_int32 * HS_MAX_COLS),
        ("key_slot", C.c_int32),
        ("group_cap", C.c_int32),
        ("merge_cap", C.c_int32),
        ("prog", hs_program),
        ("spec", hs_agg_spec),
        ("fin", hs_finish_spec),
        ("fin_prog", hs_program),
        ("out_types", C.c_int32 * HS_FINISH_MAX_OUT),
        ("out_names", (C.c_char * 64) * HS_FINISH_MAX_OUT),
        ("key_computed", C.c_int32),
        ("n_kcols", C.c_int32),
        ("kcol_ids", C.c_int32 * HS_MAX_COLS),
        ("key_prog", hs_program),
    ]


HS_JOIN_STAGE_PLAN_VERSION = 1


class hs_join_stage_plan(C.Structure):
    _fields_ = [
        ("version", C.c_int32),
        ("build_key_col", C.c_int32),
        ("build_payload_col", C.c_int32),
        ("probe_key_col", C.c_int32),
        ("n_parts", C.c_int32),
        ("n_cols", C.c_int32),
        ("col_ids", C.c_int32 * HS_MAX_COLS),
        ("key_slot", C.c_int32),
        ("group_cap", C.c_int32),
        ("merge_cap", C.c_int32),
        ("prog", hs_program),
        ("spec", hs_agg_spec),
        ("fin", hs_finish_spec),
        ("fin_prog", hs_program),
        ("out_types", C.c_int32 * HS_FINISH_MAX_OUT),
        ("out_names", (C.c_char * 64) * HS_FINISH_MAX_OUT),
    ]


HS_SELECT_STAGE_PLAN_VERSION = 1


class hs_select_stage_plan(C.Structure):
    _fields_ = [
        ("version", C.c_int32),
        ("n_cols", C.c_int32),
        ("col_ids", C.c_int32 * HS_MAX_COLS),
        ("filter", hs_program),
        ("n_pcols", C.c_int32),
        ("pcol_ids", C.c_int32 * HS_MAX_COLS),
        ("project", hs_program),
        ("project_kinds", C.c_int32 * HS_MAX_OUTS),
        ("n_out", C.c_int32),
        ("out_src", C.c_int32 * HS_FINISH_MAX_OUT),
        ("out_types", C.c_int32 * HS_FINISH_MAX_OUT),
        ("out_names", (C.c_char * 64) * HS_FINISH_MAX_OUT),
    ]


class hs_trace_slice(C.Structure):
    _fields_ = [("name", C.c_char * 96), ("start_us", C.c_double), ("dur_us", C.c_double)]


class hs_result_col(C.Structure):
    _fields_ = [("kind", C.c_int32), ("width", C.c_int32), ("data", C.c_void_p), ("n_rows", C.c_int64)]


class hs_radix_plan(C.Structure):
    _fields_ = [("f", C.c_int64 * 48)]


HS_STAGE_PLAN_VERSION = 2

_P = C.c_void_p
_I64 = C.c_int64
_I32 = C.c_int32
_COLP = C.POINTER(hs_col)
_PROGP = C.POINTER(hs_program)
_SPECP = C.POINTER(hs_agg_spec)
_GEOMP = C.POINTER(hs_agg_geom)

# name -> (restype, argtypes); every symbol include/hipspark.h declares
SIGNATURES: dict[str, tuple] = {
    "hs_last_error": (C.c_char_p, []),
    "hs_version": (C.c_int, []),
    "hs_sizeof": (C.c_size_t, [_I32]),
    "hs_scan_ws_bytes": (C.c_size_t, [_I64]),
    "hs_str_offsets": (C.c_int, [_P, _P, _I64, _P, _P, _P]),
    "hs_eval": (C.c_int, [_P, _COLP, _I32, _PROGP, _P, _I64, _P, C.POINTER(_P), C.POINTER(_I32), _I32, _P]),
    "hs_compact": (C.c_int, [_P, _P, _I64, _P, _P, _P]),
    "hs_gather_fixed": (C.c_int, [_P, _P, _I32, _I64, _P, _I64, _P, _P, _P]),
    "hs_gather_str_lens": (C.c_int, [_P, _COLP, _I64, _P, _I64, _P, _P]),
    "hs_gather_str_bytes": (C.c_int, [_P, _COLP, _I64, _P, _I64, _P, _P]),
    "hs_concat_lens": (C.c_int, [_P, _COLP, _I32, _I64, _P, _P]),
    "hs_concat_bytes": (C.c_int, [_P, _COLP, _I32, _I64, _P, _P]),
    "hs_agg_partial_geom": (C.c_int, [C.POINTER(_I64), _I64, _I32, _I32, _GEOMP]),
    "hs_agg_partial_chunks": (C.c_int, [C.POINTER(_I64), _I64, _GEOMP, C.POINTER(hs_chunk), C.POINTER(_I64)]),
    "hs_agg_partial": (
        C.c_int,
        [_P, _COLP, _I32, _I32, _PROGP, _SPECP, _P, _P, _I64, _GEOMP, _P, _P, _P, _P, _P, _P, _P],
    ),
    "hs_agg_shared_geom": (C.c_int, [C.POINTER(_I64), _I64, _I32, _I32, _GEOMP]),
    "hs_agg_shared": (C.c_int, [_P, _COLP, _I32, _I32, _PROGP, _SPECP, _P, _I64, _GEOMP, _P, _P, _P, _P, _P, _P, _P]),
    "hs_agg_shared_units": (C.c_int, [_P, _COLP, _I32, _I32, _I32, _I32, _PROGP, _SPECP, _P, _GEOMP, _P, _P, _P, _P, _P,
                                      _P, _P]),
    "hs_agg_shared_join8": (C.c_int, [_P, _COLP, _I32, _I32, _I32, C.POINTER(hs_join8), _I32, _PROGP, _SPECP, _P, _GEOMP,
                                      _P, _P, _P, _P, _P, _P, _P]),
    "hs_agg_units_merge": (C.c_int, [_P, _P, _I32, _I32, _I32, _SPECP, _P, _P, _P]),
    "hs_agg_units_to_slab": (C.c_int, [_P, _P, _P, _I32, _I32, _SPECP, _P, C.POINTER(hs_slab_desc), _P]),
    "hs_agg_partial_slab": (
        C.c_int,
        [_P, _COLP, _I32, _I32, _PROGP, _SPECP, _P, _P, _I64, _GEOMP, _P, _P, C.POINTER(hs_slab_desc), _P, _P, _P, _P],
    ),
    "hs_agg_finish_scratch_bytes": (C.c_size_t, [_I32, _I32]),
    "hs_agg_finish": (
        C.c_int,
        [_P, _P, _I32, C.POINTER(hs_slab_desc), C.POINTER(hs_finish_spec), _PROGP, _I64, _I32, _P, _P, _P, _P],
    ),
    "hs_host_device_pointer": (C.c_int, [_P, C.POINTER(_P)]),
    "hs_agg_pack": (C.c_int, [_P, _P, _P, _P, _I64, _I32, _SPECP, _P, _P, C.POINTER(_P), C.POINTER(_I32), _P, _P]),
    "hs_agg_merge": (C.c_int, [_P, _COLP, _COLP, _SPECP, _P, _I64, _I64, _P, _I32, _P, _P, _P, _P]),
    "hs_partition_ids": (C.c_int, [_P, _COLP, _P, _I64, _I32, _P]),
    "hs_partition_ws_bytes": (C.c_size_t, [_I64, _I32]),
    "hs_partition_perm": (C.c_int, [_P, _P, _I64, _I32, _P, _P, _P]),
    "hs_join_build_ws_bytes": (C.c_size_t, [_I64, _I64]),
    "hs_join_build": (C.c_int, [_P, _COLP, _I64, _I64, _P, _P, _P, _P, _P, _P]),
    "hs_group_build": (C.c_int, [_P, _COLP, _P, _I64, _I64, _I64, _P, _P, _P, _P, _P, _P]),
    "hs_group_build_units": (C.c_int, [_P, _COLP, _P, _I64, _I64, _P, _P, _I32, _I64, _P, _P, _P, _P, _P, _P]),
    "hs_lower_bound_i64": (C.c_int, [_P, _P, _I64, _P, _P, _I64, _P]),
    "hs_group_radix_plan": (C.c_int, [_I32, _I64, _I32, _I64, _P, _SPECP, _I32, _P]),
    "hs_group_radix_ws_bytes": (C.c_size_t, [_P]),
    "hs_group_radix_run": (C.c_int, [_P, _P, _COLP, _P, _I64, _P, _COLP, _P, _SPECP, _P, _P, _P]),
    "hs_group_radix_emit": (C.c_int, [_P, _P, _P, _P, _P]),
    "hs_group_radix_debug_stamps": (C.c_int, [_P]),
    "hs_agg_debug_scan_stamps": (C.c_int64, [_P, C.c_int64]),
    "hs_sort_by_order_ws_bytes": (C.c_size_t, [_I64]),
    "hs_sort_by_order": (C.c_int, [_P, _P, _I64, _I64, _P, _P, _P]),
    "hs_expand_by_bounds": (C.c_int, [_P, _P, _P, _I64, _I64, _P]),
    "hs_group_mask": (C.c_int, [_P, _P, _I64, _P]),
    "hs_group_fold": (C.c_int, [_P, _COLP, _SPECP, _P, _I64, _P, _P, _P, _P, _I64, _I32, _P, _P, _P]),
    "hs_join_count": (C.c_int, [_P, _COLP, _COLP, _I64, _I64, _P, _P, _P, _P]),
    "hs_join_fill": (C.c_int, [_P, _COLP, _COLP, _I64, _I64, _P, _P, _P, _P, _P, _P, _P]),
    "hs_exclusive_scan_i64": (C.c_int, [_P, _P, _I64, _P, _P]),
    "hs_minmax_i32": (C.c_int, [_P, _P, _I64, _P]),
    "hs_join_build_unique": (C.c_int, [_P, _P, _I64, _I32, _I64, _I32, _P, _P]),
    "hs_join_probe_unique": (C.c_int, [_P, _P, _I64, _P, _I32, _I64, _I32, _P, _I32, _P, _P, _P, _P]),
    "hs_join8_table_bytes": (C.c_size_t, [_I64]),
    "hs_join8_ws_bytes": (C.c_size_t, [_I64, _I64]),
    "hs_join8_build": (C.c_int, [_P, _P, _P, _I64, _I64, _P, _I32, _I64, _P, _P, _P]),
    "hs_minmax_i32_units": (C.c_int, [_P, _P, _P, _I64, _P]),
    "hs_join8_route_ws_bytes": (C.c_size_t, [_I64, _I32]),
    "hs_join8_route_count": (C.c_int, [_P, _P, _I64, _P, _P, _P, _I32, _I32, _P, _P]),
    "hs_join8_route": (C.c_int, [_P, _P, _P, _I64, _P, _P, _P, _I32, _I32, _P, _P, _P, _P, _P, _I64, _P]),
    "hs_join8_build_windows": (C.c_int, [_P, _P, _P, _I64, _I32, _I64, _P, _P, _P, _P]),
    "hs_join_dense_ws_bytes": (C.c_size_t, [_I64, _I64]),
    "hs_join_dense_build": (C.c_int, [_P, _P, _I64, _I32, _I64, _P, _P, _P, _P, _P]),
    "hs_join_dense_aux_bytes": (C.c_size_t, [_I64]),
    "hs_join_dense_count": (C.c_int, [_P, _P, _I64, _I32, _I64, _P, _P, _P, _P, _P]),
    "hs_join_dense_fill": (C.c_int, [_P, _I64, _P, _P, _P, _P, _P]),
    "hs_join_hash_ws_bytes": (C.c_size_t, [_I64]),
    "hs_join_hash_slots": (C.c_int64, [_I64]),
    "hs_join_hash_build": (C.c_int, [_P, _P, _I64, _P, _P, _P, _P, _P]),
    "hs_join_hash_count": (C.c_int, [_P, _P, _I64, _I64, _P, _P, _P, _P, _P]),
    "hs_remap_u8": (C.c_int, [_P, _P, _I64, _P, _P]),
    "hs_dict_build": (C.c_int, [_P, _COLP, _I64, _I32, _P, _P, _P, _P]),
    "hs_dict_assign": (C.c_int, [_P, _COLP, _I64, _I32, _P, _P, _P, _P, _P]),
    "hs_dict_combine": (C.c_int, [_P, _I32, C.POINTER(_P), C.POINTER(_I32), _I64, _P]),
    "hs_quantise": (C.c_int, [_P, _P, _I32, _I64, _P, _P, _P]),
    "hs_quantise_many": (C.c_int, [_P, _I32, C.POINTER(_P), C.POINTER(_I32), _I64, _P, C.POINTER(_P), _P]),
    "hs_slab_unpack": (C.c_int, [_P, _P, _I32, _I64, _I64, _I64, _I32, C.POINTER(_I64), C.POINTER(_I32),
                                 C.POINTER(_P), _P, _P]),
    "hs_slab_p2p_bytes": (C.c_size_t, [_I32, _I64]),
    "hs_slab_push": (C.c_int, [_P, _P, _I64, _P, _I32, _I32, _I64, _P]),
    "hs_slab_wait": (C.c_int, [_P, _P, _I32, _I64, _I64, _P, _P, _I64, _P, _I64]),
    "hs_copy_segments": (C.c_int, [_P, _P, _I32, _I64]),
    "hs_jit_set_enabled": (None, [C.c_int]),
    "hs_jit_get_enabled": (C.c_int, []),
    "hs_jit_stats": (None, [C.POINTER(_I32)]),
    "hs_jit_disk_hits": (C.c_int, []),
    "hs_jit_compile_seconds": (C.c_double, []),
    "hs_jit_last_log": (C.c_char_p, []),
    "hs_jit_compile_check": (C.c_int, [_COLP, _I32, _I32, _PROGP, _SPECP, C.c_char_p, C.POINTER(_I64), C.c_char_p, _I64]),
    "hs_jit_compile_check_shared": (C.c_int, [_COLP, _I32, _I32, _I32, _PROGP, _SPECP, C.c_char_p, C.POINTER(_I64),
                                              C.c_char_p, _I64]),
    "hs_jit_compile_check_eval": (C.c_int, [_COLP, _I32, _PROGP, C.POINTER(_I32), _I32, C.c_char_p, C.POINTER(_I64),
                                            C.c_char_p, _I64]),
    "hs_engine_create": (C.c_int, [_I32, C.POINTER(_P)]),
    "hs_engine_destroy": (None, [_P]),
    "hs_read_spans": (C.c_int, [_P, C.c_char_p, C.POINTER(hs_span), _I32]),
    "hs_engine_load_stats": (C.c_int, [_P, C.POINTER(C.c_double), C.POINTER(_I64)]),
    "hs_table_open": (C.c_int, [_P, C.c_char_p, _I32, _I32, C.POINTER(_P)]),
    "hs_table_close": (None, [_P]),
    "hs_table_info": (C.c_int, [_P, C.POINTER(_I32), C.POINTER(_I64), C.POINTER(_I32), C.POINTER(_I32)]),
    "hs_table_schema": (C.c_int, [_P, _I32, C.POINTER(_I32), C.c_char_p, _I32]),
    "hs_table_load": (C.c_int, [_P, _P, C.POINTER(_I32), _I32]),
    "hs_table_column": (C.c_int, [_P, _I32, _COLP, C.POINTER(_I64)]),
    "hs_table_attach": (C.c_int, [_P, _I32, _COLP, C.POINTER(_I32), C.POINTER(_I64), _I32, C.POINTER(_P)]),
    "hs_select_stage_prepare": (C.c_int, [_P, _P, C.POINTER(hs_select_stage_plan), C.c_size_t, C.POINTER(_P)]),
    "hs_select_stage_run": (C.c_int, [_P, _P, C.POINTER(C.c_uint32), C.POINTER(_I64)]),
    "hs_select_result_write_blockfile": (C.c_int, [_P, C.c_char_p, _I64]),
    "hs_select_stage_destroy": (None, [_P]),
    "hs_join_stage_prepare": (C.c_int, [_P, _P, _P, C.POINTER(hs_join_stage_plan), C.c_size_t, C.POINTER(_P)]),
    "hs_join_stage_run": (C.c_int, [_P, _P, C.POINTER(C.c_uint32), C.POINTER(_I64)]),
    "hs_join_stage_stats": (C.c_int, [_P, C.POINTER(_I64)]),
    "hs_join_result_write_blockfile": (C.c_int, [_P, C.c_char_p]),
    "hs_join_stage_destroy": (None, [_P]),
    "hs_stage_prepare": (C.c_int, [_P, _P, C.POINTER(hs_stage_plan), C.c_size_t, _I32, C.POINTER(_P)]),
    "hs_stage_destroy": (None, [_P]),
    "hs_stage_run": (C.c_int, [_P, _P, C.POINTER(C.c_uint32), C.POINTER(_I64)]),
    "hs_stage_launch_partial": (C.c_int, [_P, _P]),
    "hs_stage_slab": (_P, [_P, C.POINTER(_I64)]),
    "hs_stage_launch_finish": (C.c_int, [_P, _P, _P, _I32]),
    "hs_stage_wait": (C.c_int, [_P, _P, C.POINTER(C.c_uint32), C.POINTER(_I64)]),
    "hs_stage_grow": (C.c_int, [_P]),
    "hs_stage_stats": (C.c_int, [_P, C.POINTER(_I64)]),
    "hs_result_columns": (C.c_int, [_P, C.POINTER(hs_result_col), _I32, C.POINTER(_I32)]),
    "hs_result_write_blockfile": (C.c_int, [_P, C.c_char_p]),
    "hs_trace_begin": (C.c_int, [_P]),
    "hs_trace_end": (C.c_int, [_P, C.POINTER(hs_trace_slice), _I32, C.POINTER(_I32)]),
    "hs_capture_begin": (C.c_int, []),
    "hs_capture_end": (C.c_int, [C.POINTER(_P), C.POINTER(_I32)]),
    "hs_capture_replay": (C.c_int, [_P, _P]),
    "hs_capture_free": (None, [_P]),
    "hs_gen_lineitem": (C.c_int, [_P, C.c_uint64, _I64, _I64, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "hs_gen_orders": (C.c_int, [_P, C.c_uint64, _I64, _I64, _I64, _P, _P]),
}

_lib: C.CDLL | None = None


def library_path() -> Path:
    return Path(__file__).resolve().parent / "libhipspark.so"


def load_library(build_if_missing: bool = True) -> C.CDLL:
    """Load (building first if needed) libhipspark.so and bind every declared symbol."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if build_if_missing:
        from . import _build  # noqa: PLC0415

        path = _build.build()
    if not path.exists():
        raise HipSparkError(f"{path} is missing: run `python -m minispark_amd._build` (needs hipcc)")
    # torch first: it brings its own HIP runtime (libamdhip64) and owns the device / stream state this library
    # works on.  Loaded the other way round, the library would pull in the system's runtime and the two would not
    # see the same devices ("no ROCm-capable device is detected" at the first launch).
    import torch  # noqa: F401, PLC0415

    lib = C.CDLL(str(path))
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here = the library does not export a declared symbol
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load_library().hs_last_error()
        raise (HipSparkLimit if rc == 2 else HipSparkError)(f"{what or 'libhipspark'} failed (code {rc}): {msg.decode() if msg else ''}")

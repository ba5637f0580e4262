"""ctypes binding of libydb_ssa_hip.so (see include/ydb_ssa_hip.h).

The library is loaded from the in-tree build directory only; importing this module fails loudly
if it has not been built -- there is no CPU fallback for the product path.
"""
from __future__ import annotations

import ctypes
import os

from .arrow_c import ArrowArray, ArrowSchema

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libydb_ssa_hip.so")

SSA_OK, SSA_INVALID, SSA_PARSE_ERROR, SSA_UNSUPPORTED, SSA_HIP_ERROR = range(5)
STATUS_NAMES = {0: "OK", 1: "INVALID", 2: "PARSE_ERROR", 3: "UNSUPPORTED", 4: "HIP_ERROR"}


class DeviceColumn(ctypes.Structure):
    _fields_ = [
        ("name", ctypes.c_char_p),
        ("type", ctypes.c_int32),
        ("index_type", ctypes.c_int32),
        ("values", ctypes.c_void_p),
        ("validity", ctypes.c_void_p),
        ("chars", ctypes.c_void_p),
        ("chars_size", ctypes.c_int64),
        ("null_count", ctypes.c_int64),
        ("dict_offsets", ctypes.c_void_p),
        ("dict_size", ctypes.c_int64),
    ]


class LineitemColumns(ctypes.Structure):
    _fields_ = [(n, ctypes.c_void_p) for n in (
        "l_shipdate", "l_quantity", "l_extendedprice", "l_discount", "l_tax",
        "rf_offsets", "rf_chars", "ls_offsets", "ls_chars", "rf_code", "ls_code")]


# Every symbol include/ydb_ssa_hip.h declares.
EXPORTED_SYMBOLS = [
    "ssa_device_count", "ssa_version", "ssa_shutdown", "ssa_free_string", "ssa_program_create", "ssa_program_create_with_kernels", "ssa_group_by_create", "ssa_program_destroy",
    "ssa_program_debug_string", "ssa_program_source_columns", "ssa_program_early_filter_columns",
    "ssa_program_step_count", "ssa_program_explain", "ssa_program_apply", "ssa_program_build_filter", "ssa_batch_import",
    "ssa_batch_from_device", "ssa_dictionary_stats", "ssa_program_apply_device", "ssa_program_apply_many", "ssa_apply_many_stats", "ssa_keyed_stats", "ssa_keyed_sort_stats", "ssa_keyed_compact_stats",
    "ssa_batch_pack_layout", "ssa_batch_pack_data", "ssa_batch_unpack", "ssa_batch_concat", "ssa_batch_slice", "ssa_batch_rename",
    "ssa_batch_column_name", "ssa_merge_sorted", "ssa_batch_export", "ssa_batch_rows",
    "ssa_batch_columns", "ssa_batch_destroy", "ssa_sharding_hash", "ssa_sharding_split", "ssa_batch_partition_by_key", "ssa_profile_enable", "ssa_profile_read", "ssa_specialization_stats",
    "ssa_filter_simple", "ssa_filter_from_runs", "ssa_filter_from_program", "ssa_filter_predicate", "ssa_filter_and", "ssa_filter_or",
    "ssa_filter_combine_sequential_and", "ssa_filter_info", "ssa_filter_runs", "ssa_filter_bits", "ssa_filter_filtered_count",
    "ssa_filter_apply", "ssa_filter_destroy", "ssa_filter_from_program_noalloc",
    "ssa_batch_from_blob", "ssa_batches_from_blobs", "ssa_blob_describe",
]

# ... and include/ydb_ssa_bench.h (libydb_ssa_bench.so: benchmark and test support, not part of the executor)
BENCH_LIB_PATH = os.path.join(_HERE, "lib", "libydb_ssa_bench.so")
BENCH_EXPORTED_SYMBOLS = ["ssa_gen_lineitem"]

_lib = None
_bench = None


class GroupByAssign(ctypes.Structure):
    """struct ssa_group_by_assign (CH::GroupByOptions::Assign)."""
    _fields_ = [("function", ctypes.c_int32), ("result_column", ctypes.c_char_p), ("argument", ctypes.c_char_p)]


def load_bench() -> ctypes.CDLL:
    """The benchmark-support library (synthetic lineitem generated in HBM)."""
    global _bench
    if _bench is None:
        load()  # (torch's HIP runtime first, as below)
        if not os.path.exists(BENCH_LIB_PATH):
            raise ImportError(f"{BENCH_LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'`")
        _bench = ctypes.CDLL(BENCH_LIB_PATH)
        _bench.ssa_gen_lineitem.argtypes = [ctypes.POINTER(LineitemColumns), ctypes.c_int64, ctypes.c_int64, ctypes.c_uint64, ctypes.c_int]
        _bench.ssa_gen_lineitem.restype = ctypes.c_int
    return _bench


def load() -> ctypes.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  The executor has no CPU fallback.")
    # One HIP runtime per process: PyTorch-ROCm ships its own libamdhip64 and cannot initialise after the system one
    # has been loaded by this library, so when torch is installed (the tests and the benchmark use it for device memory)
    # it is imported first and both share the runtime it brought.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = ctypes.CDLL(LIB_PATH)
    P = ctypes.POINTER
    vp, cp, i64, i32, sz = ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int64, ctypes.c_int, ctypes.c_size_t
    lib.ssa_device_count.restype = i32
    lib.ssa_version.restype = cp
    lib.ssa_shutdown.argtypes = []
    lib.ssa_shutdown.restype = None
    import atexit
    atexit.register(lib.ssa_shutdown)  # no kernel build may be running while the process unloads its libraries
    lib.ssa_free_string.argtypes = [vp]
    lib.ssa_program_create.argtypes = [vp, sz, i32, P(ctypes.c_uint32), P(cp), sz, P(ArrowArray), P(ArrowSchema), i32,
                                       P(vp), P(vp)]
    lib.ssa_program_create.restype = i32
    lib.ssa_program_create_with_kernels.argtypes = [vp, sz, i32, P(ctypes.c_uint32), P(cp), sz, P(ArrowArray), P(ArrowSchema),
                                                    P(ctypes.c_uint32), sz, i32, P(vp), P(vp)]
    lib.ssa_program_create_with_kernels.restype = i32
    lib.ssa_group_by_create.argtypes = [P(GroupByAssign), sz, i32, P(vp), P(vp)]
    lib.ssa_group_by_create.restype = i32
    lib.ssa_program_destroy.argtypes = [vp]
    for f in (lib.ssa_program_debug_string, lib.ssa_program_source_columns, lib.ssa_program_early_filter_columns):
        f.argtypes = [vp]
        f.restype = vp
    lib.ssa_program_step_count.argtypes = [vp]
    lib.ssa_program_step_count.restype = sz
    lib.ssa_program_explain.argtypes = [vp, P(ArrowSchema), P(vp)]
    lib.ssa_program_explain.restype = vp
    lib.ssa_program_apply.argtypes = [vp, P(ArrowArray), P(ArrowSchema), P(ArrowArray), P(ArrowSchema), P(vp)]
    lib.ssa_program_apply.restype = i32
    lib.ssa_program_build_filter.argtypes = [vp, sz, P(ArrowArray), P(ArrowSchema), vp, P(i64), P(vp)]
    lib.ssa_program_build_filter.restype = i32
    lib.ssa_batch_import.argtypes = [vp, P(ArrowArray), P(ArrowSchema), P(vp), P(vp)]
    lib.ssa_batch_import.restype = i32
    lib.ssa_batch_from_device.argtypes = [P(DeviceColumn), sz, i64, P(vp), P(vp)]
    lib.ssa_batch_from_device.restype = i32
    lib.ssa_dictionary_stats.argtypes = [P(i64), P(i64)]
    lib.ssa_dictionary_stats.restype = None
    lib.ssa_program_apply_many.argtypes = [vp, P(vp), sz, P(vp), P(vp)]
    lib.ssa_program_apply_many.restype = i32
    lib.ssa_batch_pack_layout.argtypes = [vp, vp, vp, sz, P(sz), P(sz), P(vp)]
    lib.ssa_batch_pack_layout.restype = i32
    lib.ssa_batch_pack_data.argtypes = [vp, vp, vp, sz, P(vp)]
    lib.ssa_batch_pack_data.restype = i32
    lib.ssa_batch_unpack.argtypes = [vp, sz, vp, sz, P(vp), P(vp)]
    lib.ssa_batch_unpack.restype = i32
    lib.ssa_batch_concat.argtypes = [vp, P(vp), sz, P(vp), P(vp)]
    lib.ssa_batch_concat.restype = i32
    lib.ssa_batch_slice.argtypes = [vp, i64, i64, P(vp), P(vp)]
    lib.ssa_batch_slice.restype = i32
    lib.ssa_batch_rename.argtypes = [vp, P(cp), sz, P(vp), P(vp)]
    lib.ssa_batch_rename.restype = i32
    lib.ssa_batch_column_name.argtypes = [vp, sz]
    lib.ssa_batch_column_name.restype = vp
    lib.ssa_merge_sorted.argtypes = [vp, P(vp), P(vp), sz, P(cp), sz, P(cp), sz, i32, P(vp), P(vp)]
    lib.ssa_merge_sorted.restype = i32
    lib.ssa_apply_many_stats.argtypes = [P(i64), P(i64)]
    lib.ssa_apply_many_stats.restype = None
    lib.ssa_keyed_stats.argtypes = [P(i64)]
    lib.ssa_keyed_stats.restype = None
    lib.ssa_keyed_sort_stats.argtypes = [P(i64)]
    lib.ssa_keyed_sort_stats.restype = None
    lib.ssa_keyed_compact_stats.argtypes = [P(i64)]
    lib.ssa_keyed_compact_stats.restype = None
    lib.ssa_program_apply_device.argtypes = [vp, vp, P(vp), P(vp)]
    lib.ssa_program_apply_device.restype = i32
    lib.ssa_batch_export.argtypes = [vp, vp, P(ArrowArray), P(ArrowSchema), P(vp)]
    lib.ssa_batch_export.restype = i32
    lib.ssa_batch_rows.argtypes = [vp]
    lib.ssa_batch_rows.restype = i64
    lib.ssa_batch_columns.argtypes = [vp]
    lib.ssa_batch_columns.restype = sz
    lib.ssa_batch_destroy.argtypes = [vp]
    lib.ssa_profile_enable.argtypes = [vp, i32]
    lib.ssa_profile_read.argtypes = [vp, P(ctypes.c_double), P(i64), P(i64)]
    lib.ssa_sharding_hash.argtypes = [vp, vp, P(ctypes.c_char_p), sz, ctypes.c_uint64, vp, P(vp)]
    lib.ssa_sharding_hash.restype = i32
    lib.ssa_sharding_split.argtypes = [vp, vp, P(ctypes.c_char_p), sz, ctypes.c_uint32, P(vp), vp, P(vp)]
    lib.ssa_sharding_split.restype = i32
    lib.ssa_batch_partition_by_key.argtypes = [vp, vp, P(ctypes.c_char_p), sz, ctypes.c_uint32, P(vp), vp, P(vp)]
    lib.ssa_batch_partition_by_key.restype = i32
    lib.ssa_specialization_stats.argtypes = [P(i64), P(ctypes.c_double), P(i64)]
    lib.ssa_specialization_stats.restype = None
    lib.ssa_filter_simple.argtypes = [i32, P(vp)]
    lib.ssa_filter_simple.restype = i32
    lib.ssa_filter_from_runs.argtypes = [vp, i32, vp, sz, P(vp), P(vp)]
    lib.ssa_filter_from_runs.restype = i32
    lib.ssa_filter_from_program.argtypes = [vp, sz, vp, P(vp), P(vp)]
    lib.ssa_filter_from_program.restype = i32
    lib.ssa_batch_from_blob.argtypes = [vp, vp, sz, vp, P(vp), P(vp)]
    lib.ssa_batch_from_blob.restype = i32
    lib.ssa_blob_describe.argtypes = [vp, sz, vp, P(vp)]
    lib.ssa_blob_describe.restype = vp
    lib.ssa_batches_from_blobs.argtypes = [vp, vp, vp, sz, vp, vp, P(vp)]
    lib.ssa_batches_from_blobs.restype = i32
    lib.ssa_filter_from_program_noalloc.argtypes = [vp, sz, vp, vp, P(i64), P(vp)]
    lib.ssa_filter_from_program_noalloc.restype = i32
    lib.ssa_filter_predicate.argtypes = [vp, vp, P(ArrowArray), P(ArrowSchema), i32, P(vp), P(vp)]
    lib.ssa_filter_predicate.restype = i32
    for f in (lib.ssa_filter_and, lib.ssa_filter_or, lib.ssa_filter_combine_sequential_and):
        f.argtypes = [vp, vp, vp, P(vp), P(vp)]
        f.restype = i32
    lib.ssa_filter_info.argtypes = [vp, P(i32), P(i32), P(i64)]
    lib.ssa_filter_info.restype = None
    lib.ssa_filter_runs.argtypes = [vp, vp, vp, P(sz), P(i32), P(vp)]
    lib.ssa_filter_runs.restype = i32
    lib.ssa_filter_bits.argtypes = [vp, vp, vp, P(vp)]
    lib.ssa_filter_bits.restype = i32
    lib.ssa_filter_filtered_count.argtypes = [vp, vp, P(i64), P(vp)]
    lib.ssa_filter_filtered_count.restype = i32
    lib.ssa_filter_apply.argtypes = [vp, vp, vp, P(vp), P(vp)]
    lib.ssa_filter_apply.restype = i32
    lib.ssa_filter_destroy.argtypes = [vp]
    lib.ssa_filter_destroy.restype = None
    _lib = lib
    return lib


def take_string(lib, ptr) -> str:
    if not ptr:
        return ""
    s = ctypes.string_at(ptr).decode(errors="replace")
    lib.ssa_free_string(ptr)
    return s

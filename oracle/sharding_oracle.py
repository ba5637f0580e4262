"""ctypes front end of the sharding oracle (libsharding_oracle.so).  TEST INFRASTRUCTURE -- the GPU product never imports this.

`row_bytes` restates NHash::TXX64::AppendField (ydb/core/formats/arrow/hash/calcer.cpp:31-50): the bytes a row feeds to
the hash are, column after column, the value bytes of every non-null field -- little-endian C values for numbers and
timestamps, one byte for a boolean, the bytes of a string; decimals add nothing.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from typing import Dict, List, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libsharding_oracle.so")
COMPUTE_SHARDING_SEED = 34323543  # ydb/core/formats/arrow/permutations.cpp:204

_lib = None


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            subprocess.check_call(["make", "-C", _HERE, "libsharding_oracle.so"])
        lib = ctypes.CDLL(_LIB_PATH)
        lib.orc_xxh64.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64]
        lib.orc_xxh64.restype = ctypes.c_uint64
        lib.orc_xxh64_rows.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_uint64, ctypes.c_void_p]
        lib.orc_xxh64_rows.restype = None
        lib.orc_sharding_split.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p]
        lib.orc_sharding_split.restype = None
        _lib = lib
    return _lib


def xxh64(data: bytes, seed: int = 0) -> int:
    buf = ctypes.create_string_buffer(data, len(data)) if data else ctypes.create_string_buffer(1)
    return int(load().orc_xxh64(ctypes.addressof(buf), len(data), seed))


def row_bytes(columns: Sequence, row: int) -> bytes:
    """columns: list of (values, valid or None) where values is a numpy array or a list of str / bytes / None."""
    out = bytearray()
    for values, valid in columns:
        if valid is not None and not valid[row]:
            continue
        v = values[row]
        if v is None:
            continue
        if isinstance(v, str):
            out += v.encode()
        elif isinstance(v, (bytes, bytearray)):
            out += bytes(v)
        elif isinstance(values, np.ndarray) and values.dtype == np.bool_:
            out += b"\x01" if v else b"\x00"
        else:
            out += np.asarray(v).tobytes()
    return bytes(out)


def hash_rows(columns: Sequence, nrows: int, seed: int) -> np.ndarray:
    rows = [row_bytes(columns, i) for i in range(nrows)]
    offsets = np.zeros(nrows + 1, dtype=np.int64)
    if nrows:
        offsets[1:] = np.cumsum([len(r) for r in rows])
    blob = np.frombuffer(b"".join(rows) + b"\x00", dtype=np.uint8).copy()
    out = np.zeros(max(1, nrows), dtype=np.uint64)
    load().orc_xxh64_rows(blob.ctypes.data, offsets.ctypes.data, nrows, seed, out.ctypes.data)
    return out[:nrows]


def split(hashes_as_i64: np.ndarray, shards: int) -> Tuple[np.ndarray, List[int]]:
    """TShardingSplitIndex: (permutation, rows per shard)."""
    h = np.ascontiguousarray(hashes_as_i64, dtype=np.int64)
    n = len(h)
    perm = np.zeros(max(1, n), dtype=np.uint32)
    sizes = np.zeros(shards, dtype=np.uint32)
    load().orc_sharding_split(h.ctypes.data, n, shards, perm.ctypes.data, sizes.ctypes.data)
    return perm[:n], [int(x) for x in sizes]

"""ctypes front end of the TColumnFilter oracle (libfilter_oracle.so).  TEST INFRASTRUCTURE -- the GPU product never imports this.

`RleFilter` wraps the run-length restatement of ydb/core/formats/arrow/arrow_filter.{h,cpp}; `predicate_filter` restates
TColumnFilter::MakePredicateFilter (arrow_filter.cpp:17-136, 259-311) with numpy: rows are compared with a border tuple
key after key while they still equal it -- numbers with `<` and `>` (a NaN is neither, so the next key decides), strings
by memcmp and then by length (what Python's bytes comparison does), validity bitmaps never looked at.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from typing import List, Optional, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libfilter_oracle.so")
LESS, LESS_OR_EQUAL, GREATER, GREATER_OR_EQUAL = 1, 2, 3, 4  # ECompareType (arrow_filter.h:13-18)

_lib = None


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            subprocess.check_call(["make", "-C", _HERE, "libfilter_oracle.so"])
        lib = ctypes.CDLL(_LIB_PATH)
        vp = ctypes.c_void_p
        lib.orc_filter_from_bools.argtypes = [vp, ctypes.c_uint32]
        lib.orc_filter_from_bools.restype = vp
        lib.orc_filter_simple.argtypes = [ctypes.c_int]
        lib.orc_filter_simple.restype = vp
        lib.orc_filter_merge.argtypes = [vp, vp, ctypes.c_int]
        lib.orc_filter_merge.restype = vp
        lib.orc_filter_combine_sequential_and.argtypes = [vp, vp]
        lib.orc_filter_combine_sequential_and.restype = vp
        lib.orc_filter_info.argtypes = [vp] + [ctypes.POINTER(ctypes.c_int)] * 3 + [ctypes.POINTER(ctypes.c_uint32),
                                                                                    ctypes.POINTER(ctypes.c_uint64)]
        lib.orc_filter_info.restype = None
        lib.orc_filter_runs.argtypes = [vp, vp]
        lib.orc_filter_runs.restype = None
        lib.orc_filter_bools.argtypes = [vp, vp]
        lib.orc_filter_bools.restype = None
        lib.orc_filter_filtered_count.argtypes = [vp]
        lib.orc_filter_filtered_count.restype = ctypes.c_int64
        lib.orc_filter_free.argtypes = [vp]
        lib.orc_filter_free.restype = None
        _lib = lib
    return _lib


class FilterAbort(Exception):
    """The reference would abort here (Y_ABORT_UNLESS / AFL_VERIFY)."""


class RleFilter:
    def __init__(self, handle):
        if not handle:
            raise FilterAbort()
        self._h = handle

    # -- constructors --------------------------------------------------------------------------------------
    @staticmethod
    def from_bools(values: Sequence[bool]) -> "RleFilter":
        v = np.ascontiguousarray(np.asarray(values, dtype=np.bool_).astype(np.uint8))
        buf = v if len(v) else np.zeros(1, dtype=np.uint8)
        return RleFilter(load().orc_filter_from_bools(buf.ctypes.data, len(v)))

    @staticmethod
    def allow() -> "RleFilter":
        return RleFilter(load().orc_filter_simple(1))

    @staticmethod
    def deny() -> "RleFilter":
        return RleFilter(load().orc_filter_simple(0))

    # -- algebra -------------------------------------------------------------------------------------------
    def And(self, other: "RleFilter") -> "RleFilter":
        return RleFilter(load().orc_filter_merge(self._h, other._h, 0))

    def Or(self, other: "RleFilter") -> "RleFilter":
        return RleFilter(load().orc_filter_merge(self._h, other._h, 1))

    def CombineSequentialAnd(self, ext: "RleFilter") -> "RleFilter":
        return RleFilter(load().orc_filter_combine_sequential_and(self._h, ext._h))

    # -- state ---------------------------------------------------------------------------------------------
    def info(self) -> Tuple[bool, bool, bool, int, int]:
        """(Filter.empty(), DefaultFilterValue, LastValue, Count, number of runs)"""
        e, d, l = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        c, n = ctypes.c_uint32(), ctypes.c_uint64()
        load().orc_filter_info(self._h, ctypes.byref(e), ctypes.byref(d), ctypes.byref(l), ctypes.byref(c), ctypes.byref(n))
        return bool(e.value), bool(d.value), bool(l.value), int(c.value), int(n.value)

    def runs(self) -> np.ndarray:
        n = self.info()[4]
        out = np.zeros(max(1, n), dtype=np.uint32)
        load().orc_filter_runs(self._h, out.ctypes.data)
        return out[:n]

    def bools(self) -> np.ndarray:
        count = self.info()[3]
        out = np.zeros(max(1, count), dtype=np.uint8)
        load().orc_filter_bools(self._h, out.ctypes.data)
        return out[:count].astype(np.bool_)

    def filtered_count(self) -> Optional[int]:
        v = int(load().orc_filter_filtered_count(self._h))
        return None if v < 0 else v

    def __del__(self):
        try:
            if self._h:
                load().orc_filter_free(self._h)
                self._h = None
        except Exception:
            pass


def compare_rows(columns: Sequence, border: Sequence) -> np.ndarray:
    """CompositeCompare: -1 / 0 / +1 per row.  columns[k] is a numpy array or a list of bytes, border[k] its border value."""
    n = len(columns[0]) if columns else 0
    res = np.zeros(n, dtype=np.int8)
    for values, b in zip(columns, border):
        undecided = res == 0
        if not undecided.any():
            break  # "no borders": the key tail cannot change anything (arrow_filter.cpp:132-134)
        if isinstance(values, np.ndarray) and values.dtype.kind in "iufb":
            bv = values.dtype.type(b)
            with np.errstate(invalid="ignore"):
                lt, gt = values < bv, values > bv
            res[undecided & lt] = -1
            res[undecided & gt] = 1
        else:
            bb = bytes(b)
            for i in np.nonzero(undecided)[0]:
                v = bytes(values[i])
                res[i] = -1 if v < bb else (1 if v > bb else 0)
    return res


def predicate_filter(columns: Sequence, border: Sequence, compare_type: int) -> RleFilter:
    res = compare_rows(columns, border)
    if compare_type == LESS:
        bits = res < 0
    elif compare_type == LESS_OR_EQUAL:
        bits = res <= 0
    elif compare_type == GREATER:
        bits = res > 0
    else:
        bits = res >= 0
    return RleFilter.from_bools(bits)

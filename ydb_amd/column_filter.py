"""Host-side mirror of NArrow::TColumnFilter (ydb/core/formats/arrow/arrow_filter.h:20-211) over the C-ABI.

The filter's bits live in HBM; every method is one `ssa_filter_*` call.  Method names and meanings are the reference's:
`And`, `Or`, `CombineSequentialAnd`, `MakePredicateFilter`, `Apply`, `GetFilteredCount`, `BuildSimpleFilter`, `Size`,
`IsEmpty`, `IsTotalAllowFilter`, `IsTotalDenyFilter`, `BuildAllowFilter`, `BuildDenyFilter`.  Where the reference
aborts (filters of different sizes, a second filter that does not fit the first) an `SsaError` with code INVALID is
raised instead.  The executor (HIP stream, device) is the one of the `ProgramContainer` the filter is made with.
"""
from __future__ import annotations

import ctypes
from typing import Dict, Optional, Sequence, Tuple, Union

import numpy as np

from . import _lib
from .arrow_c import HostBatch
from .program import DeviceBatch, ProgramContainer, SsaError

LESS, LESS_OR_EQUAL, GREATER, GREATER_OR_EQUAL = 1, 2, 3, 4  # ECompareType


class ColumnFilter:
    def __init__(self, container: ProgramContainer, handle):
        self._pc = container
        self._lib = container._lib
        self._h = handle

    # -- construction --------------------------------------------------------------------------------------
    @staticmethod
    def _wrap(pc: ProgramContainer, rc: int, out: ctypes.c_void_p, err: ctypes.c_void_p) -> "ColumnFilter":
        msg = _lib.take_string(pc._lib, err.value)
        if rc != _lib.SSA_OK:
            raise SsaError(rc, msg)
        return ColumnFilter(pc, out.value)

    @staticmethod
    def BuildAllowFilter(pc: ProgramContainer) -> "ColumnFilter":
        out = ctypes.c_void_p()
        pc._lib.ssa_filter_simple(1, ctypes.byref(out))
        return ColumnFilter(pc, out.value)

    @staticmethod
    def BuildDenyFilter(pc: ProgramContainer) -> "ColumnFilter":
        out = ctypes.c_void_p()
        pc._lib.ssa_filter_simple(0, ctypes.byref(out))
        return ColumnFilter(pc, out.value)

    @staticmethod
    def FromRuns(pc: ProgramContainer, first_value: bool, runs: Sequence[int]) -> "ColumnFilter":
        r = np.ascontiguousarray(np.asarray(runs, dtype=np.uint32))
        buf = r if len(r) else np.zeros(1, dtype=np.uint32)
        out, err = ctypes.c_void_p(), ctypes.c_void_p()
        rc = pc._lib.ssa_filter_from_runs(pc._h, 1 if first_value else 0, buf.ctypes.data, len(r), ctypes.byref(out), ctypes.byref(err))
        return ColumnFilter._wrap(pc, rc, out, err)

    @staticmethod
    def FromValues(pc: ProgramContainer, values: Sequence[bool]) -> "ColumnFilter":
        """TColumnFilter(std::vector<bool>&&): the values are run-length encoded on the host and expanded in HBM."""
        v = np.asarray(values, dtype=np.bool_)
        if len(v) == 0:
            return ColumnFilter.FromRuns(pc, True, [])
        edges = np.flatnonzero(v[1:] != v[:-1]) + 1
        bounds = np.concatenate(([0], edges, [len(v)]))
        return ColumnFilter.FromRuns(pc, bool(v[0]), np.diff(bounds))

    @staticmethod
    def FromProgram(pc: ProgramContainer, batch: DeviceBatch, step: int = 0) -> Optional["ColumnFilter"]:
        """TProgramStep::BuildFilter with the result kept in HBM; None when the step has no Filter command."""
        out, err = ctypes.c_void_p(), ctypes.c_void_p()
        rc = pc._lib.ssa_filter_from_program(pc._h, step, batch._h, ctypes.byref(out), ctypes.byref(err))
        msg = _lib.take_string(pc._lib, err.value)
        if rc != _lib.SSA_OK:
            raise SsaError(rc, msg)
        return ColumnFilter(pc, out.value) if out.value else None

    @staticmethod
    def MakePredicateFilter(pc: ProgramContainer, batch: DeviceBatch, border: Union[HostBatch, Dict[str, object]],
                            compare_type: int) -> "ColumnFilter":
        hb = border if isinstance(border, HostBatch) else HostBatch(border)
        out, err = ctypes.c_void_p(), ctypes.c_void_p()
        rc = pc._lib.ssa_filter_predicate(pc._h, batch._h, ctypes.byref(hb.array), ctypes.byref(hb.schema), compare_type,
                                          ctypes.byref(out), ctypes.byref(err))
        return ColumnFilter._wrap(pc, rc, out, err)

    # -- algebra -------------------------------------------------------------------------------------------
    def _binary(self, fn, other: "ColumnFilter") -> "ColumnFilter":
        out, err = ctypes.c_void_p(), ctypes.c_void_p()
        rc = fn(self._pc._h, self._h, other._h, ctypes.byref(out), ctypes.byref(err))
        return ColumnFilter._wrap(self._pc, rc, out, err)

    def And(self, other: "ColumnFilter") -> "ColumnFilter":
        return self._binary(self._lib.ssa_filter_and, other)

    def Or(self, other: "ColumnFilter") -> "ColumnFilter":
        return self._binary(self._lib.ssa_filter_or, other)

    def CombineSequentialAnd(self, ext: "ColumnFilter") -> "ColumnFilter":
        return self._binary(self._lib.ssa_filter_combine_sequential_and, ext)

    # -- state ---------------------------------------------------------------------------------------------
    def _info(self) -> Tuple[bool, bool, int]:
        e, d, c = ctypes.c_int(), ctypes.c_int(), ctypes.c_int64()
        self._lib.ssa_filter_info(self._h, ctypes.byref(e), ctypes.byref(d), ctypes.byref(c))
        return bool(e.value), bool(d.value), int(c.value)

    def IsEmpty(self) -> bool:
        return self._info()[0]

    def DefaultFilterValue(self) -> bool:
        return self._info()[1]

    def Size(self) -> int:
        return self._info()[2]

    def Runs(self) -> Tuple[np.ndarray, bool]:
        """(the `Filter` vector of run lengths, LastValue)"""
        n, last, err = ctypes.c_size_t(0), ctypes.c_int(0), ctypes.c_void_p()
        rc = self._lib.ssa_filter_runs(self._pc._h, self._h, None, ctypes.byref(n), ctypes.byref(last), ctypes.byref(err))
        self._pc._check(rc, err)
        runs = np.zeros(max(1, n.value), dtype=np.uint32)
        err = ctypes.c_void_p()
        rc = self._lib.ssa_filter_runs(self._pc._h, self._h, runs.ctypes.data, ctypes.byref(n), ctypes.byref(last), ctypes.byref(err))
        self._pc._check(rc, err)
        return runs[:n.value], bool(last.value)

    def IsTotalAllowFilter(self) -> bool:
        empty, default, _ = self._info()
        if empty:
            return default
        runs, last = self.Runs()
        return len(runs) == 1 and last

    def IsTotalDenyFilter(self) -> bool:
        empty, default, _ = self._info()
        if empty:
            return not default
        runs, last = self.Runs()
        return len(runs) == 1 and not last

    def BuildSimpleFilter(self) -> np.ndarray:
        count = self.Size()
        bits = np.zeros((count + 7) // 8 + 8, dtype=np.uint8)
        err = ctypes.c_void_p()
        rc = self._lib.ssa_filter_bits(self._pc._h, self._h, bits.ctypes.data, ctypes.byref(err))
        self._pc._check(rc, err)
        return np.unpackbits(bits, bitorder="little")[:count].astype(np.bool_)

    def GetFilteredCount(self) -> Optional[int]:
        out, err = ctypes.c_int64(0), ctypes.c_void_p()
        rc = self._lib.ssa_filter_filtered_count(self._pc._h, self._h, ctypes.byref(out), ctypes.byref(err))
        self._pc._check(rc, err)
        return None if out.value < 0 else int(out.value)

    def Apply(self, batch: DeviceBatch) -> DeviceBatch:
        out, err = ctypes.c_void_p(), ctypes.c_void_p()
        rc = self._lib.ssa_filter_apply(self._pc._h, self._h, batch._h, ctypes.byref(out), ctypes.byref(err))
        self._pc._check(rc, err)
        return DeviceBatch(self._lib, out.value)

    def close(self):
        if self._h:
            self._lib.ssa_filter_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

"""Host-side mirror of NArrow::NMerger::TMergePartialStream (ydb/core/formats/arrow/reader/merger.h:11-101) over
HBM-resident sources: AddSource collects sorted batches (with their optional filters), DrainAll merges them on the
device into one sorted batch with one row -- the newest version -- per key."""
from __future__ import annotations

import ctypes
from typing import List, Optional, Sequence

from . import _lib
from .program import DeviceBatch, ProgramContainer, SsaError


class MergePartialStream:
    def __init__(self, container: ProgramContainer, sort_columns: Sequence[str], reverse: bool = False,
                 version_columns: Sequence[str] = ()):
        if not sort_columns:
            raise ValueError("the sort schema has no columns")  # Y_ABORT_UNLESS(SortSchema->num_fields()), merger.h:48
        self._pc = container
        self._sort = list(sort_columns)
        self._versions = list(version_columns)
        self._reverse = bool(reverse)
        self._sources: List[DeviceBatch] = []
        self._filters: List[Optional[object]] = []

    def AddSource(self, batch: Optional[DeviceBatch], filter=None) -> None:
        """merger.h:78-89: sources without rows, and sources whose filter denies every row, are not taken."""
        if batch is None or batch.rows == 0:
            return
        if filter is not None and filter.IsTotalDenyFilter():
            return
        self._sources.append(batch)
        self._filters.append(None if (filter is None or filter.IsTotalAllowFilter()) else filter)

    def GetSourcesCount(self) -> int:
        return len(self._sources)

    def IsEmpty(self) -> bool:
        return not self._sources

    def DrainCurrentTo(self, read_to, include_finish: bool) -> Optional[DeviceBatch]:
        """merger.cpp:39-70 (PutControlPoint + DrainToControlPoint): the merged rows whose key is below the border
        `read_to` (a one-row batch of the sort columns) -- and equal to it when `include_finish`: the control point sorts
        in front of the rows of its key (batch_iterator.cpp:33-42) and DrainToControlPoint drains that key only with
        includeFinish.  The other rows stay in the stream for the next drain.  Every source is sorted, so the rows that
        leave are a prefix of it: one predicate filter per source finds its length, the prefix goes through the merge,
        the rest is kept as a view.  Forward order only."""
        if self._reverse:
            raise NotImplementedError("DrainCurrentTo over sources read from their ends")
        from .column_filter import ColumnFilter
        LESS, LESS_OR_EQUAL = 1, 2
        head = MergePartialStream(self._pc, self._sort, False, self._versions)
        rest_sources, rest_filters = [], []
        for batch, flt in zip(self._sources, self._filters):
            cut = ColumnFilter.MakePredicateFilter(self._pc, batch, read_to, LESS_OR_EQUAL if include_finish else LESS)
            k = cut.GetFilteredCount()
            k = batch.rows if k is None else k
            bits = flt.BuildSimpleFilter() if flt is not None else None
            if k > 0:
                head.AddSource(self._pc.Slice(batch, 0, k), ColumnFilter.FromValues(self._pc, list(bits[:k])) if bits is not None else None)
            if k < batch.rows:
                rest_sources.append(self._pc.Slice(batch, k, batch.rows - k))
                rest_filters.append(ColumnFilter.FromValues(self._pc, list(bits[k:])) if bits is not None else None)
        self._sources, self._filters = rest_sources, rest_filters
        return head.DrainAll()

    def DrainAll(self) -> Optional[DeviceBatch]:
        """merger.cpp:145-150.  None when there is no source to drain."""
        if not self._sources:
            return None
        lib = self._pc._lib
        n = len(self._sources)
        src = (ctypes.c_void_p * n)(*[b._h for b in self._sources])
        flt = (ctypes.c_void_p * n)(*[(f._h if f is not None else None) for f in self._filters])
        keys = (ctypes.c_char_p * len(self._sort))(*[k.encode() for k in self._sort])
        vers = (ctypes.c_char_p * max(1, len(self._versions)))(*[k.encode() for k in self._versions])
        out, err = ctypes.c_void_p(), ctypes.c_void_p()
        rc = lib.ssa_merge_sorted(self._pc._h, src, flt, n, keys, len(self._sort), vers, len(self._versions), 1 if self._reverse else 0,
                                  ctypes.byref(out), ctypes.byref(err))
        msg = _lib.take_string(lib, err.value)
        if rc != _lib.SSA_OK:
            raise SsaError(rc, msg)
        self._sources, self._filters = [], []
        return DeviceBatch(lib, out.value)

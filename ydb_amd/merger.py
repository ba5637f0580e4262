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

    def _split(self, border, compare: int):
        """Per source: the number of rows at the reading end for which `row <compare> border` holds (every source is
        sorted, so they are a prefix of it, or in reverse order a suffix), found by one predicate filter."""
        from .column_filter import ColumnFilter
        counts = []
        for batch in self._sources:
            k = ColumnFilter.MakePredicateFilter(self._pc, batch, border, compare).GetFilteredCount()
            counts.append(batch.rows if k is None else k)
        return counts

    def _cut(self, counts, keep_head: bool):
        """Takes `counts[i]` rows off the reading end of every source.  Returns a stream of the rows taken when
        `keep_head`; the rows left stay in this one (sources with no rows left are dropped, as the heap drops them)."""
        from .column_filter import ColumnFilter
        head = MergePartialStream(self._pc, self._sort, self._reverse, self._versions) if keep_head else None
        rest_sources, rest_filters = [], []
        for batch, flt, k in zip(self._sources, self._filters, counts):
            bits = flt.BuildSimpleFilter() if flt is not None else None
            n = batch.rows
            taken, left = ((n - k, n), (0, n - k)) if self._reverse else ((0, k), (k, n))
            if head is not None and k > 0:
                head.AddSource(self._pc.Slice(batch, taken[0], k),
                               ColumnFilter.FromValues(self._pc, list(bits[taken[0]:taken[1]])) if bits is not None else None)
            if k < n:
                rest_sources.append(self._pc.Slice(batch, left[0], n - k))
                rest_filters.append(ColumnFilter.FromValues(self._pc, list(bits[left[0]:left[1]])) if bits is not None else None)
        self._sources, self._filters = rest_sources, rest_filters
        return head

    def DrainCurrentTo(self, read_to, include_finish: bool) -> Optional[DeviceBatch]:
        """merger.cpp:39-70 (PutControlPoint + DrainToControlPoint): the merged rows whose key comes before the border
        `read_to` (a one-row batch of the sort columns) -- and those equal to it when `include_finish`: the control point
        sorts in front of the rows of its key (batch_iterator.cpp:33-42) and DrainToControlPoint drains that key only
        with includeFinish.  "Before" is below in forward order and above when the sources are read from their ends
        (batch_iterator.h:18-24).  The other rows stay in the stream for the next drain.  One predicate filter per
        source finds how many rows leave, those go through the merge, the rest is kept as a view."""
        LESS, LESS_OR_EQUAL, GREATER, GREATER_OR_EQUAL = 1, 2, 3, 4
        if self._reverse:
            compare = GREATER_OR_EQUAL if include_finish else GREATER
        else:
            compare = LESS_OR_EQUAL if include_finish else LESS
        return self._cut(self._split(read_to, compare), True).DrainAll()

    def SkipToLowerBound(self, pos, include: bool) -> None:
        """merger.cpp:209-236: every source is moved forward to its first row that is not before `pos` -- and past the
        rows equal to it unless `include`; nothing is produced."""
        LESS, LESS_OR_EQUAL, GREATER, GREATER_OR_EQUAL = 1, 2, 3, 4
        if self._reverse:
            compare = GREATER if include else GREATER_OR_EQUAL
        else:
            compare = LESS if include else LESS_OR_EQUAL
        self._cut(self._split(pos, compare), False)

    def DrainAllParts(self, positions) -> List[DeviceBatch]:
        """merger.cpp:188-207: `positions` is a sequence of (border, include) in stream order; one drain per border and
        one for what is left, parts without rows left out."""
        parts = []
        for border, include in positions:
            part = self.DrainCurrentTo(border, include)
            if part is not None and part.rows:
                parts.append(part)
        part = self.DrainAll()
        if part is not None and part.rows:
            parts.append(part)
        return parts

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

"""CPU restatement of NArrow::NMerger::TMergePartialStream::DrainAll.  TEST INFRASTRUCTURE -- the GPU product never
imports this.

Follows ydb/core/formats/arrow/reader/merger.cpp:145-186 (DrainAll / DrainCurrentPosition), batch_iterator.cpp:25-49
(iterator order: key, then version descending; control points are not used by DrainAll), batch_iterator.h:18-24 (reverse
reads a source from its end), heap.h:35-60 and switch/compare.h:82-119 (values compared without looking at validity;
strings as bytes, then by length).  Pinned on the three merger tests of ydb/core/formats/arrow/ut/ut_arrow.cpp:655-740
(tests/test_oracle_merge.py).

A source is a dict name -> list of values (ints, floats, bytes) of equal lengths, sorted ascending by the sort columns;
`filters` gives per source None or a list of booleans (False = the row is deleted).  Among rows of one key and one
version the reference's heap leaves the order open; this restatement takes the row of the earlier source, as the
product does.
"""
from __future__ import annotations

import functools
import heapq
from typing import Dict, List, Optional, Sequence


def _cmp(x, y) -> int:
    return -1 if x < y else (1 if x > y else 0)


class _Iterator:
    def __init__(self, index: int, source: Dict[str, list], keep: Optional[Sequence[bool]], sort: Sequence[str], versions: Sequence[str], reverse: bool):
        self.index = index
        self.source = source
        self.keep = keep
        self.sort, self.versions = sort, versions
        self.rows = len(source[sort[0]])
        self.step = -1 if reverse else 1
        self.pos = self.rows - 1 if reverse else 0  # GetFirstPosition
        self.reverse = reverse

    def key(self):
        return tuple(self.source[c][self.pos] for c in self.sort)

    def version(self):
        return tuple(self.source[c][self.pos] for c in self.versions)

    def deleted(self) -> bool:
        return self.keep is not None and not self.keep[self.pos]

    def next(self) -> bool:
        self.pos += self.step
        return 0 <= self.pos < self.rows


def _order(a: _Iterator, b: _Iterator) -> int:
    """< 0 when `a` leaves the heap before `b`: smaller key first (greater when reversed), then the greater version."""
    for x, y in zip(a.key(), b.key()):
        c = _cmp(x, y)
        if c:
            return -c if a.reverse else c
    for x, y in zip(a.version(), b.version()):
        c = _cmp(x, y)
        if c:
            return -c
    return _cmp(a.index, b.index)


def drain_all(sources: Sequence[Dict[str, list]], sort: Sequence[str], versions: Sequence[str] = (), reverse: bool = False,
              filters: Optional[Sequence[Optional[Sequence[bool]]]] = None) -> Dict[str, list]:
    names = list(sources[0].keys()) if sources else []
    out: Dict[str, list] = {n: [] for n in names}
    wrap = functools.cmp_to_key(_order)
    heap = []
    for i, s in enumerate(sources):
        keep = filters[i] if filters else None
        rows = len(s[sort[0]])
        if rows == 0:  # AddSource
            continue
        if keep is not None and not any(keep):
            continue
        if keep is not None and all(keep):
            keep = None
        it = _Iterator(i, s, keep, sort, versions, reverse)
        heap.append(it)
    items = [wrap(it) for it in heap]
    heapq.heapify(items)
    while items:  # DrainCurrentPosition
        top = items[0].obj
        if not top.deleted():
            for n in names:
                out[n].append(top.source[n][top.pos])
        start = top.key()
        first = True
        while items and (first or items[0].obj.key() == start):
            cur = items[0].obj
            if cur.next():  # SortHeap.Next(): UpdateTop or RemoveTop
                heapq.heapreplace(items, wrap(cur))
            else:
                heapq.heappop(items)
            first = False
    return out


def drain_current_to(sources: Sequence[Dict[str, list]], sort: Sequence[str], read_to: Sequence, include_finish: bool, versions: Sequence[str] = (),
                     filters: Optional[Sequence[Optional[Sequence[bool]]]] = None, reverse: bool = False):
    """TMergePartialStream::DrainCurrentTo (merger.cpp:39-70): a control point with the key `read_to` is put into the
    heap, where it sorts in front of the rows of its own key (batch_iterator.cpp:33-42); positions are drained until it
    is reached, and the position of its key as well when `include_finish` (merger.cpp:47-59).  With `reverse` the
    sources are read from their ends and "in front of" means a greater key (batch_iterator.h:18-24).
    Returns (drained rows, what is left of every source, what is left of every filter)."""
    names = list(sources[0].keys()) if sources else []
    out: Dict[str, list] = {n: [] for n in names}
    border = tuple(read_to)
    before = (lambda k: k > border) if reverse else (lambda k: k < border)
    heap = []
    for i, s in enumerate(sources):
        keep = filters[i] if filters else None
        if len(s[sort[0]]) == 0:
            continue
        heap.append(_Iterator(i, s, keep, sort, versions, reverse))
    wrap = functools.cmp_to_key(_order)
    items = [wrap(it) for it in heap]
    heapq.heapify(items)
    reached = False
    while items and not reached:
        top = items[0].obj
        if not before(top.key()):  # the control point is on top: it precedes rows of its own key and all that follow
            reached = True
            if not include_finish or top.key() != border:
                break
        if not top.deleted():
            for n in names:
                out[n].append(top.source[n][top.pos])
        start = top.key()
        first = True
        while items and (first or items[0].obj.key() == start):
            cur = items[0].obj
            if cur.next():
                heapq.heapreplace(items, wrap(cur))
            else:
                heapq.heappop(items)
            first = False
    left_pos = {it.obj.index: it.obj.pos for it in items}
    rest, rest_filters = [], []
    for i, s in enumerate(sources):
        n = len(s[sort[0]])
        if reverse:  # rows 0 .. pos are left
            lo, hi = 0, left_pos.get(i, -1) + 1
        else:
            lo, hi = left_pos.get(i, n), n
        rest.append({c: v[lo:hi] for c, v in s.items()})
        rest_filters.append(None if not filters or filters[i] is None else list(filters[i][lo:hi]))
    return out, rest, rest_filters


def skip_to_lower_bound(sources: Sequence[Dict[str, list]], sort: Sequence[str], pos: Sequence, include: bool,
                        filters: Optional[Sequence[Optional[Sequence[bool]]]] = None, reverse: bool = False):
    """TMergePartialStream::SkipToLowerBound (merger.cpp:209-236): while the heap's top is before `pos` -- or at it and
    not `include` -- its source is moved to its first row not before `pos` (SkipToLower) and, at `pos` without `include`,
    one row further; a source that ends on the way leaves the heap.  Restated source by source: the rows it drops are
    those before `pos` and, without `include`, those equal to it.  Returns (what is left of every source, of every filter)."""
    border = tuple(pos)
    rest, rest_filters = [], []
    for i, s in enumerate(sources):
        n = len(s[sort[0]])
        keys = [tuple(s[c][r] for c in sort) for r in range(n)]
        if reverse:
            gone = sum(1 for k in keys if k > border or (k == border and not include))
            lo, hi = 0, n - gone
        else:
            gone = sum(1 for k in keys if k < border or (k == border and not include))
            lo, hi = gone, n
        rest.append({c: v[lo:hi] for c, v in s.items()})
        rest_filters.append(None if not filters or filters[i] is None else list(filters[i][lo:hi]))
    return rest, rest_filters

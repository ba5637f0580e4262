"""The merge oracle against the reference's own merger tests (ydb/core/formats/arrow/ut/ut_arrow.cpp:655-740):
MergingSortedInputStream, MergingSortedInputStreamReversed and MergingSortedInputStreamReplace over MakeTable1000
(ut_arrow.cpp:413-432), plus by-hand cases of deleted rows."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from oracle import merge_oracle  # noqa: E402


def table1000():
    """MakeTable1000: row i = (i8 i / 100, i16 (i % 100) / 10, i32 i % 10), sorted by the three."""
    return {"i8": [i // 100 for i in range(1000)], "i16": [(i % 100) // 10 for i in range(1000)], "i32": [i % 10 for i in range(1000)]}


def cut(t, start, count):
    return {k: v[start:start + count] for k, v in t.items()}


def restore(t, i):
    return t["i8"][i] * 100 + t["i16"][i] * 10 + t["i32"][i]


SLICES = [(0, 100), (100, 200), (200, 400), (500, 50), (600, 1)]


def test_merging_sorted_input_stream():
    t = table1000()
    got = merge_oracle.drain_all([cut(t, a, n) for a, n in SLICES], ["i8", "i16", "i32"], versions=["i8"])
    assert len(got["i8"]) == 601  # UNIT_ASSERT_VALUES_EQUAL(sorted->num_rows(), 601)
    assert [restore(got, i) for i in range(601)] == list(range(601))


def test_merging_sorted_input_stream_reversed():
    t = table1000()
    got = merge_oracle.drain_all([cut(t, a, n) for a, n in SLICES], ["i8", "i16", "i32"], versions=["i8"], reverse=True)
    assert len(got["i8"]) == 601
    assert [restore(got, i) for i in range(601)] == list(range(600, -1, -1))


def test_merging_sorted_input_stream_replace():
    t = table1000()
    sources = []
    for snap, start in enumerate((0, 200, 400, 600)):
        s = cut(t, start, 400)
        s["snap"] = [snap] * 400  # AddSnapColumn
        sources.append(s)
    got = merge_oracle.drain_all(sources, ["i8", "i16", "i32"], versions=["snap"])
    assert len(got["i8"]) == 1000
    assert [restore(got, i) for i in range(1000)] == list(range(1000))  # CheckSorted1000, IsSortedAndUnique
    counts = {v: got["snap"].count(v) for v in range(4)}
    assert counts == {0: 200, 1: 200, 2: 200, 3: 400}


def test_a_deleted_newest_version_hides_its_key():
    # key 2: the newest version (source 1) is deleted -> the key is gone, the older row of source 0 is skipped with it
    # key 3: only the older version is deleted -> the newest stays
    a = {"k": [1, 2, 3], "v": [1, 1, 1], "x": [10, 20, 30]}
    b = {"k": [2, 3, 4], "v": [2, 2, 2], "x": [21, 31, 41]}
    got = merge_oracle.drain_all([a, b], ["k"], versions=["v"], filters=[[True, True, False], [False, True, True]])
    assert got == {"k": [1, 3, 4], "v": [1, 2, 2], "x": [10, 31, 41]}
    # duplicates inside one source: its iterator shows the heap the first of them (the last when reading from the end),
    # the others are skipped whatever their version; sources that are deleted as a whole take no part
    c = {"k": [5, 5, 6], "v": [1, 2, 1], "x": [1, 2, 3]}
    got = merge_oracle.drain_all([c, a], ["k"], versions=["v"], filters=[None, [False, False, False]])
    assert got == {"k": [5, 6], "v": [1, 1], "x": [1, 3]}
    got = merge_oracle.drain_all([c], ["k"], versions=["v"], reverse=True)
    assert got == {"k": [6, 5], "v": [1, 2], "x": [3, 2]}
    d = {"k": [5, 7], "v": [3, 0], "x": [9, 8]}
    got = merge_oracle.drain_all([c, d], ["k"], versions=["v"])
    assert got == {"k": [5, 6, 7], "v": [3, 1, 0], "x": [9, 3, 8]}


def test_drain_current_to_stops_at_the_control_point():
    a = {"k": [1, 3, 5, 7], "v": [1, 1, 1, 1]}
    b = {"k": [2, 3, 6, 7], "v": [2, 2, 2, 2]}
    got, rest, _ = merge_oracle.drain_current_to([a, b], ["k"], [5], include_finish=False, versions=["v"])
    assert got == {"k": [1, 2, 3], "v": [1, 2, 2]}
    assert rest == [{"k": [5, 7], "v": [1, 1]}, {"k": [6, 7], "v": [2, 2]}]
    got, rest, _ = merge_oracle.drain_current_to([a, b], ["k"], [5], include_finish=True, versions=["v"])
    assert got == {"k": [1, 2, 3, 5], "v": [1, 2, 2, 1]}
    assert rest == [{"k": [7], "v": [1]}, {"k": [6, 7], "v": [2, 2]}]
    # a border between keys, and one beyond every key
    got, rest, _ = merge_oracle.drain_current_to([a, b], ["k"], [4], include_finish=True, versions=["v"])
    assert got["k"] == [1, 2, 3] and rest[0]["k"] == [5, 7]
    got, rest, _ = merge_oracle.drain_current_to([a, b], ["k"], [100], include_finish=False, versions=["v"])
    assert got["k"] == [1, 2, 3, 5, 6, 7] and all(len(r["k"]) == 0 for r in rest)
    # draining in two steps is draining in one
    first, rest, _ = merge_oracle.drain_current_to([a, b], ["k"], [3], include_finish=True, versions=["v"])
    second = merge_oracle.drain_all(rest, ["k"], ["v"])
    whole = merge_oracle.drain_all([a, b], ["k"], ["v"])
    assert {n: first[n] + second[n] for n in whole} == whole


def test_drain_current_to_from_the_ends():
    """Sources read from their ends: the rows above the border leave, greatest first (batch_iterator.h:18-24)."""
    a = {"k": [1, 3, 5, 7], "v": [1, 1, 1, 1]}
    b = {"k": [2, 3, 6, 7], "v": [2, 2, 2, 2]}
    got, rest, _ = merge_oracle.drain_current_to([a, b], ["k"], [3], include_finish=False, versions=["v"], reverse=True)
    assert got == {"k": [7, 6, 5], "v": [2, 2, 1]}
    assert rest == [{"k": [1, 3], "v": [1, 1]}, {"k": [2, 3], "v": [2, 2]}]
    got, rest, _ = merge_oracle.drain_current_to([a, b], ["k"], [3], include_finish=True, versions=["v"], reverse=True)
    assert got == {"k": [7, 6, 5, 3], "v": [2, 2, 1, 2]}
    assert rest == [{"k": [1], "v": [1]}, {"k": [2], "v": [2]}]
    first, rest, _ = merge_oracle.drain_current_to([a, b], ["k"], [4], include_finish=True, versions=["v"], reverse=True)
    second = merge_oracle.drain_all(rest, ["k"], ["v"], True)
    whole = merge_oracle.drain_all([a, b], ["k"], ["v"], True)
    assert {n: first[n] + second[n] for n in whole} == whole


def test_skip_to_lower_bound():
    a = {"k": [1, 3, 3, 7], "v": [1, 2, 1, 1]}
    b = {"k": [2, 4], "v": [2, 2]}
    rest, _ = merge_oracle.skip_to_lower_bound([a, b], ["k"], [3], include=True)
    assert rest == [{"k": [3, 3, 7], "v": [2, 1, 1]}, {"k": [4], "v": [2]}]
    rest, _ = merge_oracle.skip_to_lower_bound([a, b], ["k"], [3], include=False)
    assert rest == [{"k": [7], "v": [1]}, {"k": [4], "v": [2]}]
    rest, flt = merge_oracle.skip_to_lower_bound([a, b], ["k"], [3], include=True, filters=[[True, False, True, True], None], reverse=True)
    assert rest == [{"k": [1, 3, 3], "v": [1, 2, 1]}, {"k": [2], "v": [2]}] and flt == [[True, False, True], None]
    rest, _ = merge_oracle.skip_to_lower_bound([a, b], ["k"], [100], include=True)
    assert all(len(r["k"]) == 0 for r in rest)
    # what a skip leaves is what a drain to the same border leaves
    _, left, _ = merge_oracle.drain_current_to([a, b], ["k"], [3], include_finish=False, versions=["v"])
    assert left == merge_oracle.skip_to_lower_bound([a, b], ["k"], [3], include=True)[0]

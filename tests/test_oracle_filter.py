"""The TColumnFilter oracle against what pins it (CPU): the reference's own known answers (ut_column_filter.cpp:9-27,
ut_arrow.cpp:604-634, held in tests/golden/column_filter_vectors.json) and, on random filters, the plain boolean-array
meaning of every operation (the run vectors must be the maximal runs of the resulting bits)."""
import json
import os

import numpy as np
import pytest

from oracle import filter_oracle as fo

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def vectors():
    with open(os.path.join(ROOT, "tests", "golden", "column_filter_vectors.json")) as f:
        return {c["name"]: c for c in json.load(f)["cases"]}


def key_comparison_columns(n=1000):
    i = np.arange(n)
    return {"i8": (i // 100).astype(np.int8), "i16": ((i % 100) // 10).astype(np.int16), "i32": (i % 10).astype(np.int32)}


def maximal_runs(bits):
    bits = np.asarray(bits, dtype=bool)
    if len(bits) == 0:
        return np.zeros(0, dtype=np.uint32)
    edges = np.flatnonzero(bits[1:] != bits[:-1]) + 1
    return np.diff(np.concatenate(([0], edges, [len(bits)]))).astype(np.uint32)


def check_state(f, bits):
    """A sized filter holds exactly the maximal runs of `bits`; LastValue is the last bit."""
    empty, default, last, count, nruns = f.info()
    if len(bits) == 0:
        assert empty and count == 0
        return
    assert not empty and count == len(bits)
    assert np.array_equal(f.runs(), maximal_runs(bits))
    assert last == bool(bits[-1])
    assert np.array_equal(f.bools(), np.asarray(bits, dtype=bool))


def test_merge_filters_known_answer():
    c = vectors()["MergeFilters"]
    r = fo.RleFilter.from_bools(c["a"]).Or(fo.RleFilter.from_bools(c["b"]))
    assert r.info()[3] == c["size"]
    assert r.bools().astype(int).tolist() == c["expect"]


def test_combine_filters_known_answer():
    c = vectors()["CombineFilters"]
    r = fo.RleFilter.from_bools(c["a"]).CombineSequentialAnd(fo.RleFilter.from_bools(c["b"]))
    assert r.info()[3] == c["size"]
    assert r.bools().astype(int).tolist() == c["expect"]


def test_key_comparison_known_answer():
    c = vectors()["KeyComparison"]
    cols = key_comparison_columns(c["rows"])
    names = list(c["border"])
    for name, ct in (("LESS", fo.LESS), ("LESS_OR_EQUAL", fo.LESS_OR_EQUAL), ("GREATER", fo.GREATER), ("GREATER_OR_EQUAL", fo.GREATER_OR_EQUAL)):
        f = fo.predicate_filter([cols[k] for k in names], [c["border"][k] for k in names], ct)
        lead, value = c["expect_leading"][name]
        bits = f.bools()
        assert len(bits) == c["rows"]
        assert np.all(bits[:lead] == value) and np.all(bits[lead:] != value)  # CheckFilter (ut_arrow.cpp:398-411)


def random_bits(rng, n, mean_run):
    if n == 0:
        return np.zeros(0, dtype=bool)
    flips = rng.random(n) < 1.0 / mean_run
    return (np.cumsum(flips) + rng.integers(0, 2)) % 2 == 1


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 1000, 50_000])
@pytest.mark.parametrize("mean_run", [1.5, 40, 5000])
def test_algebra_is_the_boolean_meaning(n, mean_run):
    rng = np.random.default_rng(n * 7 + int(mean_run))
    a, b = random_bits(rng, n, mean_run), random_bits(rng, n, mean_run)
    fa, fb = fo.RleFilter.from_bools(a), fo.RleFilter.from_bools(b)
    check_state(fa, a)
    check_state(fa.And(fb), a & b)
    check_state(fa.Or(fb), a | b)
    ext = random_bits(rng, int(a.sum()), mean_run)
    if len(ext):
        expect = a.copy()
        expect[np.flatnonzero(a)] = ext
        check_state(fa.CombineSequentialAnd(fo.RleFilter.from_bools(ext)), expect)
    kept = fa.filtered_count()
    assert kept == int(a.sum())


def test_filters_without_a_size():
    allow, deny = fo.RleFilter.allow(), fo.RleFilter.deny()
    f = fo.RleFilter.from_bools([1, 0, 0, 1])
    assert fo.RleFilter.from_bools([]).info()[:2] == (True, True)          # no rows: the allow filter
    assert allow.And(deny).info()[:2] == (True, False) and allow.Or(deny).info()[:2] == (True, True)
    assert f.And(allow).runs().tolist() == [1, 2, 1] and f.And(deny).info()[:2] == (True, False)
    assert f.Or(allow).info()[:2] == (True, True) and f.Or(deny).runs().tolist() == [1, 2, 1]
    assert allow.CombineSequentialAnd(f).runs().tolist() == [1, 2, 1] and deny.CombineSequentialAnd(f).info()[:2] == (True, False)
    assert f.CombineSequentialAnd(allow).runs().tolist() == [1, 2, 1] and f.CombineSequentialAnd(deny).info()[:2] == (True, False)
    assert allow.filtered_count() is None and deny.filtered_count() == 0
    assert fo.RleFilter.from_bools([1, 1]).filtered_count() == 2 and fo.RleFilter.from_bools([0, 0]).filtered_count() == 0


def test_where_the_reference_aborts():
    with pytest.raises(fo.FilterAbort):
        fo.RleFilter.from_bools([1, 0]).And(fo.RleFilter.from_bools([1, 0, 1]))
    with pytest.raises(fo.FilterAbort):
        fo.RleFilter.from_bools([1, 0, 1]).CombineSequentialAnd(fo.RleFilter.from_bools([1, 0, 1]))
    with pytest.raises(fo.FilterAbort):
        fo.RleFilter.from_bools([1, 0, 1]).CombineSequentialAnd(fo.RleFilter.from_bools([1]))


def test_predicate_strings_floats_and_nan():
    s = [b"", b"a", b"ab", b"abc", b"b", b"\xff"]
    bits = fo.predicate_filter([s], [b"ab"], fo.LESS).bools()
    assert bits.tolist() == [True, True, False, False, False, False]
    x = np.array([1.0, np.nan, 2.0, 2.0], dtype=np.float64)
    y = np.array([5, 1, 1, 9], dtype=np.uint32)
    # NaN is neither below nor above the border: the next key decides
    assert fo.predicate_filter([x, y], [2.0, 5], fo.LESS).bools().tolist() == [True, True, True, False]
    assert fo.predicate_filter([x, y], [2.0, 5], fo.GREATER_OR_EQUAL).bools().tolist() == [False, False, False, True]

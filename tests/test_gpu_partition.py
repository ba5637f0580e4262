"""GROUP BY whose rows that miss the workgroups' own tables are written out as records, sorted by region and aggregated
region by region inside LDS -- emitted from there straight into the output columns, or (SSA_KEYED_EMIT=0) applied to a
table in HBM -- which is what groupings of a million groups and more do by themselves; forced here on small inputs with
SSA_KEYED_PARTITION=1, with regions of the size LDS allows -- one round of the sort -- and of 64 slots -- two rounds.
Every result must equal the CPU oracle's and the row-by-row path's."""
import ctypes

import numpy as np
import pytest

from cases import compare_batches
from ydb_amd.arrow_c import Col
from ydb_amd.ssa_pb import AGG, F, Aggregate, Program

pytestmark = pytest.mark.gpu


@pytest.fixture(params=[None, "64"], ids=["regions_by_lds", "regions_of_64"], autouse=True)
def region_slots(request, monkeypatch):
    if request.param is not None:
        monkeypatch.setenv("SSA_KEYED_REGION", request.param)
    return request.param


@pytest.fixture(params=["1", "0"], ids=["emitted_from_lds", "table_in_hbm"], autouse=True)
def emit_mode(request, monkeypatch):
    monkeypatch.setenv("SSA_KEYED_EMIT", request.param)
    return request.param


def applied_by_region():
    from ydb_amd import _lib
    n = ctypes.c_int64(0)
    _lib.load().ssa_keyed_stats(ctypes.byref(n))
    return n.value


def sorted_again_with_counts():
    from ydb_amd import _lib
    n = ctypes.c_int64(0)
    _lib.load().ssa_keyed_sort_stats(ctypes.byref(n))
    return n.value


def run_three_ways(monkeypatch, resolver, program, batch, expect_region=True, lds=None):
    """Results with the records applied by region, with every row updating the table itself, and the oracle's."""
    from oracle import ssa_oracle
    from ydb_amd.program import ProgramContainer
    if lds is not None:
        monkeypatch.setenv("SSA_KEYED_LDS", lds)
    monkeypatch.setenv("SSA_JIT", "1")  # the records are written by the kernel built for the step: build it at the first scan
    out = []
    for mode in ("1", "0"):
        monkeypatch.setenv("SSA_KEYED_PARTITION", mode)
        pc = ProgramContainer()
        ok, err = pc.Init(resolver, program)
        assert ok, err
        dev = pc.Import(batch)
        before = applied_by_region()
        # twice: the second scan runs with the table sized by the first
        for _ in range(2):
            res = pc.Export(pc.ApplyDevice(dev))
        delta = applied_by_region() - before
        if mode == "1" and expect_region:
            assert delta >= 2
        if mode == "0":
            assert delta == 0
        out.append(res)
    out.append(ssa_oracle.Oracle(resolver, program).apply(batch))
    return out


def nullable(rng, a, frac):
    return Col(a, valid=rng.random(len(a)) >= frac)


def test_one_word_keys_all_aggregate_kinds(monkeypatch):
    rng = np.random.default_rng(501)
    n = 300_000
    batch = {
        "k": rng.integers(0, 60_000, n).astype(np.int64),
        "i": nullable(rng, rng.integers(-1000, 1000, n).astype(np.int32), 0.2),
        "f": nullable(rng, rng.random(n) * 100.0 + 1.0, 0.1),  # (positive: sums compare to a relative tolerance)
        "u": rng.integers(0, 1 << 40, n).astype(np.uint64),
    }
    res = [(1, "k"), (2, "i"), (3, "f"), (4, "u")]
    p = Program().group_by([Aggregate(100, AGG.COUNT, []), Aggregate(101, AGG.COUNT, [2]), Aggregate(102, AGG.SUM, [2]),
                            Aggregate(103, AGG.MIN, [2]), Aggregate(104, AGG.MAX, [3]), Aggregate(105, AGG.SUM, [3]),
                            Aggregate(106, AGG.MAX, [4]), Aggregate(107, AGG.MIN, [3]), Aggregate(108, AGG.SUM, [4])], keys=[1])
    region, direct, oracle = run_three_ways(monkeypatch, res, p, batch)
    compare_batches(region, direct, key_names=["k"], float_sum_names=["105"])
    compare_batches(region, oracle, key_names=["k"], float_sum_names=["105"])


def test_count_only_carries_no_operands(monkeypatch):
    rng = np.random.default_rng(502)
    n = 399_999
    batch = {"k": rng.integers(-(1 << 62), 1 << 62, n).astype(np.int64)}
    batch["k"][::3] = batch["k"][1::3]  # some keys twice
    p = Program().group_by([Aggregate(100, AGG.COUNT, [])], keys=[1])
    region, direct, oracle = run_three_ways(monkeypatch, [(1, "k")], p, batch)
    compare_batches(region, direct, key_names=["k"])
    compare_batches(region, oracle, key_names=["k"])


def test_two_word_keys_behind_a_filter(monkeypatch):
    rng = np.random.default_rng(503)
    n = 250_000
    batch = {
        "a": nullable(rng, rng.integers(0, 300, n).astype(np.int64), 0.05),
        "b": rng.integers(0, 300, n).astype(np.int64),
        "v": rng.integers(0, 100, n).astype(np.int32),
    }
    res = [(1, "a"), (2, "b"), (3, "v")]
    p = (Program().assign_const(50, "Int32", 20).assign_func(51, F.CMP_GREATER_EQUAL, [3, 50]).filter(51)
         .assign_func(52, F.MATH_ADD, [3, 50])
         .group_by([Aggregate(100, AGG.COUNT, []), Aggregate(101, AGG.SUM, [52]), Aggregate(102, AGG.MIN, [3])], keys=[1, 2]))
    region, direct, oracle = run_three_ways(monkeypatch, res, p, batch)
    compare_batches(region, direct, key_names=["a", "b"])
    compare_batches(region, oracle, key_names=["a", "b"])


def test_string_keys_short_and_long(monkeypatch):
    rng = np.random.default_rng(504)
    n = 120_000
    words = ["w%d" % i for i in range(9000)] + ["a much longer search phrase number %d" % i for i in range(9000)] + [""]
    pick = rng.integers(0, len(words), n)
    pick[rng.random(n) < 0.3] = len(words) - 1  # the empty phrase is the heavy hitter
    batch = {
        "s": Col([words[i] for i in pick], type="string"),
        "v": nullable(rng, rng.integers(0, 1000, n).astype(np.int64), 0.1),
    }
    res = [(1, "s"), (2, "v")]
    p = Program().group_by([Aggregate(100, AGG.COUNT, []), Aggregate(101, AGG.SUM, [2]), Aggregate(102, AGG.MAX, [2])], keys=[1])
    region, direct, oracle = run_three_ways(monkeypatch, res, p, batch)
    compare_batches(region, direct, key_names=["s"])
    compare_batches(region, oracle, key_names=["s"])


def test_string_keys_with_few_hash_values(monkeypatch):
    """Every long key hashed to one of eight values (a debug switch of the library): groups are told apart by their
    key bytes alone, in the region pass as in the scan."""
    rng = np.random.default_rng(505)
    n = 20_000
    words = ["the phrase with the number %d in it" % i for i in range(200)]
    batch = {"s": Col([words[i] for i in rng.integers(0, len(words), n)], type="string"), "v": rng.integers(0, 9, n).astype(np.int32)}
    res = [(1, "s"), (2, "v")]
    p = Program().group_by([Aggregate(100, AGG.COUNT, []), Aggregate(101, AGG.SUM, [2])], keys=[1])
    monkeypatch.setenv("SSA_DEBUG_FLAGS", "64")
    region, direct, oracle = run_three_ways(monkeypatch, res, p, batch, lds="16,1")
    compare_batches(region, direct, key_names=["s"])
    compare_batches(region, oracle, key_names=["s"])


def test_string_keys_that_differ_in_their_last_bytes(monkeypatch):
    """Long keys that share a hash (a debug switch of the library) are told apart by their bytes: keys of 9 .. 80 bytes
    that differ in the last byte only, keys that are the start of other keys, and the column's last string, whose bytes
    end where the buffer ends."""
    rng = np.random.default_rng(513)
    stem = "the quick brown fox jumps over the lazy dog and runs on through the fields of barley and rye"
    words = []
    for n in list(range(9, 81)):
        words += [stem[:n - 1] + c for c in "abc"]
    words += [stem[:n] for n in (10, 20, 31, 32, 33, 40, 64)]
    words = sorted(set(words))
    n = 30_000
    pick = rng.integers(0, len(words), n)
    pick[-1] = len(words) - 1
    batch = {"s": Col([words[i] for i in pick], type="string"), "v": rng.integers(0, 9, n).astype(np.int32)}
    res = [(1, "s"), (2, "v")]
    p = Program().group_by([Aggregate(100, AGG.COUNT, []), Aggregate(101, AGG.SUM, [2])], keys=[1])
    monkeypatch.setenv("SSA_DEBUG_FLAGS", "64")
    region, direct, oracle = run_three_ways(monkeypatch, res, p, batch, lds="16,1")
    assert len(oracle["s"].values) == len(words) if hasattr(oracle["s"], "values") else True
    compare_batches(region, direct, key_names=["s"])
    compare_batches(region, oracle, key_names=["s"])


def test_float_min_max_skip_nans(monkeypatch):
    rng = np.random.default_rng(506)
    n = 150_000
    f = rng.normal(size=n)
    f[rng.random(n) < 0.05] = np.nan
    batch = {"k": rng.integers(0, 40_000, n).astype(np.int32), "f": nullable(rng, f, 0.1)}
    res = [(1, "k"), (2, "f")]
    p = Program().group_by([Aggregate(100, AGG.MIN, [2]), Aggregate(101, AGG.MAX, [2]), Aggregate(102, AGG.COUNT, [2])], keys=[1])
    region, direct, oracle = run_three_ways(monkeypatch, res, p, batch)
    compare_batches(region, direct, key_names=["k"])
    compare_batches(region, oracle, key_names=["k"])


def test_no_level_one_table_every_row_is_a_record(monkeypatch):
    rng = np.random.default_rng(507)
    n = 70_003
    batch = {"k": rng.integers(0, 50, n).astype(np.int16), "v": rng.integers(-5, 5, n).astype(np.int64)}
    res = [(1, "k"), (2, "v")]
    p = Program().group_by([Aggregate(100, AGG.COUNT, []), Aggregate(101, AGG.SUM, [2])], keys=[1])
    region, direct, oracle = run_three_ways(monkeypatch, res, p, batch, lds="0,1")
    compare_batches(region, direct, key_names=["k"])
    compare_batches(region, oracle, key_names=["k"])


def test_some_takes_the_first_row_of_each_group(monkeypatch):
    rng = np.random.default_rng(508)
    n = 90_000
    batch = {"k": rng.integers(0, 30_000, n).astype(np.int64), "v": np.arange(n, dtype=np.int64)}
    res = [(1, "k"), (2, "v")]
    p = Program().group_by([Aggregate(100, AGG.SOME, [2]), Aggregate(101, AGG.COUNT, [])], keys=[1])
    region, direct, oracle = run_three_ways(monkeypatch, res, p, batch)
    compare_batches(region, direct, key_names=["k"])
    compare_batches(region, oracle, key_names=["k"])


def test_more_groups_than_the_last_scan_found(monkeypatch):
    """The output columns of the region pass are sized by the step's last scan: a batch with many times the groups
    runs the pass again with longer ones."""
    from oracle import ssa_oracle
    from ydb_amd.program import ProgramContainer
    monkeypatch.setenv("SSA_JIT", "1")
    monkeypatch.setenv("SSA_KEYED_PARTITION", "1")
    rng = np.random.default_rng(510)
    few = {"k": rng.integers(0, 500, 100_000).astype(np.int64), "v": rng.integers(0, 100, 100_000).astype(np.int32)}
    many = {"k": rng.integers(0, 1 << 40, 400_000).astype(np.int64), "v": rng.integers(0, 100, 400_000).astype(np.int32)}
    p = Program().group_by([Aggregate(100, AGG.COUNT, []), Aggregate(101, AGG.SUM, [2])], keys=[1])
    res = [(1, "k"), (2, "v")]
    pc = ProgramContainer()
    ok, err = pc.Init(res, p)
    assert ok, err
    for batch in (few, few, many, few, many):
        got = pc.Export(pc.ApplyDevice(pc.Import(batch)))
        compare_batches(got, ssa_oracle.Oracle(res, p).apply(batch), key_names=["k"])


def test_a_sort_whose_places_run_out_of_room_is_done_again_with_counts(monkeypatch, emit_mode, region_slots):
    """The first of the sort's two rounds goes without its counting pass, every digit's place in the sorted arrays with
    room to spare; with a third of the room (a switch of the library) some place is too small: the scan's records are
    sorted again with counts, the results are the same, and the step's next scans count from the start."""
    from oracle import ssa_oracle
    from ydb_amd.program import ProgramContainer
    if emit_mode != "1" or region_slots is None:
        pytest.skip("two rounds of the sort, the groups emitted from LDS: the table in HBM is always filled from counted places")
    monkeypatch.setenv("SSA_JIT", "1")
    monkeypatch.setenv("SSA_KEYED_PARTITION", "1")
    monkeypatch.setenv("SSA_KEYED_LDS", "0,1")  # (no tables of the workgroups' own: every row is a record)
    rng = np.random.default_rng(511)
    n = 350_000
    batch = {"k": rng.integers(0, 90_000, n).astype(np.int64), "v": nullable(rng, rng.integers(-100, 100, n).astype(np.int32), 0.1)}
    res = [(1, "k"), (2, "v")]
    p = Program().group_by([Aggregate(100, AGG.COUNT, []), Aggregate(101, AGG.SUM, [2]), Aggregate(102, AGG.MIN, [2])], keys=[1])
    want = ssa_oracle.Oracle(res, p).apply(batch)
    pc = ProgramContainer()
    ok, err = pc.Init(res, p)
    assert ok, err
    dev = pc.Import(batch)
    compare_batches(pc.Export(pc.ApplyDevice(dev)), want, key_names=["k"])  # (sizes the table; room to spare)
    monkeypatch.setenv("SSA_RADIX_ROOM_PERMILLE", "300")
    before = sorted_again_with_counts()
    compare_batches(pc.Export(pc.ApplyDevice(dev)), want, key_names=["k"])
    assert sorted_again_with_counts() - before == 1
    compare_batches(pc.Export(pc.ApplyDevice(dev)), want, key_names=["k"])
    assert sorted_again_with_counts() - before == 1


def test_large_grouping_takes_the_region_pass_by_itself(monkeypatch, region_slots):
    """Four million rows of two million keys: the second scan knows the group count and sorts its records by region."""
    from ydb_amd.program import ProgramContainer
    if region_slots is not None:
        pytest.skip("once, with the regions the library chooses")
    monkeypatch.delenv("SSA_KEYED_PARTITION", raising=False)
    monkeypatch.setenv("SSA_JIT", "1")
    rng = np.random.default_rng(509)
    n = 1 << 22
    k = rng.integers(0, 1 << 21, n).astype(np.int64)
    batch = {"k": k, "v": rng.integers(0, 100, n).astype(np.int32)}
    p = Program().group_by([Aggregate(100, AGG.COUNT, []), Aggregate(101, AGG.SUM, [2])], keys=[1])
    pc = ProgramContainer()
    ok, err = pc.Init([(1, "k"), (2, "v")], p)
    assert ok, err
    dev = pc.Import(batch)
    first = pc.Export(pc.ApplyDevice(dev))
    before = applied_by_region()
    second = pc.Export(pc.ApplyDevice(dev))
    assert applied_by_region() - before == 1
    compare_batches(second, first, key_names=["k"])
    keys, counts = np.unique(k, return_counts=True)
    order = np.argsort(second["k"].values)
    assert np.array_equal(second["k"].values[order], keys)
    assert np.array_equal(second["100"].values[order], counts.astype(np.uint64))
    sums = np.zeros(1 << 21, np.int64)
    np.add.at(sums, k, batch["v"].astype(np.int64))
    assert np.array_equal(second["101"].values[order], sums[keys])

"""GROUP BY behind a filter by the kernel that takes the rows that passed the filter one by one
(SSA_KEYED_HOT=2 forces that kernel wherever a step qualifies; by itself it is chosen by timing on large scans).  Results
must equal the CPU oracle's and those of the kernel that does not compact."""
import ctypes

import numpy as np
import pytest

from cases import compare_batches
from ydb_amd.arrow_c import Col
from ydb_amd.ssa_pb import AGG, F, Aggregate, Program

pytestmark = pytest.mark.gpu


def compacted_scans():
    from ydb_amd import _lib
    n = ctypes.c_int64(0)
    _lib.load().ssa_keyed_compact_stats(ctypes.byref(n))
    return n.value


def run_both(monkeypatch, resolver, program, batch, expect_compact=True):
    from oracle import ssa_oracle
    from ydb_amd.program import ProgramContainer
    monkeypatch.setenv("SSA_JIT", "1")
    out = []
    for way in ("2", "0"):
        monkeypatch.setenv("SSA_KEYED_HOT", way)
        pc = ProgramContainer()
        ok, err = pc.Init(resolver, program)
        assert ok, err
        dev = pc.Import(batch)
        before = compacted_scans()
        for _ in range(2):  # (the second scan runs with the table sized by the first)
            res = pc.Export(pc.ApplyDevice(dev))
        delta = compacted_scans() - before
        assert (delta >= 2) if (way == "2" and expect_compact) else (delta == 0)  # (a scan repeated with a larger table counts again)
        out.append(res)
    out.append(ssa_oracle.Oracle(resolver, program).apply(batch))
    return out


@pytest.mark.parametrize("pass_share", [0.0, 0.03, 0.5, 0.97, 1.0])
@pytest.mark.parametrize("n", [1, 63, 64, 4097, 200_003])
def test_small_key_behind_a_filter(monkeypatch, pass_share, n):
    rng = np.random.default_rng(int(1000 * pass_share) + n)
    adv = rng.integers(1, 31, n).astype(np.int16)
    adv[rng.random(n) >= pass_share] = 0
    batch = {"adv": adv, "w": rng.integers(0, 2000, n).astype(np.int16), "x": rng.integers(-(1 << 40), 1 << 40, n).astype(np.int64),
             "f": rng.random(n) * 10.0 + 1.0, "b": rng.integers(-100, 100, n).astype(np.int8), "u": rng.integers(0, 1 << 31, n).astype(np.uint32)}
    res = [(1, "adv"), (2, "w"), (3, "x"), (4, "f"), (5, "b"), (6, "u")]
    p = (Program().assign_const(50, "Int32", 0).assign_func(51, F.CMP_NOT_EQUAL, [1, 50]).filter(51)
         .group_by([Aggregate(100, AGG.COUNT, []), Aggregate(101, AGG.SUM, [2]), Aggregate(102, AGG.MIN, [3]), Aggregate(103, AGG.MAX, [3]),
                    Aggregate(104, AGG.SUM, [4]), Aggregate(105, AGG.MIN, [5]), Aggregate(106, AGG.MAX, [6]), Aggregate(107, AGG.SUM, [3])], keys=[1]))
    compact, plain, oracle = run_both(monkeypatch, res, p, batch)
    compare_batches(compact, plain, key_names=["adv"], float_sum_names=["104"])
    compare_batches(compact, oracle, key_names=["adv"], float_sum_names=["104"])


def test_two_keys_and_a_key_that_is_also_an_argument(monkeypatch):
    rng = np.random.default_rng(77)
    n = 150_001
    batch = {"a": rng.integers(0, 40, n).astype(np.int32), "b": rng.integers(0, 7, n).astype(np.uint8), "v": rng.integers(0, 1000, n).astype(np.int64),
             "d": rng.integers(15000, 15100, n).astype(np.int32)}
    res = [(1, "a"), (2, "b"), (3, "v"), (4, "d")]
    p = (Program().assign_const(50, "Int32", 15010).assign_func(51, F.CMP_LESS, [4, 50]).filter(51)
         .assign_const(52, "Int64", 500).assign_func(53, F.CMP_GREATER, [3, 52]).filter(53)
         .group_by([Aggregate(100, AGG.COUNT, []), Aggregate(101, AGG.SUM, [1]), Aggregate(102, AGG.MAX, [3]), Aggregate(103, AGG.MIN, [4])], keys=[1, 2]))
    compact, plain, oracle = run_both(monkeypatch, res, p, batch)
    compare_batches(compact, plain, key_names=["a", "b"])
    compare_batches(compact, oracle, key_names=["a", "b"])


def test_many_groups_behind_a_filter(monkeypatch):
    rng = np.random.default_rng(78)
    n = 300_000
    batch = {"k": rng.integers(0, 50_000, n).astype(np.int32), "v": rng.integers(0, 100, n).astype(np.int32)}
    p = (Program().assign_const(50, "Int32", 10).assign_func(51, F.CMP_LESS, [2, 50]).filter(51)
         .group_by([Aggregate(100, AGG.COUNT, []), Aggregate(101, AGG.SUM, [2])], keys=[1]))
    compact, plain, oracle = run_both(monkeypatch, [(1, "k"), (2, "v")], p, batch)
    compare_batches(compact, plain, key_names=["k"])
    compare_batches(compact, oracle, key_names=["k"])


def test_steps_that_do_not_qualify_run_the_other_kernels(monkeypatch):
    """Nullable arguments, computed arguments and SOME read validity bitmaps, registers or row numbers: not compacted."""
    rng = np.random.default_rng(79)
    n = 50_000
    batch = {"k": rng.integers(0, 9, n).astype(np.int16), "v": Col(rng.integers(0, 100, n).astype(np.int32), valid=rng.random(n) > 0.2),
             "w": rng.integers(0, 100, n).astype(np.int32)}
    res = [(1, "k"), (2, "v"), (3, "w")]
    flt = Program().assign_const(50, "Int32", 30).assign_func(51, F.CMP_LESS, [3, 50]).filter(51)
    for aggs in ([Aggregate(100, AGG.SUM, [2])], [Aggregate(100, AGG.SOME, [3])]):
        p = (Program().assign_const(50, "Int32", 30).assign_func(51, F.CMP_LESS, [3, 50]).filter(51).group_by(aggs, keys=[1]))
        compact, plain, oracle = run_both(monkeypatch, res, p, batch, expect_compact=False)
        compare_batches(compact, oracle, key_names=["k"])
    p = (Program().assign_const(50, "Int32", 30).assign_func(51, F.CMP_LESS, [3, 50]).filter(51).assign_func(52, F.MATH_ADD, [3, 50])
         .group_by([Aggregate(100, AGG.SUM, [52])], keys=[1]))
    compact, plain, oracle = run_both(monkeypatch, res, p, batch, expect_compact=False)
    compare_batches(compact, oracle, key_names=["k"])
    del flt


def test_shard_sized_batches_try_every_way_and_settle(monkeypatch):
    """Batches of a million rows: the step's first applies cycle through the ways of running its kernel until each has
    been timed on four million rows, then one is kept.  Whatever way runs, the result is the oracle's."""
    from oracle import ssa_oracle
    from ydb_amd.program import ProgramContainer
    monkeypatch.setenv("SSA_JIT", "1")
    monkeypatch.delenv("SSA_KEYED_HOT", raising=False)
    rng = np.random.default_rng(81)
    n = 1_100_000
    adv = rng.integers(1, 31, n).astype(np.int16)
    adv[rng.random(n) >= 0.03] = 0
    batch = {"adv": adv, "w": rng.integers(0, 2000, n).astype(np.int16)}
    res = [(1, "adv"), (2, "w")]
    p = (Program().assign_const(50, "Int32", 0).assign_func(51, F.CMP_NOT_EQUAL, [1, 50]).filter(51)
         .group_by([Aggregate(100, AGG.COUNT, []), Aggregate(101, AGG.SUM, [2]), Aggregate(102, AGG.MAX, [2])], keys=[1]))
    exp = ssa_oracle.Oracle(res, p).apply(batch)
    pc = ProgramContainer()
    ok, err = pc.Init(res, p)
    assert ok, err
    dev = pc.Import(batch)
    before = compacted_scans()
    for _ in range(16):
        compare_batches(pc.Export(pc.ApplyDevice(dev)), exp, key_names=["adv"])
    assert compacted_scans() - before >= 4  # (the third way was among those timed)

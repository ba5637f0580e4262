"""Aggregations of shard-sized batches: the launch that counts, emits and empties small group-by tables right behind the
scan (one wait on the stream per apply), and the key-less results written by the finalize launch where the output
columns hold them.  Results must equal the oracle's and those of the path with a launch per stage (SSA_NO_SMALL_EMIT=1)."""
import os

import numpy as np
import pytest

from cases import compare_batches
from test_gpu_many import LI, q1, q6, slice_of
from ydb_amd.arrow_c import Col
from ydb_amd.ssa_pb import AGG, Aggregate, Program

pytestmark = pytest.mark.gpu


def apply_all(program, batches, resolver=LI):
    from oracle import ssa_oracle
    from ydb_amd.program import ProgramContainer
    pc = ProgramContainer()
    ok, err = pc.Init(resolver, program)
    assert ok, err
    orc = ssa_oracle.Oracle(resolver, program)
    return [pc.Export(pc.ApplyDevice(pc.Import(b))) for b in batches], [orc.apply(b) for b in batches]


def staged(fn):
    os.environ["SSA_NO_SMALL_EMIT"] = "1"
    try:
        return fn()
    finally:
        del os.environ["SSA_NO_SMALL_EMIT"]


def test_one_table_serves_a_row_of_applies_with_other_keys_each():
    """The step's table is emptied by the launch that emits it: nothing of an apply may show in the next one's groups."""
    rng = np.random.default_rng(71)
    batches = []
    for i, n in enumerate([50_000, 3, 70_000, 1, 20_000, 0, 40_000]):
        b = slice_of(rng, n)
        b["flag"] = (rng.integers(0, 3, n) + 5 * i).astype(np.int8)  # other keys in every batch
        b["status"] = rng.integers(0, 2 + i, n).astype(np.uint8)
        batches.append(b)
    got, want = apply_all(q1([5, 6]), batches)
    old, _ = staged(lambda: apply_all(q1([5, 6]), batches))
    for g, w, o in zip(got, want, old):
        compare_batches(g, w, key_names=["flag", "status"], float_sum_names=["130", "131", "132"])
        compare_batches(g, o, key_names=["flag", "status"], float_sum_names=["130", "131", "132"])


def test_a_batch_of_many_groups_between_batches_of_few():
    """The table outgrows the launch (more than 8192 slots) and comes back: both ways give the groups of their batch."""
    rng = np.random.default_rng(72)
    few = lambda n: {"k": rng.integers(0, 7, n).astype(np.int64), "v": rng.integers(-1000, 1000, n).astype(np.int64)}  # noqa: E731
    many = {"k": rng.integers(0, 30_000, 120_000).astype(np.int64), "v": rng.integers(-1000, 1000, 120_000).astype(np.int64)}
    batches = [few(30_000), many, few(30_000), few(10), many, few(2_000)]
    program = Program().group_by([Aggregate(10, AGG.SUM, [2]), Aggregate(11, AGG.COUNT, []), Aggregate(12, AGG.MIN, [2]),
                                  Aggregate(13, AGG.MAX, [2])], keys=[1])
    got, want = apply_all(program, batches, resolver=[(1, "k"), (2, "v")])
    for g, w in zip(got, want):
        compare_batches(g, w, key_names=["k"])
    assert [len(g["k"].valid) for g in got] == [7, len(set(many["k"].tolist())), 7, len(set(batches[3]["k"].tolist())), len(set(many["k"].tolist())), 7]


@pytest.mark.parametrize("groups", [200, 300, 3000])
def test_up_to_a_few_thousand_groups_leave_through_the_one_launch(groups):
    """Up to 256 groups are written an aggregate per thread, more a group per thread."""
    rng = np.random.default_rng(73 + groups)
    n = 150_000
    batch = {"k": rng.integers(0, groups, n).astype(np.int32), "k2": rng.integers(0, 2, n).astype(np.uint16),
             "v": rng.integers(-1000, 1000, n).astype(np.int64), "f": rng.integers(0, 100, n) / 4.0}
    program = Program().group_by([Aggregate(10, AGG.SUM, [3]), Aggregate(11, AGG.COUNT, []), Aggregate(12, AGG.MIN, [4]),
                                  Aggregate(13, AGG.MAX, [3]), Aggregate(14, AGG.SUM, [4])], keys=[1, 2])
    resolver = [(1, "k"), (2, "k2"), (3, "v"), (4, "f")]
    for _ in range(2):  # (the second apply sizes the table by the first one's groups)
        got, want = apply_all(program, [batch, batch], resolver=resolver)
    for g, w in zip(got, want):
        compare_batches(g, w, key_names=["k", "k2"], float_sum_names=["14"])


def test_key_less_results_with_nulls_decimals_and_some():
    """Every way the finalize launch formats a result: plain, float32, null, the row of some()."""
    rng = np.random.default_rng(74)
    n = 40_000
    batch = {"a": Col(rng.integers(-50, 50, n).astype(np.int32), valid=rng.random(n) >= 0.3),
             "f": rng.integers(0, 1000, n).astype(np.float32), "never": Col(np.zeros(n, np.int64), valid=np.zeros(n, bool)),
             "u": rng.integers(0, 60000, n).astype(np.uint16)}
    program = Program().group_by([Aggregate(10, AGG.SUM, [1]), Aggregate(11, AGG.MIN, [2]), Aggregate(12, AGG.MAX, [3]),
                                  Aggregate(13, AGG.COUNT, [3]), Aggregate(14, AGG.SOME, [1]), Aggregate(15, AGG.SOME, [4]),
                                  Aggregate(16, AGG.SUM, [3]), Aggregate(17, AGG.MAX, [4])])
    resolver = [(1, "a"), (2, "f"), (3, "never"), (4, "u")]
    got, want = apply_all(program, [batch, {k: (v[:0] if not isinstance(v, Col) else Col(v.values[:0], valid=v.valid[:0])) for k, v in batch.items()}],
                          resolver=resolver)
    for g, w in zip(got, want):
        compare_batches(g, w)
    assert got[0]["12"].to_pylist() == [None] and got[0]["13"].to_pylist() == [0]


def test_tpch_programs_both_ways():
    rng = np.random.default_rng(75)
    batches = [slice_of(rng, n) for n in (100_000, 1, 2048)]
    for program, keys in ((q1([7, 8]), ["rf", "ls"]), (q1([5, 6]), ["flag", "status"]), (q6(), None)):
        got, want = apply_all(program, batches)
        old, _ = staged(lambda: apply_all(program, batches))
        sums = ["130", "131", "132"] if keys else ["130"]
        for g, w, o in zip(got, want, old):
            compare_batches(g, w, key_names=keys, float_sum_names=sums)
            compare_batches(g, o, key_names=keys, float_sum_names=sums)

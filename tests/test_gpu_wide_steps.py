"""Steps wider than one scan holds (more aggregates, computed outputs or temporaries than the kernels' fixed sizes): key-less
aggregations and projections are run as several scans and put together; results against the CPU oracle."""
import numpy as np
import pytest

from cases import compare_batches
from ydb_amd.arrow_c import Col
from ydb_amd.ssa_pb import AGG, F, Aggregate, Column, Program

pytestmark = pytest.mark.gpu

RESOLVER = [(1, "w"), (2, "x"), (3, "k")]


def batch(n, seed):
    rng = np.random.default_rng(seed)
    return {"w": Col(rng.integers(0, 3000, n).astype(np.int16), type="int16"),
            "x": Col(rng.integers(-10**6, 10**6, n).astype(np.int64), valid=rng.random(n) > 0.1, type="int64"),
            "k": Col(rng.integers(0, 7, n).astype(np.int32), type="int32")}


def both(program, data):
    from oracle import ssa_oracle
    from ydb_amd.program import ProgramContainer
    pc = ProgramContainer()
    ok, err = pc.Init(RESOLVER, program)
    assert ok, err
    first = pc.ApplyProgram(data)
    again = pc.ApplyProgram(data)  # (the second apply takes the kept split)
    exp = ssa_oracle.Oracle(RESOLVER, program).apply(data)
    return first, again, exp


@pytest.mark.parametrize("n", [1, 5000, 300_001])
def test_ninety_sums_in_one_step(n):
    """ClickBench Q29's shape: SUM(w), SUM(w + 1), ... SUM(w + 89) behind a filter."""
    p = Program().assign_const(90, "Int32", 100).assign_func(91, F.CMP_GREATER, [1, 90]).filter(91)
    aggs = [Aggregate(Column(name="s0"), AGG.SUM, [1])]
    for j in range(1, 90):
        p.assign_const(100 + j, "Int16", j)
        p.assign_func(300 + j, F.MATH_ADD, [1, 100 + j])
        aggs.append(Aggregate(Column(name="s%d" % j), AGG.SUM, [300 + j]))
    aggs.append(Aggregate(Column(name="rows"), AGG.COUNT, [2]))
    p.group_by(aggs)
    first, again, exp = both(p, batch(n, 11))
    assert list(first.keys()) == ["s%d" % j for j in range(90)] + ["rows"]
    compare_batches(first, exp)
    compare_batches(again, exp)


@pytest.mark.parametrize("n", [0, 7000, 150_000])
def test_a_projection_of_thirty_computed_columns(n):
    p = Program().assign_const(90, "Int32", 3).assign_func(91, F.CMP_LESS, [3, 90]).filter(91)
    outs = []
    for j in range(30):
        p.assign_const(100 + j, "Int64", j + 1)
        p.assign_func(200 + j, F.MATH_MULTIPLY, [2, 100 + j])
        p.assign_func(300 + j, F.MATH_ADD, [200 + j, 1])
        outs.append(300 + j)
    p.project(outs + [3])
    first, again, exp = both(p, batch(n, 12))
    compare_batches(first, exp)
    compare_batches(again, exp)


def test_a_keyed_aggregation_keeps_its_limit():
    from ydb_amd.program import ProgramContainer, SsaError
    p = Program()
    aggs = []
    for j in range(20):
        p.assign_const(100 + j, "Int64", j)
        p.assign_func(300 + j, F.MATH_ADD, [2, 100 + j])
        aggs.append(Aggregate(Column(name="s%d" % j), AGG.SUM, [300 + j]))
    p.group_by(aggs, keys=[3])
    pc = ProgramContainer()
    ok, err = pc.Init(RESOLVER, p)
    assert ok, err
    with pytest.raises(SsaError) as e:
        pc.ApplyProgram(batch(1000, 13))
    assert "too many" in str(e.value)

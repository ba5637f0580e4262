"""The operator-level seam (ssa_group_by_create: CH::GroupByOptions of the Arrow function "ch.group_by") through the C-ABI:
the group counts of the reference's own aggregator test (ydb/library/arrow_clickhouse/ut_aggregator.cpp:33-88,246-259, the
vectors of tests/golden/reference_vectors.json), and the same aggregations as a program on the CPU oracle."""
import numpy as np
import pytest

from cases import aggregator_block, compare_batches, pylist
from ydb_amd.arrow_c import Col
from ydb_amd.ssa_pb import AGG, Aggregate, Column, Program

pytestmark = pytest.mark.gpu

KEY, ANY, COUNT, MIN, MAX, SUM = 0, 1, 2, 3, 4, 5  # CH::AggFunctionId


def seam(assigns, batch):
    from ydb_amd.program import ProgramContainer
    pc = ProgramContainer()
    ok, err = pc.InitGroupBy(assigns)
    assert ok, err
    return pc.ApplyProgram(batch)


def test_group_counts_of_the_reference_aggregator_test(golden):
    case = golden["aggregator_group_counts"]
    batch = aggregator_block(case["rows"])
    for label, keys in (("keys_0_1", ["i16", "ui32"]), ("keys_0_2", ["i16", "s1"]), ("keys_2_3", ["s1", "s2"]),
                        ("keys_0_1_2_3", ["i16", "ui32", "s1", "s2"]), ("keys_0_1", ["ui32", "i16"]), ("keys_0_2", ["s1", "i16"])):
        # the count between the keys: the result has the order of the assigns
        assigns = [(KEY, keys[0], None), (COUNT, "cnt", None)] + [(KEY, k, None) for k in keys[1:]]
        out = seam(assigns, batch)
        assert list(out.keys()) == [keys[0], "cnt"] + keys[1:]
        assert len(out["cnt"].valid) == case["expect"][label], (label, keys)
        assert out["cnt"].type == "uint64"
        assert int(np.sum(out["cnt"].values)) == case["rows"]


def mixed_batch(rng, n):
    return {"k1": Col(rng.integers(0, 50, n).astype(np.int32), type="int32"),
            "k2": Col([b"x" * int(v) for v in rng.integers(0, 4, n)], type="string"),
            "v": Col(rng.integers(-1000, 1000, n).astype(np.int64), valid=rng.random(n) > 0.2, type="int64"),
            "f": Col(rng.random(n) * 100, type="double"),
            "s": Col([("s%03d" % int(v)).encode() for v in rng.integers(0, 500, n)], type="string")}


@pytest.mark.parametrize("n", [0, 1, 1000, 70_001])
def test_aggregates_equal_the_program_on_the_oracle(n):
    from oracle import ssa_oracle
    rng = np.random.default_rng(n + 5)
    batch = mixed_batch(rng, n)
    assigns = [(SUM, "sum_v", "v"), (KEY, "k1", None), (MIN, "min_f", "f"), (MAX, "max_s", "s"), (COUNT, "cnt_v", "v"), (KEY, "k2", None),
               (ANY, "any_v", "v"), (COUNT, "rows", None), (SUM, "sum_f", "f")]
    got = seam(assigns, batch)
    resolver = [(1, "k1"), (2, "k2"), (3, "v"), (4, "f"), (5, "s")]
    ids = {name: i for i, name in resolver}
    fn = {SUM: AGG.SUM, MIN: AGG.MIN, MAX: AGG.MAX, COUNT: AGG.COUNT, ANY: AGG.SOME}
    # count of rows: in a program a COUNT without argument becomes COUNT(first key) (ssa_program_optimizer.cpp:7-24),
    # which is the same number while that key has no nulls
    aggs = [Aggregate(Column(name=res), fn[f], [ids[arg]] if arg else []) for f, res, arg in assigns if f != KEY]
    p = Program().group_by(aggs, keys=[ids["k1"], ids["k2"]]).project([res for _, res, _ in assigns])
    exp = ssa_oracle.Oracle(resolver, p).apply(batch)
    if n == 0:
        assert list(got.keys()) == list(exp.keys())
        assert all(len(c.valid) == 0 for c in got.values())
        return
    compare_batches(got, exp, key_names=["k1", "k2"], float_sum_names=["sum_f"])


def test_no_keys_one_row():
    rng = np.random.default_rng(3)
    batch = mixed_batch(rng, 5000)
    out = seam([(MAX, "mx", "v"), (COUNT, "c", "v"), (SUM, "sf", "f")], batch)
    valid = batch["v"].valid
    vals = np.asarray(batch["v"].values)
    assert pylist(out["mx"]) == [int(vals[valid].max())]
    assert pylist(out["c"]) == [int(valid.sum())]
    assert abs(pylist(out["sf"])[0] - float(np.sum(batch["f"].values))) <= 1e-9 * float(np.sum(batch["f"].values))


def test_a_handle_runs_batch_after_batch():
    from ydb_amd.program import ProgramContainer
    pc = ProgramContainer()
    ok, err = pc.InitGroupBy([(KEY, "k1", None), (SUM, "t", "v")])
    assert ok, err
    for seed in range(3):
        rng = np.random.default_rng(seed)
        batch = mixed_batch(rng, 3000 + seed)
        out = pc.ApplyProgram(batch)
        exp = {}
        for k, v, ok_ in zip(batch["k1"].values, batch["v"].values, batch["v"].valid):
            exp.setdefault(int(k), 0)
            exp[int(k)] += int(v) if ok_ else 0
        assert dict(zip(pylist(out["k1"]), pylist(out["t"]))) == exp

"""ClickBench scan-aggregate programs as the reference's KQP layer pushes them down
(ydb/tests/functional/clickbench/canondata/test.test_plans_column_/queries-original-plan-column-{0,1,2,3,6,7,20}: the
`SsaProgram` nodes; column ids 6 = EventDate, 10 = UserID, 14 = URL, 21 = ResolutionWidth, 41 = AdvEngineID) over a
synthetic `hits` with the schema's types (ydb/library/workload/clickbench/click_bench_schema.sql:6-46).  GPU vs oracle."""
import numpy as np
import pytest

from cases import compare_batches
from ydb_amd.arrow_c import Col
from ydb_amd.ssa_pb import AGG, F, YQL, Aggregate, Program

pytestmark = pytest.mark.gpu

HITS = [(6, "EventDate"), (10, "UserID"), (14, "URL"), (21, "ResolutionWidth"), (41, "AdvEngineID")]


def hits(n, seed=11):
    rng = np.random.default_rng(seed)
    adv = np.where(rng.random(n) < 0.97, 0, rng.integers(1, 31, n)).astype(np.int16)
    width = rng.choice(np.array([0, 1024, 1280, 1366, 1440, 1600, 1920], dtype=np.int16), n)
    ranks = np.minimum((rng.pareto(1.1, n) * 50).astype(np.int64), max(1, n // 6))
    user = (ranks * np.int64(0x9E3779B97F4A7C15 & 0x7FFFFFFFFFFFFFFF)) ^ (ranks << 17)
    date = (15887 + rng.integers(0, 31, n)).astype(np.uint16)
    hosts = ["http://google.com/search?q=", "http://yandex.ru/", "https://example.org/a/b", "http://mail.google.com/", ""]
    url = [hosts[i] + str(int(j)) for i, j in zip(rng.integers(0, len(hosts), n), rng.integers(0, 50, n))]
    return {"EventDate": date, "UserID": user, "URL": url, "ResolutionWidth": width, "AdvEngineID": adv}


def both(program, batch):
    from oracle import ssa_oracle
    from ydb_amd.program import ProgramContainer
    pc = ProgramContainer()
    ok, err = pc.Init(HITS, program)
    assert ok, err
    return pc.ApplyProgram(batch), ssa_oracle.Oracle(HITS, program).apply(batch)


def programs():
    p = {}
    p["q0"] = Program(version=5).group_by([Aggregate(106, AGG.COUNT, [])]).project([106])
    p["q1"] = (Program(version=5).assign_const(106, "Int32", 0).assign_yql(107, YQL.NotEquals, [41, 106]).filter(107)
               .group_by([Aggregate(108, AGG.COUNT, [])]).project([108]))
    p["q2"] = (Program(version=5).group_by([Aggregate(106, AGG.SUM, [21]), Aggregate(107, AGG.COUNT, [21]), Aggregate(108, AGG.COUNT, []),
                                             Aggregate(109, AGG.SUM, [41])]).project([107, 106, 108, 109]))
    p["q3"] = Program(version=5).group_by([Aggregate(106, AGG.SUM, [10]), Aggregate(107, AGG.COUNT, [10])]).project([107, 106])
    p["q6"] = Program(version=5).group_by([Aggregate(106, AGG.MAX, [6]), Aggregate(107, AGG.MIN, [6])]).project([106, 107])
    p["q7"] = (Program(version=5).assign_const(106, "Int32", 0).assign_yql(107, YQL.NotEquals, [41, 106]).filter(107).project([41]))
    p["q20"] = (Program(version=5).assign_const(106, "Bytes", b"google").assign_yql(107, YQL.StringContains, [14, 106]).filter(107)
                .group_by([Aggregate(108, AGG.COUNT, [])]).project([108]))
    return p


@pytest.mark.parametrize("name", ["q0", "q1", "q2", "q3", "q6", "q7", "q20"])
@pytest.mark.parametrize("n", [0, 1, 100_000])
def test_pushed_down_programs(name, n):
    prog = programs()[name]
    batch = hits(n)
    if name == "q0":
        batch = {"EventDate": batch["EventDate"]}  # count(*) reads the scan's default column only
    got, exp = both(prog, batch)
    compare_batches(got, exp)


@pytest.mark.parametrize("n", [1000, 300_000])
def test_group_bys_above_the_pushdown(n):
    """The GROUP BYs of Q7 / Q15 / SearchPhrase are not pushed down at this snapshot
    (ydb/core/kqp/opt/physical/kqp_opt_phy_olap_agg.cpp:302-305); as SSA programs they are the group-by workloads of the
    scope: a small int16 key, a high-cardinality int64 key, a string key."""
    batch = hits(n)
    p = (Program().assign_const(106, "Int32", 0).assign_yql(107, YQL.NotEquals, [41, 106]).filter(107)
         .group_by([Aggregate(108, AGG.COUNT, [])], keys=[41]))
    got, exp = both(p, batch)
    compare_batches(got, exp, key_names=["AdvEngineID"])
    p = Program().group_by([Aggregate(108, AGG.COUNT, []), Aggregate(109, AGG.MAX, [21])], keys=[10])
    got, exp = both(p, batch)
    compare_batches(got, exp, key_names=["UserID"])
    p = Program().group_by([Aggregate(108, AGG.COUNT, []), Aggregate(109, AGG.MIN, [6])], keys=[14])
    got, exp = both(p, batch)
    compare_batches(got, exp, key_names=["URL"])


@pytest.mark.parametrize("name", ["cb_q1", "cb_q2", "cb_q3", "cb_q6", "cb_q20", "cb_adv", "cb_userid", "cb_phrase"])
def test_benchmark_workloads_on_the_device_generated_table(name):
    """The programs bench.py times (ydb_amd/clickbench.py), over the table its generator leaves in HBM, against the
    oracle on the same columns copied to the host."""
    from oracle import ssa_oracle
    from ydb_amd import clickbench
    from ydb_amd.program import ProgramContainer
    prog = clickbench.programs()[name]
    cols = clickbench.COLUMNS[name][0]
    table = clickbench.DeviceHits(150_000, columns=cols)
    pc = ProgramContainer()
    ok, err = pc.Init(clickbench.HITS_RESOLVER, prog)
    assert ok, err
    got = pc.Export(pc.ApplyDevice(table.batch(cols)))
    exp = ssa_oracle.Oracle(clickbench.HITS_RESOLVER, prog).apply(table.host(cols))
    keys = {"cb_adv": ["AdvEngineID"], "cb_userid": ["UserID"], "cb_phrase": ["SearchPhrase"]}.get(name)
    compare_batches(got, exp, key_names=keys) if keys else compare_batches(got, exp)
    if name == "cb_userid":
        assert len(got["UserID"].valid) > 20_000       # a high-cardinality grouping, as the workload is meant to be
    if name == "cb_phrase":
        assert len(got["SearchPhrase"].valid) > 5_000
    if name == "cb_q20":
        assert 1_000 < int(got["108"].to_pylist()[0]) < 4_000   # about one URL in 64 carries the word

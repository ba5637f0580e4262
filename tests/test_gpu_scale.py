"""Full-size checks on the device-generated lineitem (no host copy of the data exists at these sizes):
size-independent properties -- shard linearity through the merge program, row-count conservation, repeatability --
plus an oracle comparison on a prefix the CPU finishes in seconds."""
import numpy as np
import pytest

from cases import compare_batches

pytestmark = pytest.mark.gpu

ROWS = 64_000_000  # ~3 GB of Q1 columns; the benchmark itself runs the same code at 600 M rows

# Floating-point SUMs over millions of rows: the reference's keyed aggregator adds sequentially in row order
# (AggregateFunctionSum.h:31-47), so its own result carries a rounding error of up to n * 2^-53 relative (3e-10 at
# 3 M rows; measured ~4e-12).  The GPU sums in a tree-like order and is closer to the exact sum; the two are compared
# with a tolerance that covers the reference's error, and the GPU is also checked against the exact rational sum.
LARGE_SUM_RTOL = 1e-10


def _scan(prog, cols, li):
    from ydb_amd import tpch
    from ydb_amd.program import ProgramContainer
    pc = ProgramContainer()
    ok, err = pc.Init(tpch.LINEITEM_RESOLVER, prog)
    assert ok, err
    return pc.Export(pc.ApplyDevice(li.batch(cols)))


@pytest.mark.parametrize("codes", [True, False])
def test_q1_shards_merge_to_the_whole(codes):
    from ydb_amd import shard, tpch
    from ydb_amd.program import ProgramContainer
    prog = tpch.q1_program(codes)
    cols = tpch.Q1_COLUMNS_CODES if codes else tpch.Q1_COLUMNS_UTF8
    whole = _scan(prog, cols, tpch.DeviceLineitem(ROWS, utf8_flags=not codes, code_flags=codes))
    parts = []
    nshards = 4
    for i in range(nshards):
        li = tpch.DeviceLineitem(ROWS // nshards, row0=i * (ROWS // nshards), utf8_flags=not codes, code_flags=codes)
        parts.append(_scan(prog, cols, li))
    mp, mr = shard.merge_program(prog, tpch.LINEITEM_RESOLVER)
    pc = ProgramContainer()
    ok, err = pc.Init(mr, mp)
    assert ok, err
    merged = pc.ApplyProgram(shard.partial_batch(parts, prog, tpch.LINEITEM_RESOLVER))
    compare_batches(merged, whole, key_names=cols[-2:], float_sum_names=["130", "131", "132", "133", "136"], rtol=1e-12)
    # every row that passes the filter is counted exactly once
    assert int(np.sum(whole["138"].values)) == int(np.sum([np.sum(p["138"].values) for p in parts]))
    assert len(whole["138"].values) == 4


def test_q1_matches_oracle_on_a_prefix_and_counts_add_up():
    from oracle import ssa_oracle
    from ydb_amd import tpch
    from ydb_amd.program import ProgramContainer
    from ydb_amd.ssa_pb import AGG, F, Aggregate, Program
    n = 3_000_000
    li = tpch.DeviceLineitem(n)
    host = tpch.lineitem_host(0, n)
    for codes in (True, False):
        prog = tpch.q1_program(codes)
        cols = tpch.Q1_COLUMNS_CODES if codes else tpch.Q1_COLUMNS_UTF8
        got = _scan(prog, cols, li)
        exp = ssa_oracle.Oracle(tpch.LINEITEM_RESOLVER, prog).apply({c: host[c] for c in cols})
        compare_batches(got, exp, key_names=cols[-2:], float_sum_names=["130", "131", "132", "133", "136"], rtol=LARGE_SUM_RTOL)
        # l_discount is a whole number of hundredths: its exact per-group sum is an integer / 100
        sel = host["l_shipdate"] <= tpch.Q1_DATE
        hundredths = np.rint(host["l_discount"] * 100).astype(np.int64)
        k1 = np.asarray(host[cols[-2]])[sel]
        k2 = np.asarray(host[cols[-1]])[sel]
        for r in range(len(got["136"].values)):
            a = got[cols[-2]].values[r]
            b = got[cols[-1]].values[r]
            a = a.decode() if isinstance(a, bytes) else a
            b = b.decode() if isinstance(b, bytes) else b
            exact = int(hundredths[sel][(k1 == a) & (k2 == b)].sum()) / 100.0
            assert abs(got["136"].values[r] - exact) <= 1e-13 * exact, (got["136"].values[r], exact)
    # rows passing Q1's filter, counted by a key-less program, equal the sum of the group counts
    p = (Program().assign_const(100, "Int32", tpch.Q1_DATE).assign_func(110, F.CMP_LESS_EQUAL, [tpch.L_SHIPDATE, 100]).filter(110)
         .group_by([Aggregate(120, AGG.COUNT, [])]))
    total = _scan(p, ["l_shipdate"], li)
    assert int(total["120"].values[0]) == int(np.sum(got["138"].values)) == int((host["l_shipdate"] <= tpch.Q1_DATE).sum())


def test_q6_full_size_is_repeatable_and_linear():
    from ydb_amd import tpch
    prog = tpch.q6_program()
    li = tpch.DeviceLineitem(ROWS, utf8_flags=False, code_flags=False)
    a = _scan(prog, tpch.Q6_COLUMNS, li)["130"].values[0]
    b = _scan(prog, tpch.Q6_COLUMNS, li)["130"].values[0]
    assert a == b  # fixed grid, fixed reduction order: bit-identical
    halves = []
    for i in range(2):
        h = tpch.DeviceLineitem(ROWS // 2, row0=i * (ROWS // 2), utf8_flags=False, code_flags=False)
        halves.append(_scan(prog, tpch.Q6_COLUMNS, h)["130"].values[0])
    assert abs((halves[0] + halves[1]) - a) <= 1e-12 * abs(a)


def test_high_cardinality_group_by_at_scale():
    """16 M rows, ~4 M groups keyed by the price column's bit pattern: group counts add up to the row count and the
    per-group sums add up to the key-less sum (table growth / retry path included)."""
    from ydb_amd import tpch
    from ydb_amd.ssa_pb import AGG, Aggregate, Program
    n = 16_000_000
    li = tpch.DeviceLineitem(n, utf8_flags=False, code_flags=False)
    p = Program().group_by([Aggregate(120, AGG.COUNT, []), Aggregate(121, AGG.SUM, [tpch.L_QUANTITY])], keys=[tpch.L_EXTENDEDPRICE])
    out = _scan(p, ["l_extendedprice", "l_quantity"], li)
    assert int(np.sum(out["120"].values)) == n
    keys = out["l_extendedprice"].values
    assert len(np.unique(keys)) == len(keys)
    total = _scan(Program().group_by([Aggregate(130, AGG.SUM, [tpch.L_QUANTITY])]), ["l_quantity"], li)["130"].values[0]
    assert abs(float(np.sum(out["121"].values)) - total) <= 1e-9 * total


def test_mixed_q1_q6_stream_over_one_resident_shard():
    """The stream of the sharded SF1000 configuration at 64 M rows: Q1 and Q6 alternate over one resident table, each
    with its own program handle; every result of the stream equals the result of a single run in a fresh handle."""
    from ydb_amd import tpch
    from ydb_amd.program import ProgramContainer
    li = tpch.DeviceLineitem(ROWS, utf8_flags=True, code_flags=False)
    q1, q6 = tpch.q1_program(False), tpch.q6_program()
    want1 = _scan(q1, tpch.Q1_COLUMNS_UTF8, li)
    want6 = _scan(q6, tpch.Q6_COLUMNS, li)
    pc1, pc6 = ProgramContainer(), ProgramContainer()
    assert pc1.Init(tpch.LINEITEM_RESOLVER, q1)[0] and pc6.Init(tpch.LINEITEM_RESOLVER, q6)[0]
    b1, b6 = li.batch(tpch.Q1_COLUMNS_UTF8), li.batch(tpch.Q6_COLUMNS)
    for _ in range(4):
        got1 = pc1.Export(pc1.ApplyDevice(b1))
        got6 = pc6.Export(pc6.ApplyDevice(b6))
        compare_batches(got1, want1, key_names=["l_returnflag", "l_linestatus"], float_sum_names=["130", "131", "132", "133", "136"], rtol=1e-12)
        assert got6["130"].values[0] == want6["130"].values[0]   # key-less: fixed grid and reduction order


@pytest.mark.parametrize("rows", [600_037_902, 750_000_000])
def test_full_shard_sizes(rows):
    """TPC-H SF100 (one GPU) and SF1000 / 8 (one GPU of eight) as one resident batch: the properties that need no host copy
    of the data -- group counts add up to the key-less count of the filter, two half shards merge to the whole, Q6 is
    the sum of its halves, the result is repeatable."""
    from ydb_amd import shard, tpch
    from ydb_amd.program import ProgramContainer
    from ydb_amd.ssa_pb import AGG, F, Aggregate, Program
    sums = ["130", "131", "132", "133", "136"]
    keys = ["l_returnflag", "l_linestatus"]
    q1, q6 = tpch.q1_program(False), tpch.q6_program()
    li = tpch.DeviceLineitem(rows, utf8_flags=True, code_flags=False)
    whole1 = _scan(q1, tpch.Q1_COLUMNS_UTF8, li)
    again1 = _scan(q1, tpch.Q1_COLUMNS_UTF8, li)
    compare_batches(again1, whole1, key_names=keys, float_sum_names=sums, rtol=1e-12)
    whole6 = _scan(q6, tpch.Q6_COLUMNS, li)["130"].values[0]
    assert _scan(q6, tpch.Q6_COLUMNS, li)["130"].values[0] == whole6
    p = (Program().assign_const(100, "Int32", tpch.Q1_DATE).assign_func(110, F.CMP_LESS_EQUAL, [tpch.L_SHIPDATE, 100]).filter(110)
         .group_by([Aggregate(120, AGG.COUNT, [])]))
    passing = int(_scan(p, ["l_shipdate"], li)["120"].values[0])
    assert int(np.sum(whole1["138"].values)) == passing and 0 < passing < rows
    assert len(whole1["138"].values) == 4
    del li
    half = rows // 2
    parts, q6_halves = [], []
    for row0, n in ((0, half), (half, rows - half)):
        h = tpch.DeviceLineitem(n, row0=row0, utf8_flags=True, code_flags=False)
        parts.append(_scan(q1, tpch.Q1_COLUMNS_UTF8, h))
        q6_halves.append(_scan(q6, tpch.Q6_COLUMNS, h)["130"].values[0])
        del h
    mp, mr = shard.merge_program(q1, tpch.LINEITEM_RESOLVER)
    pc = ProgramContainer()
    assert pc.Init(mr, mp)[0]
    merged = pc.ApplyProgram(shard.partial_batch(parts, q1, tpch.LINEITEM_RESOLVER))
    compare_batches(merged, whole1, key_names=keys, float_sum_names=sums, rtol=1e-12)
    for name in ("134", "135", "137", "138"):   # counts are exact
        assert sorted(int(v) for v in merged[name].values) == sorted(int(v) for v in whole1[name].values)
    assert abs((q6_halves[0] + q6_halves[1]) - whole6) <= 1e-12 * abs(whole6)


def test_q6_at_sf1_matches_the_oracle():
    """TPC-H Q6 over an SF1 lineitem (6 001 215 rows, the size the reference's own CPU-runnable configuration names):
    the whole table in one batch on the device, the oracle over the same rows on the host."""
    from oracle import ssa_oracle
    from ydb_amd import tpch
    n = 6_001_215
    got = _scan(tpch.q6_program(), tpch.Q6_COLUMNS, tpch.DeviceLineitem(n, utf8_flags=False, code_flags=False))
    host = tpch.lineitem_host(0, n)
    exp = ssa_oracle.Oracle(tpch.LINEITEM_RESOLVER, tpch.q6_program()).apply({c: host[c] for c in tpch.Q6_COLUMNS})
    assert list(got.keys()) == list(exp.keys()) == ["130"]
    assert got["130"].type == exp["130"].type
    g, e = got["130"].to_pylist(), exp["130"].to_pylist()
    assert len(g) == len(e) == 1
    # (the reference sums pairwise, the GPU tile by tile: both within a few ulps of the exact sum of ~1.1e5 terms)
    assert abs(g[0] - e[0]) <= 1e-12 * abs(e[0]), (g, e)

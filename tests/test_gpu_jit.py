"""Scan kernels built at run time for one step program (ydb_amd/csrc/jit.h): the same checks as the generic kernels
pass, against the oracle, when every scan is forced through a kernel compiled for its program (SSA_JIT=1); the
policy counters move as documented; SSA_JIT=0 keeps everything on the generic kernels."""
import os

import numpy as np
import pytest

import test_gpu_parity as P
from cases import compare_batches

pytestmark = pytest.mark.gpu


def _stats():
    from ydb_amd.program import ProgramContainer
    return ProgramContainer().SpecializationStats()


@pytest.fixture
def forced(monkeypatch):
    monkeypatch.setenv("SSA_JIT", "1")
    before = _stats()
    yield before
    after = _stats()
    assert after[2] > before[2], "no launch used a kernel built for its program"


@pytest.mark.parametrize("n", [1, 1025, 100_003])
@pytest.mark.parametrize("nulls", [0.0, 0.1])
def test_keyless_aggregation(forced, n, nulls):
    P.test_filter_sum_nokeys(n, nulls)


@pytest.mark.parametrize("n", [1, 4097, 100_003])
@pytest.mark.parametrize("nulls", [0.0, 0.1])
def test_group_by_small_keys(forced, n, nulls):
    P.test_filter_group_by_small_keys(n, nulls)


def test_group_by_single_byte_key(forced):
    P.test_group_by_single_byte_key_ignores_argument_nulls(50_000)


@pytest.mark.parametrize("n", [5, 70_000])
def test_group_by_string_keys(forced, n):
    P.test_group_by_string_keys(n)


def test_group_by_long_and_null_strings(forced):
    P.test_group_by_long_and_null_strings()


def test_group_by_high_cardinality(forced):
    P.test_group_by_high_cardinality(200_000)


def test_projection_with_filter_and_strings(forced):
    P.test_projection_with_filter_and_strings(70_001)


def test_multi_step_program(forced):
    P.test_multi_step_program()


def test_errors_match(forced):
    P.test_errors_match()


def test_tpch_q1_and_q6_on_the_generated_lineitem(forced):
    from oracle import ssa_oracle
    from ydb_amd import tpch
    from ydb_amd.program import ProgramContainer
    n = 500_000
    li = tpch.DeviceLineitem(n)
    host = tpch.lineitem_host(0, n)
    for prog, cols, keys, sums in (
            (tpch.q1_program(True), tpch.Q1_COLUMNS_CODES, tpch.Q1_COLUMNS_CODES[-2:], ["130", "131", "132", "133", "136"]),
            (tpch.q1_program(False), tpch.Q1_COLUMNS_UTF8, tpch.Q1_COLUMNS_UTF8[-2:], ["130", "131", "132", "133", "136"]),
            (tpch.q6_program(), tpch.Q6_COLUMNS, [], None)):
        pc = ProgramContainer()
        ok, err = pc.Init(tpch.LINEITEM_RESOLVER, prog)
        assert ok, err
        got = pc.Export(pc.ApplyDevice(li.batch(cols)))
        exp = ssa_oracle.Oracle(tpch.LINEITEM_RESOLVER, prog).apply({c: host[c] for c in cols})
        if sums is None:
            sums = list(got.keys())
        compare_batches(got, exp, key_names=keys, float_sum_names=sums, rtol=1e-11)


def test_a_kernel_is_built_once_per_program(monkeypatch):
    monkeypatch.setenv("SSA_JIT", "1")
    rng = np.random.default_rng(7)
    batch = P.lineitem_like(rng, 5000)
    # a program no other test uses (its constant is part of the kernel)
    from ydb_amd.program import ProgramContainer
    from ydb_amd.ssa_pb import AGG, F, Aggregate, Program
    p = (Program().assign_const(100, "Int32", 10463).assign_func(110, F.CMP_LESS_EQUAL, [1, 100]).filter(110)
         .group_by([Aggregate(130, AGG.SUM, [3]), Aggregate(131, AGG.COUNT, [])]))
    pc = ProgramContainer()
    ok, err = pc.Init(P.LI, p)
    assert ok, err
    b0 = _stats()
    pc.ApplyProgram(batch)
    b1 = _stats()
    pc.ApplyProgram(batch)
    b2 = _stats()
    assert b1[0] == b0[0] + 1 and b1[1] > b0[1]
    assert b2[0] == b1[0] and b2[2] == b1[2] + 1


def test_small_scans_and_ssa_jit_0_stay_on_the_generic_kernels(monkeypatch):
    from ydb_amd.ssa_pb import AGG, F, Aggregate, Program
    rng = np.random.default_rng(8)
    batch = P.lineitem_like(rng, 5000)
    # a program of a shape no other test uses (a kernel that exists already is used whatever the size of the scan, and a
    # kernel is built for a program's shape: other literals would not make it another program)
    p = (Program().assign_const(100, "Int32", 10464).assign_func(110, F.CMP_LESS_EQUAL, [1, 100]).filter(110)
         .assign_func(111, F.CMP_GREATER, [1, 100]).assign_func(112, F.CMP_NOT_EQUAL, [110, 111]).filter(112)
         .group_by([Aggregate(130, AGG.SUM, [3]), Aggregate(131, AGG.COUNT, []), Aggregate(132, AGG.MAX, [3]), Aggregate(133, AGG.COUNT, [3])]))
    for mode in (None, "0"):
        if mode is None:
            monkeypatch.delenv("SSA_JIT", raising=False)
        else:
            monkeypatch.setenv("SSA_JIT", mode)
        before = _stats()
        got, exp = P.both(P.LI, p, batch)
        compare_batches(got, exp, float_sum_names=["130"])
        assert _stats() == before


@pytest.mark.parametrize("n", [1, 4095, 4096, 4097, 16_385, 300_007])
def test_narrow_keyless_aggregates_take_wide_tiles(forced, n):
    """Key-less aggregates over a few narrow columns are built with 8 or 16 rows per thread (tiles of 2048 / 4096 rows):
    sums, counts, min / max and a filter over int16 / int32 / int64 columns without nulls, against the oracle."""
    from oracle import ssa_oracle
    from ydb_amd.program import ProgramContainer
    from ydb_amd.ssa_pb import AGG, YQL, Aggregate, Program
    rng = np.random.default_rng(n)
    batch = {"a": rng.integers(-3, 30, n).astype(np.int16), "w": rng.integers(0, 2000, n).astype(np.int16),
             "d": rng.integers(15000, 16000, n).astype(np.int32), "u": rng.integers(-2**62, 2**62, n).astype(np.int64)}
    resolver = [(1, "a"), (2, "w"), (3, "d"), (4, "u")]
    progs = [
        (Program(version=5).assign_const(10, "Int32", 0).assign_yql(11, YQL.NotEquals, [1, 10]).filter(11)
         .group_by([Aggregate(12, AGG.COUNT, [])]).project([12]), ["a"]),                                   # 2 bytes per row: 16 rows per thread
        (Program().group_by([Aggregate(10, AGG.SUM, [2]), Aggregate(11, AGG.COUNT, [2]), Aggregate(12, AGG.SUM, [1])]), ["a", "w"]),
        (Program().group_by([Aggregate(10, AGG.MIN, [3]), Aggregate(11, AGG.MAX, [3])]), ["d"]),
        (Program().group_by([Aggregate(10, AGG.SUM, [4]), Aggregate(11, AGG.MAX, [4]), Aggregate(12, AGG.MIN, [1])]), ["a", "u"]),  # 10 bytes: 8 rows
        (Program().assign_func(10, 14, [1, 2]).group_by([Aggregate(11, AGG.SUM, [10]), Aggregate(12, AGG.COUNT, [])]), ["a", "w"]),  # a temporary: 8 rows
    ]
    for prog, cols in progs:
        b = {c: batch[c] for c in cols}
        pc = ProgramContainer()
        ok, err = pc.Init(resolver, prog)
        assert ok, err
        compare_batches(pc.ApplyProgram(b), ssa_oracle.Oracle(resolver, prog).apply(b))


@pytest.mark.parametrize("n", [1, 2047, 2049, 250_003])
def test_narrow_group_by_takes_wide_tiles(forced, n):
    """A filter and a small integer key over narrow columns: the keyed kernel is built with 8 rows per thread."""
    from oracle import ssa_oracle
    from ydb_amd.program import ProgramContainer
    from ydb_amd.ssa_pb import AGG, YQL, Aggregate, Program
    rng = np.random.default_rng(n + 3)
    batch = {"a": np.where(rng.random(n) < 0.9, 0, rng.integers(1, 31, n)).astype(np.int16), "w": rng.integers(0, 2000, n).astype(np.int16)}
    resolver = [(1, "a"), (2, "w")]
    progs = [
        Program().assign_const(10, "Int32", 0).assign_yql(11, YQL.NotEquals, [1, 10]).filter(11).group_by([Aggregate(12, AGG.COUNT, [])], keys=[1]),
        Program().group_by([Aggregate(12, AGG.COUNT, []), Aggregate(13, AGG.SUM, [2]), Aggregate(14, AGG.MAX, [2])], keys=[1]),
        Program().group_by([Aggregate(12, AGG.COUNT, []), Aggregate(13, AGG.MIN, [1])], keys=[2]),   # 2000 groups
    ]
    for prog, key in zip(progs, ("a", "a", "w")):
        pc = ProgramContainer()
        ok, err = pc.Init(resolver, prog)
        assert ok, err
        compare_batches(pc.ApplyProgram(batch), ssa_oracle.Oracle(resolver, prog).apply(batch), key_names=[key])


def test_general_like_patterns(forced):
    P.test_general_like_patterns()


def test_substring_search_over_whole_tiles(forced):
    P.test_substring_search_over_whole_tiles(4096 + 17)


def test_built_kernel_is_the_specialised_one(monkeypatch):
    """A kernel built for a step program must be that program's straight-line code, not the interpreter compiled once
    more (what a too-small unrolling limit of the run-time compiler silently leaves behind): at 32 M rows the built
    TPC-H Q1 kernel has measured 2.5-3x faster than the generic one; anything under 1.5x means the specialisation is lost."""
    from ydb_amd import tpch
    from ydb_amd.program import ProgramContainer
    n = 32_000_000
    li = tpch.DeviceLineitem(n, utf8_flags=True, code_flags=False)
    times = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("SSA_JIT", mode)
        pc = ProgramContainer()
        ok, err = pc.Init(tpch.LINEITEM_RESOLVER, tpch.q1_program(False))
        assert ok, err
        batch = li.batch(tpch.Q1_COLUMNS_UTF8)
        for _ in range(2):
            pc.ApplyDevice(batch)
        pc.EnableProfile(True)
        pc.ReadProfile()
        for _ in range(5):
            pc.ApplyDevice(batch)
        ms, launches, _ = pc.ReadProfile()
        times[mode] = ms / launches
    assert times["1"] * 1.5 < times["0"], times


@pytest.mark.parametrize("jit", ["0", "1"])
def test_a_multiply_and_an_add_are_rounded_separately(jit, monkeypatch):
    """Two instructions of a step program stay two roundings in a kernel built for it: a * b + c equals the oracle's
    bit for bit (a fused multiply-add differs in the last bit on about a quarter of random rows)."""
    from ydb_amd.ssa_pb import F, Program
    monkeypatch.setenv("SSA_JIT", jit)
    rng = np.random.default_rng(1)
    n = 200_000
    batch = {"a": rng.normal(0, 1, n), "b": rng.normal(0, 1, n), "c": rng.normal(0, 1, n),
             "f": rng.normal(0, 1, n).astype(np.float32), "g": rng.normal(0, 1, n).astype(np.float32)}
    res = [(1, "a"), (2, "b"), (3, "c"), (4, "f"), (5, "g")]
    p = (Program(version=5).assign_func(100, F.MATH_MULTIPLY, [1, 2]).assign_func(101, F.MATH_ADD, [100, 3])
         .assign_func(102, F.MATH_SUBTRACT, [3, 100]).assign_yql(103, 5, [2, 3]).assign_yql(104, 4, [103, 1])
         .assign_func(105, F.MATH_MULTIPLY, [4, 5]).assign_func(106, F.MATH_ADD, [105, 4])
         .project([101, 102, 104, 106]))
    got, exp = P.both(res, p, batch)
    compare_batches(got, exp)


def test_a_query_that_comes_back_with_other_literals_runs_the_kernel_built_for_its_shape(monkeypatch):
    """Q6 with twenty sets of substitution parameters: the constants of a program are data of a launch, not part of the
    kernel built for it (they are data in the reference too: ydb/core/tx/program/program.cpp:253-295), so one kernel is
    built and every result equals the oracle's for its own literals."""
    from oracle import ssa_oracle
    from ydb_amd import tpch
    from ydb_amd.program import ProgramContainer
    monkeypatch.setenv("SSA_JIT", "1")
    n = 300_000
    host = tpch.lineitem_host(0, n)
    batch = {c: host[c] for c in tpch.Q6_COLUMNS}
    rng = np.random.default_rng(77)
    built0, _, launches0 = _stats()
    results = set()
    for i in range(20):
        lo = int(8036 + rng.integers(0, 2000))
        args = dict(date_lo=lo, date_hi=lo + int(rng.integers(30, 400)), discount=float(rng.integers(2, 10)) / 100.0,
                    quantity=float(rng.integers(10, 50)))
        prog = tpch.q6_program(**args)
        pc = ProgramContainer()
        ok, err = pc.Init(tpch.LINEITEM_RESOLVER, prog)
        assert ok, err
        got = pc.ApplyProgram(batch)
        exp = ssa_oracle.Oracle(tpch.LINEITEM_RESOLVER, prog).apply(batch)
        g, e = got["130"].to_pylist()[0], exp["130"].to_pylist()[0]
        assert (g is None) == (e is None), (args, g, e)
        if e is not None:
            assert abs(g - e) <= 1e-12 * abs(e), (args, g, e)
        results.add(e)
    built1, _, launches1 = _stats()
    assert len(results) > 10  # (the literals did change the answer)
    assert launches1 - launches0 >= 20
    assert built1 - built0 <= 1, "a kernel was built per literal"


def test_patterns_of_one_length_share_a_kernel(monkeypatch):
    """The bytes of a LIKE pattern are data of the launch as well; its length shapes the code."""
    from oracle import ssa_oracle
    from ydb_amd.program import ProgramContainer
    from ydb_amd.ssa_pb import AGG, F, Aggregate, Program
    monkeypatch.setenv("SSA_JIT", "1")
    rng = np.random.default_rng(78)
    words = ["google", "yandex", "amazon", "github", "reddit", "zzzzzz"]
    urls = ["http://%s.%s/%d" % (words[int(rng.integers(0, 5))], "com" if rng.random() < 0.5 else "org", int(rng.integers(0, 1000))) for _ in range(50_000)]
    batch = {"url": urls}
    res = [(1, "url")]
    built0, _, _ = _stats()
    counts = []
    for w in words:
        prog = (Program(version=5).assign_const(10, "Bytes", w.encode()).assign_func(11, F.STR_MATCH, [1, 10]).filter(11)
                .group_by([Aggregate(20, AGG.COUNT, [])]).project([20]))
        pc = ProgramContainer()
        ok, err = pc.Init(res, prog)
        assert ok, err
        got = pc.ApplyProgram(batch)
        exp = ssa_oracle.Oracle(res, prog).apply(batch)
        assert got["20"].to_pylist() == exp["20"].to_pylist()
        counts.append(got["20"].to_pylist()[0])
    built1, _, _ = _stats()
    assert counts[-1] == 0 and min(counts[:-1]) > 1000
    assert built1 - built0 <= 1, "a kernel was built per pattern"


@pytest.mark.parametrize("dtype", [np.int16, np.uint8, np.int8])
@pytest.mark.parametrize("n", [1, 8191, 8192, 8193, 300_001])
@pytest.mark.parametrize("wide", ["16", "32"])
def test_rows_of_one_and_two_bytes_in_wide_tiles(forced, monkeypatch, dtype, n, wide):
    """A batch of one narrow column is scanned 16 rows per thread by the kernels built for its programs (32 with
    SSA_WIDE_U32: tiles of 8192 rows, every bit of the row masks in use): the key-less count / sum / extremes behind a
    filter, and the filtered GROUP BY on the column every way the keyed kernel runs (register-held groups, plain, the rows
    behind the filter one by one)."""
    from ydb_amd.ssa_pb import AGG, YQL, Aggregate, Program
    if wide == "32":
        monkeypatch.setenv("SSA_WIDE_U32", "1")
    rng = np.random.default_rng(n)
    info = np.iinfo(dtype)
    x = np.where(rng.random(n) < 0.9, 0, rng.integers(max(info.min, -40), min(info.max, 90), n)).astype(dtype)
    res = [(1, "x")]
    zero = {"int16": "Int16", "uint8": "Uint8", "int8": "Int8"}[np.dtype(dtype).name]
    nokey = (Program(version=5).assign_const(100, zero, 0).assign_yql(101, YQL.NotEquals, [1, 100]).filter(101)
             .group_by([Aggregate(110, AGG.COUNT, []), Aggregate(111, AGG.SUM, [1]), Aggregate(112, AGG.MIN, [1]), Aggregate(113, AGG.MAX, [1])]))
    got, exp = P.both(res, nokey, {"x": x})
    compare_batches(got, exp)
    keyed = (Program(version=5).assign_const(100, zero, 0).assign_yql(101, YQL.NotEquals, [1, 100]).filter(101)
             .group_by([Aggregate(110, AGG.COUNT, []), Aggregate(111, AGG.SUM, [1])], keys=[1]))
    for hot in ("0", "1", "2"):
        monkeypatch.setenv("SSA_KEYED_HOT", hot)
        got, exp = P.both(res, keyed, {"x": x})
        compare_batches(got, exp, key_names=["x"])

"""Host-side logic that needs no GPU: the C-ABI library loads and exports every declared symbol, programs parse the way
the reference's builder parses them, the scan planner explains its plans, and the oracle's parser agrees."""
import os
import re

import numpy as np
import pytest

from ydb_amd import _lib, tpch
from ydb_amd.arrow_c import Col, HostBatch
from ydb_amd.program import ProgramContainer, SsaError
from ydb_amd.ssa_pb import AGG, F, YQL, Aggregate, Column, Program, olap_program

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols(header="ydb_ssa_hip.h"):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    return sorted(set(re.findall(r"\b(ssa_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    names = declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(_lib.EXPORTED_SYMBOLS) == names
    assert lib.ssa_version().decode().startswith("ydb-ssa-hip")
    # the benchmark-support library and its header
    bench = _lib.load_bench()
    assert declared_symbols("ydb_ssa_bench.h") == sorted(_lib.BENCH_EXPORTED_SYMBOLS)
    for n in _lib.BENCH_EXPORTED_SYMBOLS:
        assert hasattr(bench, n), n
        assert not hasattr(lib, n), n  # (the executor's library carries no benchmark code)


def container(program, resolver=tpch.LINEITEM_RESOLVER):
    pc = ProgramContainer()
    ok, err = pc.Init(resolver, program)
    return pc, ok, err


def test_parse_q6_and_source_columns():
    pc, ok, err = container(tpch.q6_program())
    assert ok, err
    assert pc.StepCount() == 2  # GroupBy closes a step, the trailing Projection is a step of its own
    assert sorted(pc.GetSourceColumns()) == ["l_discount", "l_extendedprice", "l_quantity", "l_shipdate"]
    dbg = pc.DebugString()
    assert "l_shipdate" in dbg and "130" in dbg


def test_oracle_parser_agrees_on_steps_and_sources():
    from oracle import ssa_oracle
    for prog in (tpch.q6_program(), tpch.q1_program(False), tpch.q1_program(True)):
        pc, ok, err = container(prog)
        assert ok, err
        o = ssa_oracle.Oracle(tpch.LINEITEM_RESOLVER, prog)
        assert o.step_count() == pc.StepCount()
        assert sorted(o.source_columns()) == sorted(pc.GetSourceColumns())


def test_parse_errors_are_soft():
    # unknown function id, GroupBy without aggregates, garbage bytes: Init returns (False, message), no exception
    pc, ok, err = container(Program().assign_func(10, 99, [L for L in (tpch.L_SHIPDATE,)]))
    assert not ok and err
    pc, ok, err = container(Program().group_by([], keys=[tpch.L_SHIPDATE]))
    assert not ok and err
    pc = ProgramContainer()
    ok, err = pc.Init(tpch.LINEITEM_RESOLVER, b"\xff\xff\xff\xff\x01garbage")
    assert not ok and err
    # CAST_TO_BINARY is not mapped by the reference's parser either
    pc, ok, err = container(Program().assign_func(10, F.CAST_TO_BINARY, [tpch.L_SHIPDATE]))
    assert not ok


def test_generated_names_and_name_masks_id():
    p = Program().assign_func(None, F.STR_LENGTH, ["l_returnflag"]).project(["0"])
    pc, ok, err = container(p)
    assert ok, err
    assert pc.GetSourceColumns() == ["l_returnflag"]
    p = Program().assign_func(Column(id=77), F.MATH_ADD, [tpch.L_QUANTITY, tpch.L_TAX]).project([77])
    pc, ok, err = container(p)
    assert ok, err
    assert "77" in pc.DebugString()


def test_early_filter_columns():
    pc, ok, err = container(tpch.q6_program())
    assert ok, err
    # the first step is not filter-only (it aggregates), so there is no early filter
    assert pc.GetEarlyFilterColumns() == []
    p = (Program().assign_const(100, "Int32", 9000).assign_func(110, F.CMP_GREATER, [tpch.L_SHIPDATE, 100]).filter(110)
         .project([tpch.L_SHIPDATE, tpch.L_QUANTITY]))
    pc, ok, err = container(p)
    assert ok, err
    assert pc.StepCount() == 1


def lineitem_schema_batch(cols):
    li = tpch.lineitem_host(0, 4)
    return HostBatch({c: li[c] for c in cols})


def test_explain_plans_without_a_gpu():
    pc, ok, err = container(tpch.q6_program())
    assert ok, err
    text = pc.Explain(lineitem_schema_batch(tpch.Q6_COLUMNS))
    assert "AGG_NOKEY" in text and "double" in text
    pc, ok, err = container(tpch.q1_program(False))
    assert ok, err
    text = pc.Explain(lineitem_schema_batch(tpch.Q1_COLUMNS_UTF8))
    assert "AGG_KEYED" in text
    # type errors surface at plan time: a filter over a double column
    pc, ok, err = container(Program().filter(tpch.L_QUANTITY).project([tpch.L_QUANTITY]))
    assert ok, err
    with pytest.raises(SsaError):
        pc.Explain(lineitem_schema_batch(["l_quantity"]))


def test_olap_wrapper_and_raw_program_bytes():
    raw = tpch.q6_program().encode()
    pc = ProgramContainer()
    ok, err = pc.Init(tpch.LINEITEM_RESOLVER, olap_program(raw), olap_wrapper=True)
    assert ok, err
    pc2 = ProgramContainer()
    ok, err = pc2.Init(tpch.LINEITEM_RESOLVER, raw, olap_wrapper=False)
    assert ok, err
    assert pc.DebugString() == pc2.DebugString()


def test_generator_restatement_is_deterministic_and_in_range():
    a = tpch.lineitem_host(1000, 5000)
    b = tpch.lineitem_host(0, 6000)
    for k in ("l_shipdate", "l_quantity", "l_extendedprice", "l_discount", "l_tax", "l_returnflag_code"):
        assert (np.asarray(a[k]) == np.asarray(b[k])[1000:]).all(), k
    assert a["l_shipdate"].min() >= 8036 and a["l_shipdate"].max() <= 10561
    assert set(a["l_returnflag"]) <= {"A", "N", "R"} and set(a["l_linestatus"]) <= {"F", "O"}
    assert a["l_discount"].max() <= 0.10 + 1e-12 and a["l_quantity"].min() >= 1


def test_product_refuses_to_run_without_a_gpu():
    """No CPU fallback: applying a program on a box without a GPU is an error, never a silently computed result."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    pc, ok, err = container(tpch.q6_program())
    assert ok, err
    li = tpch.lineitem_host(0, 16)
    with pytest.raises(SsaError):
        pc.ApplyProgram({c: li[c] for c in tpch.Q6_COLUMNS})


def test_mutated_program_bytes_never_crash_the_parsers():
    """Truncated, bit-flipped and spliced serialized programs: the product's parser answers ok / not ok, the oracle's
    parses or raises -- neither reads out of bounds or aborts (a crash would take the test process down)."""
    import random
    from oracle import ssa_oracle
    from ydb_amd import clickbench, tpch
    from ydb_amd.program import ProgramContainer
    from ydb_amd.ssa_pb import olap_program
    rnd = random.Random(1234)
    seeds = [olap_program(p) for p in clickbench.programs().values()]
    seeds += [olap_program(tpch.q1_program()), olap_program(tpch.q6_program())]
    resolver = clickbench.HITS_RESOLVER + [(100 + i, n) for i, n in enumerate(["l_quantity", "l_shipdate", "l_discount"])]
    tried = 0
    for data in seeds:
        for _ in range(150):
            b = bytearray(data)
            kind = rnd.randrange(4)
            if kind == 0 and len(b) > 1:
                del b[rnd.randrange(1, len(b)):]
            elif kind == 1:
                for _ in range(rnd.randrange(1, 4)):
                    b[rnd.randrange(len(b))] ^= 1 << rnd.randrange(8)
            elif kind == 2:
                at = rnd.randrange(len(b))
                b[at:at] = bytes(rnd.randrange(256) for _ in range(rnd.randrange(1, 9)))
            else:
                at, n = rnd.randrange(len(b)), rnd.randrange(1, 6)
                b[at:at + n] = bytes([0xFF] * n)   # long varints / huge lengths
            for flags in (True, False):
                ok, err = ProgramContainer().Init(resolver, bytes(b), olap_wrapper=flags)
                assert ok or err
                try:
                    ssa_oracle.Oracle(resolver, bytes(b), olap_wrapper=flags)
                except ssa_oracle.OracleError:
                    pass
                tried += 1
    assert tried >= 2000


def test_group_by_seam_builds_a_program_of_one_aggregation():
    """ssa_group_by_create (CH::GroupByOptions): what it accepts, what it refuses and the program it makes."""
    from ydb_amd.program import ProgramContainer
    pc = ProgramContainer()
    ok, err = pc.InitGroupBy([(5, "total", "v"), (0, "k", None)])
    assert ok, err
    assert pc.StepCount() == 1
    assert sorted(pc.GetSourceColumns()) == ["k", "v"]
    ok, err = pc.InitGroupBy([(0, "k", None), (5, "total", "v")])  # keys first: a projection puts the result in that order
    assert ok and pc.StepCount() == 2
    for bad, text in (([(6, "avg", "v"), (0, "k", None)], "Unexpected agregate function"), ([(5, "t", None), (0, "k", None)], "argument"),
                      ([(0, "k", None)], "without aggregates"), ([], "without options"), ([(5, "k", "v"), (0, "k", None)], "twice")):
        ok, err = pc.InitGroupBy(bad)
        assert not ok and text in err, (bad, err)

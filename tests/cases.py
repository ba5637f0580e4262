"""Programs and checks shared by the CPU tests (oracle vs the reference's golden vectors) and the GPU tests
(product vs the same vectors, product vs oracle).

A `run(resolver, program, batch)` callable returns {column name: column} where a column has `.type`,
`.values`, `.valid` and `.to_pylist()` -- both ydb_amd.arrow_c.ResultColumn and oracle.ssa_oracle.OracleColumn do.
"""
from __future__ import annotations

import math
from typing import Callable, Dict, List, Optional, Sequence

import numpy as np

from ydb_amd.arrow_c import Col
from ydb_amd.ssa_pb import AGG, F, YQL, Aggregate, Column, Program

NP = {"bool": np.bool_, "int8": np.int8, "int16": np.int16, "int32": np.int32, "int64": np.int64, "uint8": np.uint8,
      "uint16": np.uint16, "uint32": np.uint32, "uint64": np.uint64, "float": np.float32, "double": np.float64,
      "timestamp": np.int64, "duration": np.int64}


def col(type_name: str, values: Sequence, nulls_where: Optional[Callable] = None) -> Col:
    arr = np.array(values, dtype=NP[type_name])
    valid = None
    if nulls_where is not None:
        valid = np.array([not nulls_where(v) for v in values], dtype=np.bool_)
    return Col(arr, valid=valid, type=type_name)


def json_columns(spec: Dict[str, dict]) -> Dict[str, Col]:
    return {name: col(c["type"], c["values"]) for name, c in spec.items()}


def pylist(c) -> List:
    return c.to_pylist()


# ---- golden cases --------------------------------------------------------------------------------------------
def check_add_less_filter(case, run):
    res = [(1, "x"), (2, "y"), (3, "z")]
    p = (Program().assign_func(Column(name="res1"), F.MATH_ADD, [1, 2]).assign_func(Column(name="res2"), F.CMP_LESS, ["res1", 3])
         .filter("res2").project(["res1", "res2"]))
    out = run(res, p, json_columns(case["columns"]))
    assert len(out) == case["expect"]["ncols"]
    assert len(out["res1"].valid) == case["expect"]["nrows"]
    assert pylist(out["res1"]) == [20, 68] and out["res1"].type == "int64"
    assert pylist(out["res2"]) == [True, True] and out["res2"].type == "bool"


def check_const_add_filter(case, run):
    res = [(1, "x"), (2, "filter")]
    p = (Program().assign_const(Column(name="y"), "Int32", 56).assign_func(Column(name="res"), F.MATH_ADD, [1, "y"])
         .filter(2).project([2, "res"]))
    out = run(res, p, json_columns(case["columns"]))
    assert len(out) == case["expect"]["ncols"]
    assert len(out["res"].valid) == case["expect"]["nrows"]
    assert pylist(out["res"]) == [120, 99]


def check_min_max_nokeys(case, run):
    res = [(1, "x"), (2, "y")]
    p = Program().group_by([Aggregate(Column(name="min_x"), AGG.MIN, [1]), Aggregate(Column(name="max_y"), AGG.MAX, [2])])
    out = run(res, p, json_columns(case["columns"]))
    e = case["expect"]
    assert [c.type for c in out.values()] == e["types"]
    assert [pylist(c) for c in out.values()] == e["values"]


def check_sum_nokeys(case, run):
    res = [(1, "x"), (2, "y")]
    p = Program().group_by([Aggregate(Column(name="sum_x"), AGG.SUM, [1]), Aggregate(Column(name="sum_y"), AGG.SUM, [2])])
    out = run(res, p, json_columns(case["columns"]))
    e = case["expect"]
    assert [c.type for c in out.values()] == e["types"]
    assert [pylist(c) for c in out.values()] == e["values"]


def _xy_batch(data, nullable, null_x=True):
    is_zero = (lambda v: v == 0) if nullable else None
    return {"x": col("int16", data["x"], is_zero if null_x else None), "y": col("uint32", data["y"], is_zero)}


def check_group_by_sum(case, run):
    res = [(1, "x"), (2, "y")]
    e = case["expect"]
    for v in case["variants"]:
        keys = [1] if v["keys"] == 1 else [1, 2]
        p = Program().group_by([Aggregate(Column(name="agg_x"), AGG.SUM, [1]), Aggregate(Column(name="agg_y"), AGG.SUM, [2])], keys=keys)
        data = case["data"][v["data"]]
        batch = _xy_batch(data, v["nullable"], null_x=(v["data"] != "one_value"))
        out = run(res, p, batch)
        names = ["agg_x", "agg_y", "x"] + (["y"] if v["keys"] == 2 else [])
        assert list(out.keys()) == names, (v, list(out.keys()))
        assert [out[n].type for n in names] == e["types"][:len(names)], v
        nrows = len(out["agg_x"].valid)
        if v["data"] == "empty":
            assert nrows == 0, v
            continue
        if v["data"] == "one_value":
            assert nrows == 1 and pylist(out["agg_x"]) == [1], v
            assert pylist(out["agg_y"]) == [None if v["nullable"] else 0], v
            continue
        assert nrows == e["default"]["nrows"], v
        xs, ax, ay = pylist(out["x"]), pylist(out["agg_x"]), pylist(out["agg_y"])
        for x, a, b in zip(xs, ax, ay):
            want = e["default"]["by_key_x"]["null" if x is None else str(x)]
            assert [a, b] == want, (v, x, a, b)
        assert sorted(str(x) for x in xs) == (["-1", "None"] if v["nullable"] else ["-1", "0"]), (v, xs)


def check_group_by_min_max_some(case, run):
    res = [(1, "x"), (2, "y")]
    fns = {"min": AGG.MIN, "max": AGG.MAX, "some": AGG.SOME}
    e = case["expect"]
    for fn in case["functions"]:
        for nullable in (False, True):
            p = Program().group_by([Aggregate(Column(name="agg_x"), fns[fn], [1]), Aggregate(Column(name="agg_y"), fns[fn], [2])], keys=[1])
            out = run(res, p, _xy_batch(case["data"], nullable, null_x=False))
            assert list(out.keys()) == ["agg_x", "agg_y", "x"]
            assert [c.type for c in out.values()] == e["types"], (fn, nullable)
            assert pylist(out["x"]) == [e["x"]] and pylist(out["agg_x"]) == [e["agg_x"]]
            assert pylist(out["agg_y"]) == [None if nullable else 0], (fn, nullable)


def check_string_match(case, run):
    e = case["expect"]
    ops = {"starts_with": (F.STR_STARTS_WITH, F.STR_STARTS_WITH_IGNORE_CASE), "ends_with": (F.STR_ENDS_WITH, F.STR_ENDS_WITH_IGNORE_CASE),
           "match_substring": (F.STR_MATCH, F.STR_MATCH_IGNORE_CASE)}
    for name, (plain, icase) in ops.items():
        for strings, pattern, fid in ((case["strings"], case["pattern"], plain), (case["strings_icase"], case["pattern_icase"], icase)):
            for typ in ("string", "binary"):
                vals = [s if typ == "string" else s.encode() for s in strings]
                p = (Program().assign_const(Column(name="pat"), "Bytes", pattern.encode())
                     .assign_func(Column(name="res"), fid, [1, "pat"]).project(["res"]))
                out = run([(1, "x")], p, {"x": Col(vals, type=typ)})
                assert out["res"].type == "bool"
                assert [int(v) for v in pylist(out["res"])] == e[name], (name, typ, pattern)


def check_yql_equals_int16_float(case, run):
    p = Program().assign_yql(Column(id=15), YQL.Equals, [1, 2]).project([15])
    out = run([(1, "lhs"), (2, "rhs")], p, json_columns(case["columns"]))
    assert out["15"].type == case["expect"]["type"]
    assert pylist(out["15"]) == case["expect"]["values"]


def check_yql_add_generated_name(case, run):
    # the Assign has no target column: its name is the decimal id of the default (0) column id
    p = Program().assign_yql(None, YQL.Add, ["sum", "vat"]).project(["0"])
    out = run([(1, "sum"), (2, "vat")], p, json_columns(case["columns"]))
    e = case["expect"]
    assert list(out.keys()) == [e["name"]]
    assert out["0"].type == e["type"] and pylist(out["0"]) == e["values"]


def check_str_length(case, run):
    p = Program().assign_func(None, F.STR_LENGTH, ["uid"]).project(["0"])
    out = run([(1, "uid")], p, {"uid": Col(case["strings"], type="string")})
    assert list(out.keys()) == [case["expect"]["name"]]
    assert pylist(out["0"]) == case["expect"]["values"]
    assert out["0"].type == "int32"


def check_yql_string_kernels(case, run):
    ops = {"StringContains": YQL.StringContains, "StartsWith": YQL.StartsWith, "EndsWith": YQL.EndsWith}
    for c in case["cases"]:
        p = (Program().assign_const(Column(id=10), "Bytes", c["pattern"].encode()).assign_yql(Column(id=11), ops[c["op"]], [1, 10])
             .project([11]))
        out = run([(1, "string")], p, {"string": Col(c["strings"], type="string")})
        assert out["11"].type == case["expect"]["type"]
        assert pylist(out["11"]) == c["expect"], c


def aggregator_block(n: int) -> Dict[str, Col]:
    strings = [b"abc", b"def", b"abcd", b"defg", b"ac"]
    i = np.arange(n)
    return {"i16": col("int16", i % 9), "ui32": col("uint32", i % 7),
            "s1": Col([strings[k % 5] for k in range(n)], type="binary"), "s2": Col([strings[k % 3] for k in range(n)], type="binary")}


def check_aggregator_group_counts(case, run):
    res = [(1, "i16"), (2, "ui32"), (3, "s1"), (4, "s2")]
    batch = aggregator_block(case["rows"])
    for label, keys in (("keys_0_1", [1, 2]), ("keys_0_2", [1, 3]), ("keys_2_3", [3, 4]), ("keys_0_1_2_3", [1, 2, 3, 4]),
                        ("keys_0_1", [2, 1]), ("keys_0_2", [3, 1])):
        p = Program().group_by([Aggregate(Column(name="cnt"), AGG.COUNT, [])], keys=keys)
        out = run(res, p, batch)
        assert len(out["cnt"].valid) == case["expect"][label], (label, keys)
        assert int(np.sum(out["cnt"].values)) == case["rows"]


def check_count_with_nulls(case, run):
    ids = np.arange(1, 11, dtype=np.int32)
    batch = {"id": Col(ids, type="int32"), "level": Col(ids.copy(), valid=ids <= 5, type="int32")}
    res = [(1, "id"), (2, "level")]
    e = case["expect"]

    def filtered(lo_hi, keyed):
        p = Program()
        if lo_hi is not None:
            lo, hi = lo_hi
            p.assign_const(100, "Int32", lo).assign_const(101, "Int32", hi)
            p.assign_func(102, F.CMP_GREATER_EQUAL, [1, 100]).assign_func(103, F.CMP_LESS_EQUAL, [1, 101]).filter(102).filter(103)
        p.group_by([Aggregate(110, AGG.COUNT, [2])], keys=[1] if keyed else [])
        return run(res, p, batch)

    assert pylist(filtered((6, 10), False)["110"]) == [e["count_level_where_id_gt_5"]]
    assert pylist(filtered(None, False)["110"]) == [e["count_level"]]
    for label, rng in (("count_level_group_by_id_in_4_5", (4, 5)), ("count_level_group_by_id_in_6_7", (6, 7)),
                       ("count_level_group_by_id_in_5_6", (5, 6))):
        out = filtered(rng, True)
        got = sorted([int(k), int(c)] for k, c in zip(pylist(out["id"]), pylist(out["110"])))
        assert got == e[label], (label, got)
        assert out["110"].type == "uint64"


def check_arith_int32(case, run):
    e = case["expect"]
    for name, fid in (("add", F.MATH_ADD), ("subtract", F.MATH_SUBTRACT), ("multiply", F.MATH_MULTIPLY), ("divide", F.MATH_DIVIDE)):
        p = Program().assign_func(Column(name="res1"), fid, [1, 2]).project(["res1"])
        out = run([(1, "x"), (2, "y")], p, json_columns(case["columns"]))
        assert out["res1"].type == e["type"] and pylist(out["res1"]) == e[name], name


def check_compare_int32(case, run):
    e = case["expect"]
    for name, fid in (("equal", F.CMP_EQUAL), ("less", F.CMP_LESS), ("greater", F.CMP_GREATER), ("greater_equal", F.CMP_GREATER_EQUAL),
                      ("less_equal", F.CMP_LESS_EQUAL), ("not_equal", F.CMP_NOT_EQUAL)):
        p = Program().assign_func(Column(name="res1"), fid, [1, 2]).project(["res1"])
        out = run([(1, "x"), (2, "y")], p, json_columns(case["columns"]))
        assert out["res1"].type == "bool" and [int(v) for v in pylist(out["res1"])] == e[name], name


def check_logic_bool(case, run):
    e = case["expect"]
    for name, fid in (("and", F.BINARY_AND), ("or", F.BINARY_OR), ("xor", F.BINARY_XOR)):
        p = Program().assign_func(Column(name="res1"), fid, [1, 2]).project(["res1"])
        out = run([(1, "x"), (2, "y")], p, json_columns(case["columns"]))
        assert [int(v) for v in pylist(out["res1"])] == e[name], name
    p = Program().assign_func(Column(name="res1"), F.BINARY_NOT, [1]).project(["res1"])
    out = run([(1, "x"), (2, "y")], p, json_columns(case["columns"]))
    assert [int(v) for v in pylist(out["res1"])] == e["invert_x"]


def check_read_aggregate_strings(case, run):
    n = case["rows"]
    batch = {"num": col("int32", list(range(n))), "str": Col([str(i) for i in range(n)], type="string"),
             "bin": Col([str(i).encode() for i in range(n)], type="binary")}
    res = [(1, "num"), (2, "str"), (3, "bin")]
    for cid, typ in ((2, "string"), (3, "binary"), (1, "int32")):
        for label in ("all", "filtered_on_column_equal_1"):
            p = Program()
            if label != "all":
                p.assign_const(50, "Int32", 1).assign_func(51, F.CMP_EQUAL, [1, 50]).filter(51)
            p.group_by([Aggregate(100, AGG.MIN, [cid]), Aggregate(101, AGG.MAX, [cid]), Aggregate(102, AGG.SOME, [cid]),
                        Aggregate(103, AGG.COUNT, [cid])]).project([100, 101, 102, 103])
            out = run(res, p, batch)
            e = case["expect"][label]
            assert list(out.keys()) == ["100", "101", "102", "103"]
            assert [out[k].type for k in ("100", "101", "102")] == [typ] * 3 and out["103"].type == "uint64"

            def text(v):
                return v.decode() if isinstance(v, bytes) else str(v)
            got = [text(pylist(out[k])[0]) for k in ("100", "101", "102")] + [pylist(out["103"])[0]]
            assert got == [str(e["min"]), str(e["max"]), str(e["some"]), e["count"]], (typ, label, got)


GOLDEN_CHECKS = {
    "add_less_filter": check_add_less_filter, "const_add_filter": check_const_add_filter, "min_max_nokeys": check_min_max_nokeys,
    "sum_nokeys": check_sum_nokeys, "group_by_sum": check_group_by_sum, "group_by_min_max_some": check_group_by_min_max_some,
    "string_match": check_string_match, "yql_equals_int16_float": check_yql_equals_int16_float,
    "yql_add_generated_name": check_yql_add_generated_name, "str_length": check_str_length,
    "yql_string_kernels": check_yql_string_kernels, "aggregator_group_counts": check_aggregator_group_counts,
    "count_with_nulls": check_count_with_nulls, "arith_int32": check_arith_int32, "compare_int32": check_compare_int32,
    "logic_bool": check_logic_bool, "read_aggregate_strings": check_read_aggregate_strings,
}


# ---- comparing two result batches -----------------------------------------------------------------------------
FLOAT_SUM_RTOL = 1e-12  # relative tolerance of floating-point sums (summation order differs between GPU and CPU)


def _sort_key(v):
    if v is None:
        return (0, 0)
    if isinstance(v, float) and math.isnan(v):
        return (2, 0)
    return (1, v)


def compare_batches(got, exp, key_names: Sequence[str] = (), float_sum_names: Sequence[str] = (), rtol: float = FLOAT_SUM_RTOL,
                    signed_zero_names: Sequence[str] = ()):
    """Same column names in the same order, same types, same rows (after sorting by the key columns when given).
    Everything is compared exactly except the columns named in `float_sum_names`; in the columns named in
    `signed_zero_names` a zero must also have the expected sign."""
    assert list(got.keys()) == list(exp.keys()), (list(got.keys()), list(exp.keys()))
    for name in exp:
        assert got[name].type == exp[name].type, (name, got[name].type, exp[name].type)
        assert len(got[name].valid) == len(exp[name].valid), (name, len(got[name].valid), len(exp[name].valid))
    n = len(next(iter(exp.values())).valid) if exp else 0
    gl = {k: pylist(v) for k, v in got.items()}
    el = {k: pylist(v) for k, v in exp.items()}
    order_g = list(range(n))
    order_e = list(range(n))
    if key_names:
        order_g.sort(key=lambda r: tuple(_sort_key(gl[k][r]) for k in key_names))
        order_e.sort(key=lambda r: tuple(_sort_key(el[k][r]) for k in key_names))
    for name in exp:
        for rg, re_ in zip(order_g, order_e):
            a, b = gl[name][rg], el[name][re_]
            if a is None or b is None:
                assert a is None and b is None, (name, rg, a, b)
            elif name in float_sum_names:
                both_nan = isinstance(a, float) and isinstance(b, float) and math.isnan(a) and math.isnan(b)
                assert both_nan or abs(a - b) <= rtol * max(abs(a), abs(b)) or a == b, (name, a, b)
            elif isinstance(b, float) and math.isnan(b):
                assert isinstance(a, float) and math.isnan(a), (name, a, b)
            else:
                assert a == b, (name, rg, a, b)
                if name in signed_zero_names and isinstance(b, float) and b == 0.0:
                    assert math.copysign(1.0, a) == math.copysign(1.0, b), (name, rg, a, b)

"""The oracle's keyed aggregation against results worked out by hand from the reference's code: the order in which the
lookup-table path of a single one-byte key visits rows (IAggregateFunction.h:384-459) and the first-value rule of
MIN / MAX / SOME states (AggregateFunctionMinMaxAny.h:57-138)."""
import math

import numpy as np

from oracle import ssa_oracle
from ydb_amd.ssa_pb import AGG, Aggregate, Program

NAN = float("nan")


def run(keys, key_type, f, fn):
    p = Program().group_by([Aggregate(10, fn, [2])], keys=[1])
    out = ssa_oracle.Oracle([(1, "k"), (2, "f")], p).apply({"k": np.array(keys, dtype=key_type), "f": np.array(f, dtype=np.float64)})
    return dict(zip(out["k"].to_pylist(), out["10"].to_pylist()))


def test_wide_key_first_value_rule():
    # row order: a leading NaN stays, a later one never wins; the first of two zeros stays
    got = run([1, 1, 2, 2, 3, 3, 4, 4], np.int32, [NAN, 1.0, 1.0, NAN, 0.0, -0.0, -0.0, 0.0], AGG.MIN)
    assert math.isnan(got[1]) and got[2] == 1.0
    assert math.copysign(1, got[3]) == 1.0 and math.copysign(1, got[4]) == -1.0


def test_single_byte_key_some_is_the_first_row_of_the_lowest_stream():
    # 9 rows: rows 0..7 go to streams 0..3 (row % 4), row 8 is the tail.  Key 7 first appears in row 1 (stream 1) but
    # also in row 4 (stream 0): merging the streams in order gives row 4's value.  Key 9 lives in the tail only.
    keys = [5, 7, 7, 7, 7, 5, 5, 5, 9]
    vals = [10.0, 11.0, 12.0, 13.0, 14.0, 15.0, 16.0, 17.0, 18.0]
    got = run(keys, np.int8, vals, AGG.SOME)
    assert got == {5: 10.0, 7: 14.0, 9: 18.0}
    # the same rows under a wider key: plain row order
    assert run(keys, np.int16, vals, AGG.SOME) == {5: 10.0, 7: 11.0, 9: 18.0}


def test_single_byte_key_min_loses_values_behind_a_streams_leading_nan():
    # key 1: stream 0 = rows 0, 4 -> min(3, 2) = 2; stream 1 = rows 1, 5 -> NaN first, the 1.0 behind it is never taken
    got = run([1, 1, 0, 0, 1, 1, 0, 0], np.uint8, [3.0, NAN, 9.0, 9.0, 2.0, 1.0, 9.0, 9.0], AGG.MIN)
    assert got[1] == 2.0
    # under a wider key the NaN is ignored and 1.0 wins
    assert run([1, 1, 0, 0, 1, 1, 0, 0], np.int32, [3.0, NAN, 9.0, 9.0, 2.0, 1.0, 9.0, 9.0], AGG.MIN)[1] == 1.0


def test_unsafe_float_to_integer_casts_follow_the_x86_conversions():
    # ydb.cast runs with CastOptions(false): static_cast<T>(double) as compiled for x86-64 (cvttsd2si and truncation)
    from ydb_amd.ssa_pb import F
    vals = np.array([NAN, float("inf"), -float("inf"), 1e20, -1e20, -1.5, 3e9, 5e9, 1e19, 70000.5, 300.7, -129.9])
    want = {
        F.CAST_TO_INT64: [-2**63, -2**63, -2**63, -2**63, -2**63, -1, 3000000000, 5000000000, -2**63, 70000, 300, -129],
        F.CAST_TO_INT32: [-2**31, -2**31, -2**31, -2**31, -2**31, -1, -2**31, -2**31, -2**31, 70000, 300, -129],
        F.CAST_TO_INT16: [0, 0, 0, 0, 0, -1, 0, 0, 0, 4464, 300, -129],
        F.CAST_TO_INT8: [0, 0, 0, 0, 0, -1, 0, 0, 0, 112, 44, 127],
        F.CAST_TO_UINT8: [0, 0, 0, 0, 0, 255, 0, 0, 0, 112, 44, 127],
        F.CAST_TO_UINT16: [0, 0, 0, 0, 0, 65535, 0, 0, 0, 4464, 300, 65407],
        F.CAST_TO_UINT32: [0, 0, 0, 0, 0, 4294967295, 3000000000, 705032704, 0, 70000, 300, 4294967167],
        F.CAST_TO_UINT64: [2**63, 2**63, 2**63, 2**63, 2**63, 2**64 - 1, 3000000000, 5000000000, 10**19, 70000, 300, 2**64 - 129],
    }
    for fid, expect in want.items():
        p = Program().assign_func(100, fid, [1]).project([100])
        out = ssa_oracle.Oracle([(1, "f")], p).apply({"f": vals})
        assert out["100"].to_pylist() == expect, (fid, out["100"].to_pylist())


def test_ignore_case_patterns_that_need_unicode_folding_are_declined_by_the_oracle():
    """RE2 (the reference's ignore_case matcher) folds Unicode; the restatement folds ASCII letters only and says so for
    the patterns where that differs: bytes >= 0x80 and the letters k / s (U+212A, U+017F)."""
    import pytest
    from oracle import ssa_oracle
    from ydb_amd.arrow_c import Col
    from ydb_amd.ssa_pb import Column, F, Program
    values = ["Kelvin", "mass", "École", "abc"]
    for pattern, ok in (("ab", True), ("k", False), ("S", False), ("é", False), ("C", True)):
        prog = (Program().assign_const(Column(name="pat"), "Bytes", pattern.encode())
                .assign_func(Column(name="res"), F.STR_MATCH_IGNORE_CASE, [1, "pat"]).project(["res"]))
        orc = ssa_oracle.Oracle([(1, "x")], prog)
        if ok:
            res = orc.apply({"x": Col(values, type="string")})["res"].to_pylist()
            assert [bool(v) for v in res] == [pattern.lower() in v.lower() for v in values]
        else:
            with pytest.raises(ssa_oracle.OracleError, match="Unsupported"):
                orc.apply({"x": Col(values, type="string")})

"""GPU executor (through the C-ABI) vs the CPU oracle on the same seeded inputs: bit-exact for integers, keys,
counts, min/max and element-wise results; floating-point SUMs within FLOAT_SUM_RTOL (cases.py)."""
import numpy as np
import pytest

from cases import compare_batches
from ydb_amd.arrow_c import Col
from ydb_amd.ssa_pb import AGG, F, YQL, Aggregate, Column, Program

pytestmark = pytest.mark.gpu


def both(resolver, program, batch):
    from oracle import ssa_oracle
    from ydb_amd.program import ProgramContainer
    pc = ProgramContainer()
    ok, err = pc.Init(resolver, program)
    assert ok, err
    got = pc.ApplyProgram(batch)
    exp = ssa_oracle.Oracle(resolver, program).apply(batch)
    return got, exp


def both_fail(resolver, program, batch):
    from oracle import ssa_oracle
    from ydb_amd.program import ProgramContainer, SsaError
    pc = ProgramContainer()
    ok, err = pc.Init(resolver, program)
    assert ok, err
    with pytest.raises(SsaError) as gi:
        pc.ApplyProgram(batch)
    with pytest.raises(ssa_oracle.OracleError) as oi:
        ssa_oracle.Oracle(resolver, program).apply(batch)
    return str(gi.value), str(oi.value)


def both_chunked(resolver, ops, batch, chunk=8):
    """ops: (kind, function id, argument ids) with kind 'f' (SIMPLE_ARROW) or 'y' (YQL kernel); one computed output
    column each, at most `chunk` per program (a scan step has a bounded number of output registers)."""
    for lo in range(0, len(ops), chunk):
        p = Program()
        outs = []
        for i, (kind, fid, args) in enumerate(ops[lo:lo + chunk]):
            if kind == "f":
                p.assign_func(100 + i, fid, args)
            else:
                p.assign_yql(100 + i, fid, args)
            outs.append(100 + i)
        p.project(outs)
        got, exp = both(resolver, p, batch)
        compare_batches(got, exp)


SIZES = [0, 1, 63, 64, 65, 1023, 1024, 1025, 4097, 100_003]


def lineitem_like(rng, n, nulls=0.0):
    def maybe(a):
        if nulls <= 0:
            return a
        return Col(a, valid=rng.random(n) >= nulls)
    return {
        "ship": maybe(rng.integers(8036, 10562, n).astype(np.int32)),
        "disc": maybe(rng.integers(0, 11, n) / 100.0),
        "qty": maybe(rng.integers(1, 51, n).astype(np.float64)),
        "price": maybe(rng.integers(90000, 200000, n) / 100.0),
        "flag": maybe(rng.integers(0, 3, n).astype(np.int8)),
        "status": maybe(rng.integers(0, 2, n).astype(np.uint8)),
    }


LI = [(1, "ship"), (2, "disc"), (3, "qty"), (4, "price"), (5, "flag"), (6, "status")]


def q6_like():
    return (Program()
            .assign_const(100, "Int32", 9131).assign_const(101, "Int32", 9496)
            .assign_const(102, "Double", 0.07 - 0.01).assign_const(103, "Double", 0.07 + 0.01).assign_const(104, "Double", 25.0)
            .assign_func(110, F.CMP_GREATER_EQUAL, [1, 100]).assign_func(111, F.CMP_LESS, [1, 101])
            .assign_func(112, F.CMP_GREATER_EQUAL, [2, 102]).assign_func(113, F.CMP_LESS_EQUAL, [2, 103])
            .assign_func(114, F.CMP_LESS, [3, 104])
            .filter(110).filter(111).filter(112).filter(113).filter(114)
            .assign_func(120, F.MATH_MULTIPLY, [4, 2])
            .group_by([Aggregate(130, AGG.SUM, [120]), Aggregate(131, AGG.COUNT, []), Aggregate(132, AGG.MIN, [1]),
                       Aggregate(133, AGG.MAX, [4]), Aggregate(134, AGG.SOME, [3])])
            .project([130, 131, 132, 133, 134]))


@pytest.mark.parametrize("n", SIZES)
@pytest.mark.parametrize("nulls", [0.0, 0.1])
def test_filter_sum_nokeys(n, nulls):
    rng = np.random.default_rng(1000 + n)
    got, exp = both(LI, q6_like(), lineitem_like(rng, n, nulls))
    compare_batches(got, exp, float_sum_names=["130"])


def q1_like(keys):
    return (Program()
            .assign_const(100, "Int32", 10461).assign_const(101, "Double", 1.0)
            .assign_func(110, F.CMP_LESS_EQUAL, [1, 100]).filter(110)
            .assign_func(120, F.MATH_SUBTRACT, [101, 2]).assign_func(121, F.MATH_MULTIPLY, [4, 120])
            .assign_func(122, F.MATH_ADD, [101, 2]).assign_func(123, F.MATH_MULTIPLY, [121, 122])
            .group_by([Aggregate(130, AGG.SUM, [3]), Aggregate(131, AGG.SUM, [4]), Aggregate(132, AGG.SUM, [121]),
                       Aggregate(133, AGG.SUM, [123]), Aggregate(134, AGG.COUNT, [3]), Aggregate(135, AGG.COUNT, []),
                       Aggregate(136, AGG.MIN, [2]), Aggregate(137, AGG.MAX, [1]), Aggregate(138, AGG.SOME, [4])], keys=keys))


@pytest.mark.parametrize("n", SIZES)
@pytest.mark.parametrize("nulls", [0.0, 0.1])
def test_filter_group_by_small_keys(n, nulls):
    rng = np.random.default_rng(2000 + n)
    got, exp = both(LI, q1_like([5, 6]), lineitem_like(rng, n, nulls))
    compare_batches(got, exp, key_names=["flag", "status"], float_sum_names=["130", "131", "132", "133"])


@pytest.mark.parametrize("n", [0, 1, 1000, 50_000])
def test_group_by_single_byte_key_ignores_argument_nulls(n):
    # one 1-byte non-null key: the reference's lookup-table path does not look at argument validity
    rng = np.random.default_rng(2500 + n)
    b = lineitem_like(rng, n, 0.2)
    b["flag"] = rng.integers(0, 3, n).astype(np.int8)
    p = Program().group_by([Aggregate(130, AGG.SUM, [1]), Aggregate(131, AGG.COUNT, [1]), Aggregate(132, AGG.MIN, [1])], keys=[5])
    got, exp = both(LI, p, b)
    compare_batches(got, exp, key_names=["flag"])


@pytest.mark.parametrize("n", [0, 5, 70_000])
def test_group_by_string_keys(n):
    rng = np.random.default_rng(3000 + n)
    flags = [("A", "N", "R")[i] for i in rng.integers(0, 3, n)]
    status = [("F", "O", "")[i] for i in rng.integers(0, 3, n)]
    v = rng.integers(-1000, 1000, n).astype(np.int32)
    p = Program().group_by([Aggregate(10, AGG.SUM, [3]), Aggregate(11, AGG.COUNT, []), Aggregate(12, AGG.MAX, [3])], keys=[1, 2])
    got, exp = both([(1, "rf"), (2, "ls"), (3, "v")], p, {"rf": flags, "ls": status, "v": v})
    compare_batches(got, exp, key_names=["rf", "ls"])


def test_group_by_long_and_null_strings():
    rng = np.random.default_rng(3100)
    n = 20_000
    words = ["", "a", "ab", "x" * 7, "x" * 8, "x" * 9, "y" * 31, "prefix-" + "z" * 100]
    s = [words[i] for i in rng.integers(0, len(words), n)]
    sv = rng.random(n) > 0.1
    k = rng.integers(0, 4, n).astype(np.int64)
    p = Program().group_by([Aggregate(10, AGG.COUNT, [2]), Aggregate(11, AGG.SUM, [2])], keys=[1, 2])
    got, exp = both([(1, "s"), (2, "k")], p, {"s": Col(s, valid=sv), "k": k})
    # a null string key groups with whatever bytes its slot holds (here: the string itself), validity is dropped
    compare_batches(got, exp, key_names=["s", "k"])


@pytest.mark.parametrize("n", [0, 1, 3000, 200_000])
def test_group_by_high_cardinality(n):
    rng = np.random.default_rng(4000 + n)
    key = rng.integers(-40_000, 40_000, n).astype(np.int64)
    val = rng.integers(-1000, 1000, n).astype(np.int32)
    valid = rng.random(n) > 0.3
    f = rng.normal(size=n)
    p = Program().group_by([Aggregate(10, AGG.SUM, [2]), Aggregate(11, AGG.MIN, [2]), Aggregate(12, AGG.MAX, [2]),
                            Aggregate(13, AGG.COUNT, [2]), Aggregate(14, AGG.COUNT, []), Aggregate(15, AGG.MIN, [3]),
                            Aggregate(16, AGG.MAX, [3]), Aggregate(17, AGG.SUM, [3]), Aggregate(18, AGG.SOME, [2])], keys=[1])
    got, exp = both([(1, "k"), (2, "v"), (3, "f")], p, {"k": key, "v": Col(val, valid=valid), "f": f})
    # groups here hold a handful of rows: 1e-12 is relative to the group's sum, which can cancel to ~0
    compare_batches(got, exp, key_names=["k"], float_sum_names=["17"], rtol=1e-9)


def special_floats(rng, n, dtype=np.float64):
    """Mostly small numbers with NaN, +-inf and both zeros mixed in."""
    v = rng.integers(-3, 4, n).astype(dtype)
    pick = rng.random(n)
    v[pick < 0.10] = np.nan
    v[(pick >= 0.10) & (pick < 0.14)] = np.inf
    v[(pick >= 0.14) & (pick < 0.18)] = -np.inf
    v[(pick >= 0.18) & (pick < 0.30)] = -0.0
    v[(pick >= 0.30) & (pick < 0.42)] = 0.0
    return v


@pytest.mark.parametrize("n", [1, 7, 64, 1025, 60_000])
@pytest.mark.parametrize("ngroups", [3, 700, 40_000])
def test_keyed_min_max_keep_a_leading_nan_and_the_first_zero(n, ngroups):
    # AggregateFunctionMinMaxAny.h:96-138: the state takes the group's first non-null value, then `<` / `>` decide
    rng = np.random.default_rng(4100 + n + ngroups)
    key = rng.integers(0, ngroups, n).astype(np.int32)
    f = special_floats(rng, n)
    g = Col(special_floats(rng, n, np.float32), valid=rng.random(n) > 0.3, type="float")
    p = (Program().assign_func(100, F.MATH_ADD, [2, 2])
         .group_by([Aggregate(10, AGG.MIN, [2]), Aggregate(11, AGG.MAX, [2]), Aggregate(12, AGG.MIN, [3]), Aggregate(13, AGG.MAX, [3]),
                    Aggregate(14, AGG.MIN, [100]), Aggregate(15, AGG.COUNT, [3])], keys=[1]))
    got, exp = both([(1, "k"), (2, "f"), (3, "g")], p, {"k": key, "f": f, "g": g})
    compare_batches(got, exp, key_names=["k"], signed_zero_names=["10", "11", "12", "13", "14"])


def test_keyed_min_max_nan_rule_by_hand():
    nan = float("nan")
    key = np.array([1, 1, 1, 2, 2, 2, 3, 3, 4, 4, 5, 5], dtype=np.int64)
    f = np.array([nan, 1.0, 2.0, 1.0, nan, 2.0, nan, nan, 0.0, -0.0, -0.0, 0.0])
    p = Program().group_by([Aggregate(10, AGG.MIN, [2]), Aggregate(11, AGG.MAX, [2])], keys=[1])
    from ydb_amd.program import ProgramContainer
    pc = ProgramContainer()
    assert pc.Init([(1, "k"), (2, "f")], p)[0]
    out = pc.ApplyProgram({"k": key, "f": f})
    rows = {int(k): (mn, mx) for k, mn, mx in zip(out["k"].to_pylist(), out["10"].to_pylist(), out["11"].to_pylist())}
    import math
    assert math.isnan(rows[1][0]) and math.isnan(rows[1][1])        # NaN came first: it stays
    assert rows[2] == (1.0, 2.0)                                    # a later NaN never wins
    assert math.isnan(rows[3][0]) and math.isnan(rows[3][1])
    assert math.copysign(1, rows[4][0]) == 1.0 and math.copysign(1, rows[4][1]) == 1.0    # the first zero stays
    assert math.copysign(1, rows[5][0]) == -1.0 and math.copysign(1, rows[5][1]) == -1.0


@pytest.mark.parametrize("n", [1, 3, 4, 5, 9, 1000, 70_003])
@pytest.mark.parametrize("filtered", [False, True])
def test_single_byte_key_visits_rows_in_four_streams(n, filtered):
    # IAggregateFunction.h:384-459: SOME is the first row of the lowest stream, a zero MIN / MAX the first zero in that order
    rng = np.random.default_rng(4200 + n)
    key = rng.integers(0, 5, n).astype(np.uint8)
    v = rng.integers(0, 1 << 30, n).astype(np.int64)
    f = rng.integers(-2, 3, n).astype(np.float64)
    f[rng.random(n) < 0.3] = -0.0
    keep = (rng.random(n) < 0.6).astype(np.uint8)
    p = Program()
    if filtered:
        p.filter(4)
    p.group_by([Aggregate(10, AGG.SOME, [2]), Aggregate(11, AGG.MIN, [3]), Aggregate(12, AGG.MAX, [3]), Aggregate(13, AGG.COUNT, []),
                Aggregate(14, AGG.SUM, [2])], keys=[1])
    got, exp = both([(1, "k"), (2, "v"), (3, "f"), (4, "keep")], p, {"k": key, "v": v, "f": f, "keep": keep})
    compare_batches(got, exp, key_names=["k"], signed_zero_names=["11", "12"])


def test_single_byte_key_with_a_nan_under_min_is_declined():
    from oracle import ssa_oracle
    from ydb_amd import _lib
    from ydb_amd.program import ProgramContainer, SsaError
    key = np.array([1, 1, 2, 2, 1, 1, 2, 2], dtype=np.int8)
    f = np.array([1.0, float("nan"), 3.0, 4.0, 0.5, 6.0, 7.0, 8.0])
    p = Program().group_by([Aggregate(10, AGG.MIN, [2])], keys=[1])
    pc = ProgramContainer()
    assert pc.Init([(1, "k"), (2, "f")], p)[0]
    with pytest.raises(SsaError) as e:
        pc.ApplyProgram({"k": key, "f": f})
    assert e.value.code == _lib.SSA_UNSUPPORTED
    # the oracle follows the reference: the NaN leads stream 1 of key 1, whose later 6.0 is lost; 0.5 is in stream 0
    exp = ssa_oracle.Oracle([(1, "k"), (2, "f")], p).apply({"k": key, "f": f})
    assert sorted(zip(exp["k"].to_pylist(), exp["10"].to_pylist())) == [(1, 0.5), (2, 3.0)]
    # without the NaN the same program runs
    f[1] = 9.0
    got, exp = both([(1, "k"), (2, "f")], p, {"k": key, "f": f})
    compare_batches(got, exp, key_names=["k"])


@pytest.mark.parametrize("jit", ["0", "1"])
def test_colliding_key_hashes_are_told_apart_by_the_key_bytes(jit, monkeypatch):
    # rows whose keys are hashed (strings of more than 8 bytes, more than 119 key bits) with the hash cut to three bits:
    # hundreds of different keys per hash value, every one its own group
    monkeypatch.setenv("SSA_JIT", jit)
    monkeypatch.setenv("SSA_DEBUG_FLAGS", "64")
    rng = np.random.default_rng(4300)
    n = 40_000
    words = ["", "short", "12345678"] + ["long-key-%05d-%s" % (i, "x" * (i % 23)) for i in range(600)]
    s = [words[i] for i in rng.integers(0, len(words), n)]
    v = rng.integers(-1000, 1000, n).astype(np.int32)
    p = Program().group_by([Aggregate(10, AGG.SUM, [2]), Aggregate(11, AGG.COUNT, []), Aggregate(12, AGG.MIN, [2])], keys=[1])
    got, exp = both([(1, "s"), (2, "v")], p, {"s": s, "v": v})
    assert len(exp["s"].to_pylist()) == len(set(s))
    compare_batches(got, exp, key_names=["s"])
    # wide fixed-width keys (two nullable int64 and an int32: 162 key bits)
    a = Col(rng.integers(0, 30, n).astype(np.int64), valid=rng.random(n) > 0.1)
    b = Col(rng.integers(0, 30, n).astype(np.int64), valid=rng.random(n) > 0.1)
    c = rng.integers(0, 3, n).astype(np.int32)
    p = Program().group_by([Aggregate(10, AGG.SUM, [4]), Aggregate(11, AGG.COUNT, [])], keys=[1, 2, 3])
    got, exp = both([(1, "a"), (2, "b"), (3, "c"), (4, "v")], p, {"a": a, "b": b, "c": c, "v": v})
    compare_batches(got, exp, key_names=["a", "b", "c"])


@pytest.mark.parametrize("widths", [("int32", "int16"), ("int64", "int64"), ("uint8", "float"), ("int64", "uint64", "int32")])
def test_group_by_nullable_and_wide_keys(widths):
    rng = np.random.default_rng(5000 + len(widths) + sum(map(len, widths)))
    n = 30_000
    res, batch = [], {}
    np_types = {"int32": np.int32, "int16": np.int16, "int64": np.int64, "uint8": np.uint8, "uint64": np.uint64, "float": np.float32}
    for i, w in enumerate(widths):
        vals = rng.integers(0, 4, n).astype(np_types[w])
        batch[f"k{i}"] = Col(vals, valid=rng.random(n) > 0.2, type=w)
        res.append((i + 1, f"k{i}"))
    batch["v"] = rng.integers(0, 100, n).astype(np.uint32)
    res.append((10, "v"))
    keys = [i + 1 for i in range(len(widths))]
    p = Program().group_by([Aggregate(20, AGG.SUM, [10]), Aggregate(21, AGG.COUNT, []), Aggregate(22, AGG.MAX, [10])], keys=keys)
    got, exp = both(res, p, batch)
    compare_batches(got, exp, key_names=[f"k{i}" for i in range(len(widths))])


@pytest.mark.parametrize("n", [0, 1, 1025, 70_001])
def test_projection_with_filter_and_strings(n):
    rng = np.random.default_rng(6000 + n)
    strs = [("ab" if i % 3 == 0 else "xyz") + str(i % 7) for i in range(n)]
    sv = rng.random(n) > 0.15
    a = rng.integers(-100, 100, n).astype(np.int16)
    b = rng.integers(0, 1 << 40, n).astype(np.uint64)
    bv = rng.random(n) > 0.15
    flt = rng.random(n) > 0.4
    p = (Program().assign_const(100, "Bytes", b"ab").assign_func(110, F.STR_STARTS_WITH, [1, 100])
         .assign_func(111, F.STR_LENGTH, [1]).assign_func(112, F.MATH_ADD, [2, 3]).assign_func(113, F.CMP_GREATER, [2, 111])
         .assign_func(114, F.IS_NULL, [3]).filter(110).filter(4)
         .project([1, 111, 112, 113, 114, 2, 3, 4, 100]))
    res = [(1, "s"), (2, "a"), (3, "b"), (4, "flt")]
    got, exp = both(res, p, {"s": Col(strs, valid=sv), "a": a, "b": Col(b, valid=bv), "flt": flt})
    compare_batches(got, exp)


def test_sliced_inputs_and_bit_offsets():
    rng = np.random.default_rng(7000)
    total, off = 5000, 37
    a = rng.integers(-1000, 1000, total).astype(np.int32)
    av = rng.random(total) > 0.2
    flag = rng.random(total) > 0.5
    s = ["s%d" % (i % 11) for i in range(total)]
    p = (Program().assign_const(100, "Int32", 0).assign_func(110, F.CMP_GREATER, [1, 100]).filter(110).filter(2)
         .group_by([Aggregate(120, AGG.SUM, [1]), Aggregate(121, AGG.COUNT, [])], keys=[3]))
    batch = {"a": Col(a, valid=av, offset=off), "flag": Col(flag, offset=off), "s": Col(s, offset=off)}
    got, exp = both([(1, "a"), (2, "flag"), (3, "s")], p, batch)
    compare_batches(got, exp, key_names=["s"])


def test_dictionary_column_equals_decoded_column():
    rng = np.random.default_rng(7100)
    n = 10_000
    d = ["google", "yandex", "", "bing.com/search"]
    codes = rng.integers(0, len(d), n).astype(np.int8)
    v = rng.integers(0, 10, n).astype(np.int64)
    p = (Program().assign_const(100, "Bytes", b"g").assign_yql(110, YQL.StringContains, [1, 100]).filter(110)
         .group_by([Aggregate(120, AGG.COUNT, []), Aggregate(121, AGG.SUM, [2])], keys=[1]))
    got, exp = both([(1, "phrase"), (2, "v")], p, {"phrase": Col(codes, dictionary=d), "v": v})
    compare_batches(got, exp, key_names=["phrase"])


ARITH_TYPES = ["int8", "int16", "int32", "int64", "uint8", "uint16", "uint32", "uint64", "float", "double"]
NPT = {"int8": np.int8, "int16": np.int16, "int32": np.int32, "int64": np.int64, "uint8": np.uint8, "uint16": np.uint16,
       "uint32": np.uint32, "uint64": np.uint64, "float": np.float32, "double": np.float64}


def small_values(rng, t, n):
    if t in ("float", "double"):
        return (rng.integers(-50, 50, n) / 4.0).astype(NPT[t])
    lo = 0 if t.startswith("u") else -60
    return rng.integers(lo, 60, n).astype(NPT[t])


@pytest.mark.parametrize("ta", ARITH_TYPES)
@pytest.mark.parametrize("tb", ARITH_TYPES)
def test_arrow_binary_ops_all_type_pairs(ta, tb):
    rng = np.random.default_rng(abs(hash((ta, tb))) % (1 << 31))
    n = 700
    a = Col(small_values(rng, ta, n), valid=rng.random(n) > 0.1, type=ta)
    bvals = small_values(rng, tb, n)
    bvals[bvals == 0] = 1  # integer division by zero is an error (checked separately)
    b = Col(bvals, valid=rng.random(n) > 0.1, type=tb)
    p = Program()
    outs = []
    for i, fid in enumerate([F.MATH_ADD, F.MATH_SUBTRACT, F.MATH_MULTIPLY, F.MATH_DIVIDE, F.CMP_EQUAL, F.CMP_NOT_EQUAL, F.CMP_LESS,
                             F.CMP_LESS_EQUAL, F.CMP_GREATER, F.CMP_GREATER_EQUAL]):
        p.assign_func(100 + i, fid, [1, 2])
        outs.append(100 + i)
    p.project(outs)
    got, exp = both([(1, "a"), (2, "b")], p, {"a": a, "b": b})
    compare_batches(got, exp)


@pytest.mark.parametrize("ta", ARITH_TYPES)
@pytest.mark.parametrize("tb", ARITH_TYPES)
def test_yql_binary_kernels_all_type_pairs(ta, tb):
    rng = np.random.default_rng(abs(hash((tb, ta, 1))) % (1 << 31))
    n = 700
    a = Col(small_values(rng, ta, n), valid=rng.random(n) > 0.1, type=ta)
    b = Col(small_values(rng, tb, n), valid=rng.random(n) > 0.1, type=tb)  # zeros included: Div/Mod yield NULL
    p = Program()
    outs = []
    for i, op in enumerate([YQL.Add, YQL.Sub, YQL.Mul, YQL.Div, YQL.Mod, YQL.Equals, YQL.NotEquals, YQL.Less, YQL.LessOrEqual,
                            YQL.Greater, YQL.GreaterOrEqual]):
        p.assign_yql(100 + i, op, [1, 2])
        outs.append(100 + i)
    p.project(outs)
    got, exp = both([(1, "a"), (2, "b")], p, {"a": a, "b": b})
    compare_batches(got, exp)


def test_integer_wraparound_and_extremes():
    a64 = np.array([np.iinfo(np.int64).max, np.iinfo(np.int64).min, -1, 0, 1, 1 << 62], dtype=np.int64)
    b64 = np.array([1, -1, np.iinfo(np.int64).min, 5, -1, 4], dtype=np.int64)
    a8 = np.array([127, -128, -1, 0, 1, 100], dtype=np.int8)
    b8 = np.array([1, -1, -128, 5, -1, 100], dtype=np.int8)
    u = np.array([0, 1, (1 << 64) - 1, 1 << 63, 5, 7], dtype=np.uint64)
    ops = []
    for x, y in ((1, 2), (3, 4), (5, 5)):
        for fid in (F.MATH_ADD, F.MATH_SUBTRACT, F.MATH_MULTIPLY):
            ops.append(("f", fid, [x, y]))
    ops.append(("f", F.MATH_DIVIDE, [1, 2]))   # INT64_MIN / -1 -> 0
    ops.append(("f", F.MATH_DIVIDE, [3, 4]))   # -128 / -1 -> 0
    for op in (YQL.Div, YQL.Mod):
        ops.append(("y", op, [1, 2]))          # MIN / -1 -> NULL
        ops.append(("y", op, [3, 4]))
    both_chunked([(1, "a64"), (2, "b64"), (3, "a8"), (4, "b8"), (5, "u")], ops, {"a64": a64, "b64": b64, "a8": a8, "b8": b8, "u": u})


def test_kleene_logic_and_filters_ignore_validity():
    rng = np.random.default_rng(8000)
    n = 3000
    x = Col(rng.integers(0, 2, n).astype(np.uint8), valid=rng.random(n) > 0.3)
    y = Col(rng.integers(0, 2, n).astype(np.uint8), valid=rng.random(n) > 0.3)
    bx = Col(rng.random(n) > 0.5, valid=rng.random(n) > 0.3)
    by = Col(rng.random(n) > 0.5, valid=rng.random(n) > 0.3)
    v = rng.integers(0, 100, n).astype(np.int32)
    p = (Program().assign_yql(100, YQL.And, [1, 2]).assign_yql(101, YQL.Or, [1, 2]).assign_yql(102, YQL.Xor, [1, 2])
         .assign_func(103, F.BINARY_AND, [3, 4]).assign_func(104, F.BINARY_OR, [3, 4]).assign_func(105, F.BINARY_XOR, [3, 4])
         .assign_func(106, F.BINARY_NOT, [3]).assign_yql(107, YQL.Coalesce, [1, 2])
         .project([100, 101, 102, 103, 104, 105, 106, 107]))
    res = [(1, "x"), (2, "y"), (3, "bx"), (4, "by"), (5, "v")]
    batch = {"x": x, "y": y, "bx": bx, "by": by, "v": v}
    got, exp = both(res, p, batch)
    compare_batches(got, exp)
    # a NULL predicate row is kept or dropped by the value its slot holds
    for pred in (1, 3):
        p2 = Program().filter(pred).group_by([Aggregate(110, AGG.COUNT, [5]), Aggregate(111, AGG.SUM, [5])])
        got, exp = both(res, p2, batch)
        compare_batches(got, exp)


def test_casts():
    rng = np.random.default_rng(8100)
    n = 2000
    cols = {"i": rng.integers(-70000, 70000, n).astype(np.int32), "u": rng.integers(0, 1 << 33, n).astype(np.uint64),
            "f": (rng.normal(size=n) * 300).astype(np.float64), "g": (rng.normal(size=n) * 100).astype(np.float32),
            "s": rng.integers(0, 2_000_000_000, n).astype(np.uint32)}
    res = [(1, "i"), (2, "u"), (3, "f"), (4, "g"), (5, "s")]
    casts = [F.CAST_TO_BOOLEAN, F.CAST_TO_INT8, F.CAST_TO_INT16, F.CAST_TO_INT32, F.CAST_TO_INT64, F.CAST_TO_UINT8, F.CAST_TO_UINT16,
             F.CAST_TO_UINT32, F.CAST_TO_UINT64, F.CAST_TO_FLOAT, F.CAST_TO_DOUBLE]
    ops = []
    for src in (1, 2, 3, 4):
        for c in casts:
            if src in (3, 4) and c not in (F.CAST_TO_FLOAT, F.CAST_TO_DOUBLE, F.CAST_TO_INT64, F.CAST_TO_INT32):
                continue  # out-of-range float -> narrow int is undefined behaviour in the reference's C++ cast
            ops.append(("f", c, [src]))
    ops.append(("f", F.CAST_TO_TIMESTAMP, [5]))
    both_chunked(res, ops, cols)


def test_errors_match():
    a = np.array([1, 2, 3], dtype=np.int32)
    z = np.array([1, 0, 3], dtype=np.int32)
    g, o = both_fail([(1, "a"), (2, "z")], Program().assign_func(10, F.MATH_DIVIDE, [1, 2]).project([10]), {"a": a, "z": z})
    assert "divide by zero" in g and "divide by zero" in o
    # a division by zero under a null slot is not evaluated
    got, exp = both([(1, "a"), (2, "z")], Program().assign_func(10, F.MATH_DIVIDE, [1, 2]).project([10]),
                    {"a": a, "z": Col(z, valid=np.array([True, False, True]))})
    compare_batches(got, exp)
    # implicit casts are checked: uint64 beyond int64 compared with a signed column
    big = np.array([1, (1 << 63) + 5], dtype=np.uint64)
    s = np.array([-1, 2], dtype=np.int64)
    g, o = both_fail([(1, "u"), (2, "s")], Program().assign_func(10, F.CMP_LESS, [1, 2]).project([10]), {"u": big, "s": s})
    assert "not in range" in g and "not in range" in o
    # filter over a non-boolean column, projection of a missing column
    both_fail([(1, "a"), (2, "z")], Program().filter(1).project([1]), {"a": a, "z": z})
    both_fail([(1, "a"), (2, "z")], Program().project([1, 7]), {"a": a, "z": z})


def test_build_filter_matches_oracle():
    from oracle import ssa_oracle
    from ydb_amd.program import ProgramContainer
    rng = np.random.default_rng(8200)
    for n in (0, 1, 64, 1000, 4097):
        b = lineitem_like(rng, n, 0.1)
        p = q6_like()
        pc = ProgramContainer()
        ok, err = pc.Init(LI, p)
        assert ok, err
        mask, selected = pc.BuildFilter(b)
        exp = ssa_oracle.Oracle(LI, p).build_filter(b)
        assert selected == int(exp.sum())
        assert (mask == exp).all()


def test_multi_step_program():
    rng = np.random.default_rng(8300)
    n = 20_000
    b = lineitem_like(rng, n)
    p = (Program().assign_const(100, "Int32", 9000).assign_func(110, F.CMP_GREATER, [1, 100]).filter(110)
         .assign_func(120, F.MATH_MULTIPLY, [4, 2]).project([120, 5, 6, 3])
         .assign_const(200, "Double", 20.0).assign_func(210, F.CMP_LESS, [3, 200]).filter(210)
         .group_by([Aggregate(220, AGG.SUM, [120]), Aggregate(221, AGG.COUNT, [])], keys=[5, 6])
         .project([220, 221, 6, 5]))
    got, exp = both(LI, p, b)
    compare_batches(got, exp, key_names=["flag", "status"], float_sum_names=["220"])


def _decimal_column(rng, n, nulls=0.0):
    """(n, 2) uint64 [lo, hi] words of signed 128-bit values around +-2^70 (carries out of the low word are common)."""
    vals = [int(rng.integers(-(1 << 62), 1 << 62)) * int(rng.integers(1, 1 << 9)) + int(rng.integers(0, 1 << 40)) for _ in range(n)]
    words = np.array([[v & ((1 << 64) - 1), (v >> 64) & ((1 << 64) - 1)] for v in vals], dtype=np.uint64).reshape(n, 2)
    valid = (rng.random(n) >= nulls) if nulls > 0 else None
    return vals, Col(words, valid=valid, decimal=(38, 4))


@pytest.mark.parametrize("n", [0, 1, 1000, 30_000])
@pytest.mark.parametrize("nulls", [0.0, 0.2])
def test_decimal128_sum(n, nulls):
    """Extension without a counterpart in the reference (parity unpinned): checked against the oracle's definition and
    against Python integers."""
    rng = np.random.default_rng(9000 + n)
    vals, dec = _decimal_column(rng, n, nulls)
    key = rng.integers(0, 5, n).astype(np.int32)
    flt = rng.random(n) > 0.3
    res = [(1, "d"), (2, "k"), (3, "f")]
    batch = {"d": dec, "k": key, "f": flt}
    p = Program().filter(3).group_by([Aggregate(10, AGG.SUM, [1]), Aggregate(11, AGG.COUNT, [1])])
    got, exp = both(res, p, batch)
    compare_batches(got, exp)
    assert got["10"].type == "decimal128"
    ok = flt & (dec.valid if dec.valid is not None else True)
    want = sum(v for v, o in zip(vals, ok) if o)
    if ok.any():
        w = (want + (1 << 127)) % (1 << 128) - (1 << 127)
        assert got["10"].to_pylist() == [w]
    else:
        assert got["10"].to_pylist() == [None]
    p = Program().filter(3).group_by([Aggregate(10, AGG.SUM, [1]), Aggregate(11, AGG.COUNT, []), Aggregate(12, AGG.MAX, [2])], keys=[2])
    got, exp = both(res, p, batch)
    compare_batches(got, exp, key_names=["k"])
    for k, s in zip(got["k"].to_pylist(), got["10"].to_pylist()):
        sel = ok & (key == k)
        want = sum(v for v, o in zip(vals, sel) if o) if sel.any() else None
        assert s == want, (k, s, want)


def test_general_like_patterns():
    """match_like with patterns outside the three plain forms (wildcards in the middle, '_', escapes, UTF-8 text):
    GPU against the oracle and against the regular expression the reference builds for the pattern; as a computed
    column, as a filter, and with nulls."""
    import like_cases
    from ydb_amd.ssa_pb import F
    rng = np.random.default_rng(17)
    values = like_cases.strings(rng, 5000) + ["abc", "a\nc", "ab%c", "ab_cd", "\\", "a", "日本", "x日本y", "aébc"]
    valid = rng.random(len(values)) > 0.15
    for pattern in like_cases.PATTERNS:
        prog = (Program().assign_const(Column(name="pat"), "Bytes", pattern.encode())
                .assign_func(Column(name="res"), F.STR_MATCH_LIKE, [1, "pat"]).project(["res"]))
        got, exp = both([(1, "x")], prog, {"x": Col(values, type="string")})
        compare_batches(got, exp)
        assert [bool(v) for v in got["res"].to_pylist()] == like_cases.expected(values, pattern), pattern
        got, exp = both([(1, "x")], prog, {"x": Col(values, type="string", valid=valid)})
        compare_batches(got, exp)
        flt = (Program().assign_const(Column(name="pat"), "Bytes", pattern.encode())
               .assign_func(Column(name="res"), F.STR_MATCH_LIKE, [1, "pat"]).filter("res")
               .group_by([Aggregate(Column(name="n"), AGG.COUNT, [])]).project(["n"]))
        got, exp = both([(1, "x")], flt, {"x": Col(values, type="string")})
        compare_batches(got, exp)
        assert int(got["n"].to_pylist()[0]) == sum(like_cases.expected(values, pattern)), pattern


@pytest.mark.parametrize("n", [1024, 4096 + 17, 30_000])
def test_substring_search_over_whole_tiles(n):
    """match_substring over full tiles takes the wave-wide search (16 bytes per lane, hits mapped back to rows): a small
    alphabet so that matches are frequent, straddle row boundaries and 16-byte pieces, empty and one-byte strings, patterns
    of 1..16 bytes and longer (per-row path), with and without ASCII case folding, on utf8 and binary columns."""
    from ydb_amd.ssa_pb import F
    rng = np.random.default_rng(n)
    letters = np.array(list("abAB"))
    lens = rng.choice([0, 0, 1, 2, 3, 5, 8, 15, 16, 17, 31, 40], n)
    values = ["".join(rng.choice(letters, k)) for k in lens]
    valid = rng.random(n) > 0.1
    patterns = ["a", "ab", "aba", "abab", "bbbbbbb", "abababab", "aabbaabba", "ababababababab", "aaaaaaaaaaaaaaaa", "abababababababababab",
                "B", "Ab", "ABAB"]
    for pattern in patterns:
        for fid in (F.STR_MATCH, F.STR_MATCH_IGNORE_CASE):
            for typ in ("string", "binary"):
                col = [v if typ == "string" else v.encode() for v in values]
                prog = (Program().assign_const(Column(name="pat"), "Bytes", pattern.encode())
                        .assign_func(Column(name="res"), fid, [1, "pat"]).project(["res"]))
                got, exp = both([(1, "x")], prog, {"x": Col(col, type=typ, valid=valid)})
                compare_batches(got, exp)
                want = [(pattern.lower() in v.lower()) if fid == F.STR_MATCH_IGNORE_CASE else (pattern in v) for v in values]
                res = got["res"].to_pylist()
                assert [bool(r) for r, ok in zip(res, valid) if ok] == [w for w, ok in zip(want, valid) if ok], (pattern, fid, typ)
    # as a filter feeding a count (the ClickBench Q20 shape, YQL kernel)
    prog = (Program(version=5).assign_const(106, "Bytes", b"abba").assign_yql(107, YQL.StringContains, [1, 106]).filter(107)
            .group_by([Aggregate(108, AGG.COUNT, [])]).project([108]))
    got, exp = both([(1, "x")], prog, {"x": Col(values, type="string")})
    compare_batches(got, exp)
    assert int(got["108"].to_pylist()[0]) == sum("abba" in v for v in values)


def test_folded_constant_arrays_stay_arrays():
    """is_null of a column without nulls is known to be false for every row; whatever is computed from it alone is
    still an array of the batch's length -- a filter takes it (and drops every row), a projection materialises it."""
    n = 1000
    batch = {"a": np.arange(n, dtype=np.int32), "b": np.arange(n, dtype=np.float64)}
    res = [(1, "a"), (2, "b")]
    p = (Program().assign_func(100, F.IS_NULL, [1]).assign_func(101, F.BINARY_AND, [100, 100]).assign_func(102, F.BINARY_NOT, [101])
         .filter(101).project([2, 100]))
    got, exp = both(res, p, batch)
    compare_batches(got, exp)
    assert len(got["b"].valid) == 0
    p = (Program().assign_func(100, F.IS_NULL, [1]).assign_func(101, F.BINARY_AND, [100, 100]).assign_func(102, F.BINARY_NOT, [101])
         .filter(102).project([2, 101, 102]))
    got, exp = both(res, p, batch)
    compare_batches(got, exp)
    assert len(got["b"].valid) == n


@pytest.mark.parametrize("jit", ["0", "1"])
def test_unsafe_float_to_integer_casts(jit, monkeypatch):
    # static_cast<T>(double) where C++ leaves it open: both sides follow the x86-64 conversions (known answers for the
    # oracle in test_oracle_by_hand.py)
    monkeypatch.setenv("SSA_JIT", jit)
    nan, inf = float("nan"), float("inf")
    base = [nan, inf, -inf, 1e20, -1e20, -1.5, 3e9, 5e9, 1e19, 1.9e19, 70000.5, 300.7, -129.9, 2147483647.5, -2147483648.5, 0.0, -0.0]
    rng = np.random.default_rng(77)
    d = np.array(base + list(rng.normal(0, 1e10, 3000)) + list(rng.normal(0, 1e20, 1000)))
    f = d.astype(np.float32)
    casts = [F.CAST_TO_INT8, F.CAST_TO_INT16, F.CAST_TO_INT32, F.CAST_TO_INT64, F.CAST_TO_UINT8, F.CAST_TO_UINT16, F.CAST_TO_UINT32,
             F.CAST_TO_UINT64, F.CAST_TO_BOOLEAN]
    ops = [("f", c, [src]) for c in casts for src in (1, 2)]
    both_chunked([(1, "d"), (2, "f")], ops, {"d": d, "f": f})


@pytest.mark.parametrize("length", [1, 2, 4, 8, 3])
@pytest.mark.parametrize("typ", ["string", "binary"])
def test_string_keys_of_one_length_take_the_fixed_width_path(length, typ):
    # 70 000 rows and up: string keys whose probe (first, second, last offset) says "one length" are scanned as
    # fixed-width values while every offset is checked; the result is the serialized-key result all the same
    rng = np.random.default_rng(4400 + length)
    n = 150_000
    alphabet = [("%c" % (65 + i)) * length for i in range(5)]
    s = [alphabet[i] for i in rng.integers(0, len(alphabet), n)]
    if typ == "binary":
        s = [x.encode() for x in s]
    k = Col(rng.integers(0, 3, n).astype(np.int16), valid=rng.random(n) > 0.2)   # a nullable key next to it: raw slot values
    v = Col(rng.integers(-100, 100, n).astype(np.int32), valid=rng.random(n) > 0.1)
    f = rng.random(n)
    keep = rng.random(n) < 0.7
    p = (Program().filter(5)
         .group_by([Aggregate(10, AGG.SUM, [3]), Aggregate(11, AGG.COUNT, []), Aggregate(12, AGG.MIN, [4]), Aggregate(13, AGG.SOME, [3]),
                    Aggregate(14, AGG.COUNT, [3])], keys=[1, 2]))
    res = [(1, "s"), (2, "k"), (3, "v"), (4, "f"), (5, "keep")]
    got, exp = both(res, p, {"s": Col(s, type=typ), "k": k, "v": v, "f": f, "keep": keep})
    compare_batches(got, exp, key_names=["s", "k"])
    # one string key alone (a single one-byte string is NOT the lookup-table path of a one-byte integer key)
    p = Program().group_by([Aggregate(10, AGG.SOME, [3]), Aggregate(11, AGG.COUNT, [3]), Aggregate(12, AGG.MAX, [4])], keys=[1])
    got, exp = both(res, p, {"s": Col(s, type=typ), "k": k, "v": v, "f": f, "keep": keep})
    compare_batches(got, exp, key_names=["s"])


@pytest.mark.parametrize("where", [1, 77_777, 149_998])
def test_string_keys_that_only_look_like_one_length(where):
    # same first, second and last offsets as 150 000 one-byte strings, but two neighbours are "" and "AB": the scan
    # finds the odd offset and the step is run again the general way
    rng = np.random.default_rng(4500 + where)
    n = 150_000
    s = [("A", "N", "R")[i] for i in rng.integers(0, 3, n)]
    s[where], s[where + 1] = "", "AB"
    v = rng.integers(-100, 100, n).astype(np.int64)
    p = Program().group_by([Aggregate(10, AGG.SUM, [2]), Aggregate(11, AGG.COUNT, [])], keys=[1])
    for _ in range(2):  # the second run of the same handle does not try again
        got, exp = both([(1, "s"), (2, "v")], p, {"s": s, "v": v})
        compare_batches(got, exp, key_names=["s"])
    assert sorted(x for x in exp["s"].to_pylist()) == sorted([b"", b"A", b"AB", b"N", b"R"]) or \
        sorted(exp["s"].to_pylist()) == sorted(["", "A", "AB", "N", "R"])


@pytest.mark.parametrize("n", [0, 1, 5, 1000, 80_000])
@pytest.mark.parametrize("jit", ["0", "1"])
def test_string_min_max_some(n, jit, monkeypatch):
    # SingleValueDataString (AggregateFunctionMinMaxAny.h:160-341): the first value, then bytewise `<` / `>`
    monkeypatch.setenv("SSA_JIT", jit)
    rng = np.random.default_rng(4600 + n)
    words = ["", "a", "ab", "abc", "b", "zz", "Zz", "\xff\x01", "the quick brown fox", "the quick brown fo", "the quick brown fox!", "日本語", "日本"]
    s = Col([words[i] for i in rng.integers(0, len(words), n)], valid=rng.random(n) > 0.25, type="string")
    b = Col([words[i].encode() for i in rng.integers(0, len(words), n)], type="binary")
    k32 = rng.integers(0, 40, n).astype(np.int32)
    k8 = rng.integers(0, 4, n).astype(np.int8)
    v = rng.integers(-5, 5, n).astype(np.int64)
    keep = rng.random(n) < 0.8
    res = [(1, "s"), (2, "b"), (3, "k32"), (4, "k8"), (5, "v"), (6, "keep")]
    batch = {"s": s, "b": b, "k32": k32, "k8": k8, "v": v, "keep": keep}
    aggs = [Aggregate(10, AGG.MIN, [1]), Aggregate(11, AGG.MAX, [1]), Aggregate(12, AGG.SOME, [1]), Aggregate(13, AGG.MIN, [2]),
            Aggregate(14, AGG.MAX, [2]), Aggregate(15, AGG.SOME, [2]), Aggregate(16, AGG.COUNT, [1]), Aggregate(17, AGG.SUM, [5])]
    for keys, names in (([3], ["k32"]), ([4], ["k8"]), ([3, 4], ["k32", "k8"]), ([2], ["b"]), ([], [])):
        for filtered in (False, True):
            p = Program()
            if filtered:
                p.filter(6)
            p.group_by(aggs, keys=keys)
            got, exp = both(res, p, batch)
            compare_batches(got, exp, key_names=names)


def test_ignore_case_patterns_that_need_unicode_folding_are_declined():
    """The reference matches ignore_case patterns with RE2, which folds Unicode ('k' also matches U+212A, 's' U+017F, 'é'
    matches 'É'); the device folds ASCII letters only, so such patterns are refused (the caller's CPU path takes them),
    while patterns of other ASCII characters still match, also in data that holds non-ASCII text."""
    from ydb_amd.ssa_pb import F
    values = ["Kelvin K", "mass", "MASS", "lonſ", "École", "école", "abc", "ABC déjà"]
    for fid in (F.STR_MATCH_IGNORE_CASE, F.STR_STARTS_WITH_IGNORE_CASE, F.STR_ENDS_WITH_IGNORE_CASE):
        for pattern in ("k", "S", "mass", "é", "déj"):
            prog = (Program().assign_const(Column(name="pat"), "Bytes", pattern.encode())
                    .assign_func(Column(name="res"), fid, [1, "pat"]).project(["res"]))
            g, o = both_fail([(1, "x")], prog, {"x": Col(values, type="string")})
            assert "UNSUPPORTED" in g and "Unsupported" in o, (pattern, g, o)
        prog = (Program().assign_const(Column(name="pat"), "Bytes", b"AB")
                .assign_func(Column(name="res"), fid, [1, "pat"]).project(["res"]))
        got, exp = both([(1, "x")], prog, {"x": Col(values, type="string")})
        compare_batches(got, exp)

"""Operations a step gets only through TAssign objects built in code: known-answer vectors of the reference's own tests
and seeded inputs for GPU-versus-oracle parity.

Vectors: ydb/core/formats/arrow/ut/ut_program_step.cpp:311-400 (Round0, Round1, Gcd, Lcm, Mod, ModOrZero, Abs, Negate),
ydb/library/arrow_kernels/ut_arithmetic.cpp:69-353 and ut_round.cpp:8-34.  `run(resolver, program, batch)` returns the
result columns by name.
"""
import math

import numpy as np

from ydb_amd.arrow_c import Col
from ydb_amd.ssa_pb import OP, Column, Program

INT_TYPES = ["int8", "int16", "int32", "int64", "uint8", "uint16", "uint32", "uint64"]
SIGNED = ["int8", "int16", "int32", "int64"]
FLOATS = ["float32", "float64"]

# (operation, argument types, arguments, expected) -- expected values are in the argument type unless noted
BINARY_VECTORS = [
    (OP.Gcd, INT_TYPES, ([32, 12], [24, 10]), [8, 2]),                 # ut_arithmetic.cpp:178-186
    (OP.Gcd, ["int32"], ([64, 16, 8], [32, 12, 4]), [32, 4, 4]),       # ut_program_step.cpp:362-367
    (OP.Lcm, INT_TYPES, ([6, 12], [3, 10]), [6, 60]),                  # ut_arithmetic.cpp:239-247
    (OP.Lcm, ["int32"], ([64, 16, 8], [32, 12, 4]), [64, 48, 8]),      # ut_program_step.cpp:369-374
    (OP.Modulo, INT_TYPES, ([10, 16], [3, 5]), [1, 1]),                # ut_arithmetic.cpp:305-313
    (OP.Modulo, FLOATS, ([10.2234, 16.2347], [3.22343, 5.4234]), [1, 1]),  # :315-323
    (OP.Modulo, ["int32"], ([64, 16, 8], [3, 5, 2]), [1, 1, 0]),       # ut_program_step.cpp:376-381
    (OP.ModuloOrZero, INT_TYPES, ([10, 16], [6, 0]), [4, 0]),          # ut_arithmetic.cpp:336-344
    (OP.ModuloOrZero, FLOATS, ([10.2234, 16.2347], [0.23, 5.4234]), [0, 1]),  # :346-353
    (OP.ModuloOrZero, ["int32"], ([64, 16, 8], [3, 5, 0]), [1, 1, 0]),  # ut_program_step.cpp:383-388
]
UNARY_VECTORS = [
    (OP.Abs, SIGNED, [-32, -12, 54], [32, 12, 54]),                    # ut_arithmetic.cpp:70-77
    (OP.Abs, ["uint8", "uint16", "uint32", "uint64"], [32, 12, 54], [32, 12, 54]),
    (OP.Abs, FLOATS, [-32.4, -12.3, 54.7], [32.4, 12.3, 54.7]),
    (OP.Negate, SIGNED, [-32, -12, 54], [32, 12, -54]),                # ut_arithmetic.cpp:131-138
    (OP.Negate, FLOATS, [-32.4, -12.3, 54.7], [32.4, 12.3, -54.7]),
    (OP.Round, FLOATS, [2.34, 5.65, 10.01, 100.0], [2, 6, 10, 100]),   # ut_round.cpp:8-15
    (OP.RoundBankers, FLOATS, [2.34, 5.5, 6.5, 100.7], [2, 6, 6, 101]),  # ut_round.cpp:17-24
    (OP.RoundToExp2, INT_TYPES + FLOATS, [2.34, 5.5, 6.5, 100.7, 54], [2, 4, 4, 64, 32]),  # ut_round.cpp:26-33
    (OP.Round, ["float64"], [32.3, 12.5, 34.7], [32.0, 13.0, 35.0]),   # ut_program_step.cpp:311-317 with libm's values
    (OP.RoundBankers, ["float64"], [32.3, 12.5, 34.7], [32.0, 12.0, 35.0]),
    (OP.RoundToExp2, ["float64"], [32.3, 12.5, 34.7], [32.0, 8.0, 32.0]),
    (OP.Ceil, ["float64"], [32.3, 12.5, 34.7], [33.0, 13.0, 35.0]),    # ut_program_step.cpp:319-325
    (OP.Floor, ["float64"], [32.3, 12.5, 34.7], [32.0, 12.0, 34.0]),
    (OP.Trunc, ["float64"], [32.3, 12.5, 34.7], [32.0, 12.0, 34.0]),
]
ERROR_VECTORS = [
    (OP.Lcm, INT_TYPES, ([0, 16], [0, 5])),                            # ut_arithmetic.cpp:294-301
    (OP.Modulo, FLOATS, ([10.2234, 16.2347], [0, 5.4234])),            # ut_arithmetic.cpp:325-332
]


def type_name(t):
    return {"float32": "float", "float64": "double"}.get(t, t)


def _array(dtype, values):
    return np.array(values).astype(dtype)   # NumVecToArray: static_cast of doubles to the column's C type


def unary_program(op):
    return Program().assign_op(Column(name="res"), op, [1]).project(["res"])


def binary_program(op):
    return Program().assign_op(Column(name="res"), op, [1, 2]).project(["res"])


def check_vectors(run):
    for op, types, (x, y), want in BINARY_VECTORS:
        for t in types:
            out = run([(1, "x"), (2, "y")], binary_program(op), {"x": _array(t, x), "y": _array(t, y)})
            assert out["res"].type == type_name(t), (op, t, out["res"].type)
            assert out["res"].to_pylist() == _array(t, want).tolist(), (op, t)
    for op, types, x, want in UNARY_VECTORS:
        for t in types:
            out = run([(1, "x")], unary_program(op), {"x": _array(t, x)})
            assert out["res"].type == type_name(t), (op, t, out["res"].type)
            assert out["res"].to_pylist() == _array(t, want).tolist(), (op, t)
    # the filter programs of ut_program_step.cpp:47-73: res1 = op(x), res2 = (res1 == z), filter res2 -> 3 rows
    for op, types, x, want in UNARY_VECTORS:
        t = types[-1]
        prog = (Program().assign_op(Column(name="res1"), op, [1]).assign_func(Column(name="res2"), 1, ["res1", 2])
                .filter("res2").project(["res1", "res2"]))
        out = run([(1, "x"), (2, "z")], prog, {"x": _array(t, x), "z": _array(t, want)})
        assert len(out["res1"].to_pylist()) == len(x) and all(out["res2"].to_pylist())
    # constants (func_math.h:47-56,131-140) and is_valid
    out = run([(1, "x")], Program().assign_op(Column(name="e"), OP.E).assign_op(Column(name="pi"), OP.Pi).project(["e", "pi", 1]),
              {"x": _array("int32", [1, 2, 3])})
    assert out["e"].to_pylist() == [math.e] * 3 and out["pi"].to_pylist() == [math.pi] * 3
    out = run([(1, "x")], unary_program(OP.IsValid), {"x": Col(_array("int32", [1, 2, 3]), valid=np.array([True, False, True]))})
    assert out["res"].to_pylist() == [True, False, True]
    out = run([(1, "x")], unary_program(OP.Sqrt), {"x": _array("int32", [0, 4, 2, 1 << 20])})
    assert out["res"].type == "double" and out["res"].to_pylist() == [0.0, 2.0, math.sqrt(2.0), 1024.0]


def check_errors(run_fails):
    for op, types, (x, y) in ERROR_VECTORS:
        for t in types:
            msg = run_fails([(1, "x"), (2, "y")], binary_program(op), {"x": _array(t, x), "y": _array(t, y)})
            assert "divide by zero" in msg, (op, t, msg)
    # no kernel: gcd of floats, abs of a string (ut_arithmetic.cpp:213-235)
    msg = run_fails([(1, "x"), (2, "y")], binary_program(OP.Gcd), {"x": _array("float64", [1.0]), "y": _array("float64", [2.0])})
    assert "no kernel" in msg
    msg = run_fails([(1, "x")], unary_program(OP.Negate), {"x": Col(["a"], type="string")})
    assert "no kernel" in msg


def random_column(rng, t, n, nonzero=False, nulls=0.0):
    if t in FLOATS:
        v = np.where(rng.random(n) < 0.5, rng.normal(0, 50, n), rng.normal(0, 1e6, n))
        v = np.where(rng.random(n) < 0.1, np.round(v) + 0.5, v).astype(t)   # ties for the rounding modes
        if nonzero:
            v = np.where(np.abs(v) < 1, v + np.copysign(2, v + 0.1), v).astype(t)
    else:
        info = np.iinfo(t)
        small = rng.integers(max(info.min, -200), min(info.max, 200), n, endpoint=True)
        lo, hi = (info.min + 1, info.max) if info.min < 0 else (0, info.max)
        wide = rng.integers(lo, hi, n, dtype=np.int64 if t != "uint64" else np.uint64, endpoint=True)
        v = np.where(rng.random(n) < 0.6, small.astype(t), wide.astype(t)).astype(t)
        if nonzero:
            v = np.where(v == 0, 3, v).astype(t)
    if nulls > 0:
        return Col(v, valid=rng.random(n) >= nulls)
    return v


# libm functions (ydb/library/arrow_kernels/func_math.h:7-170): operation, python equivalent, argument range
MATH_OPS = [
    (OP.Acosh, math.acosh, (1.0, 1e6)), (OP.Atanh, math.atanh, (-0.999, 0.999)), (OP.Cbrt, lambda v: math.copysign(abs(v) ** (1 / 3), v), (-1e9, 1e9)),
    (OP.Cosh, math.cosh, (-20.0, 20.0)), (OP.Erf, math.erf, (-4.0, 4.0)), (OP.Erfc, math.erfc, (-4.0, 4.0)),
    (OP.Exp, math.exp, (-50.0, 50.0)), (OP.Exp2, lambda v: 2.0 ** v, (-60.0, 60.0)), (OP.Lgamma, math.lgamma, (0.05, 100.0)),
    (OP.Sinh, math.sinh, (-20.0, 20.0)), (OP.Tgamma, math.gamma, (0.05, 30.0)), (OP.Sqrt, math.sqrt, (0.0, 1e12)),
]
# ut_math.cpp:19-52: acosh of {2.324, 1.34234, 41.14324, 123} as float32 (computed in float), float64 and integers
ACOSH_ARGS = [2.324, 1.34234, 41.14324, 123]


def math_argument(rng, t, n, lo, hi):
    v = rng.uniform(lo, hi, n)
    if t not in FLOATS:
        info = np.iinfo(t)
        v = np.clip(np.round(v), max(info.min, math.ceil(lo)), min(info.max, math.floor(hi)))
    return v.astype(t)


def close(a, b, rtol, atol):
    if a is None or b is None:
        return a is None and b is None
    if math.isnan(b) or math.isinf(b):
        return (math.isnan(a) and math.isnan(b)) or a == b
    return abs(a - b) <= atol + rtol * abs(b)


def check_acosh_vectors(run, rtol64=1e-15, rtol32=1e-7):
    for t, rtol in (("float64", rtol64), ("float32", rtol32), ("int64", rtol64), ("uint8", rtol64), ("int16", rtol64)):
        x = _array(t, ACOSH_ARGS)
        out = run([(1, "x")], unary_program(OP.Acosh), {"x": x})
        assert out["res"].type == "double"
        want = [math.acosh(float(v)) for v in x]
        assert all(close(a, b, rtol, 0) for a, b in zip(out["res"].to_pylist(), want)), (t, out["res"].to_pylist(), want)


CHECKED = [OP.AddNotNull, OP.SubtractNotNull, OP.MultiplyNotNull, OP.DivideNotNull]


def checked_expected(op, t, a, b):
    """Python-integer model of add_checked ... divide_checked on one pair: the value, or the error message."""
    if t in FLOATS:
        raise ValueError
    info = np.iinfo(t)
    a, b = int(a), int(b)
    if op == OP.DivideNotNull:
        if b == 0:
            return "divide by zero"
        q = abs(a) // abs(b) * (1 if (a >= 0) == (b >= 0) else -1)
    else:
        q = a + b if op == OP.AddNotNull else a - b if op == OP.SubtractNotNull else a * b
    return q if info.min <= q <= info.max else "overflow"


def check_checked_arithmetic(run, run_fails):
    rng = np.random.default_rng(41)
    for t in INT_TYPES:
        info = np.iinfo(t)
        # pairs around the type's limits, one program run per pair so that each error is attributed
        edge = [info.min, info.min + 1, -1, 0, 1, 2, info.max - 1, info.max]
        edge = [v for v in edge if info.min <= v <= info.max]
        for op in CHECKED:
            for a in edge:
                for b in edge:
                    want = checked_expected(op, t, a, b)
                    batch = {"x": _array(t, [a]), "y": _array(t, [b])}
                    if isinstance(want, str):
                        assert want in run_fails([(1, "x"), (2, "y")], binary_program(op), batch), (op, t, a, b)
                    else:
                        got = run([(1, "x"), (2, "y")], binary_program(op), batch)["res"]
                        assert got.type == type_name(t) and got.to_pylist() == [want], (op, t, a, b)
    # null slots are not visited: an overflowing pair under a null is no error and leaves a null
    x = Col(_array("int8", [100, 100, 3]), valid=np.array([True, False, True]))
    y = _array("int8", [1, 100, 0])
    out = run([(1, "x"), (2, "y")], binary_program(OP.AddNotNull), {"x": x, "y": y})
    assert out["res"].to_pylist() == [101, None, 3]
    assert "divide by zero" in run_fails([(1, "x"), (2, "y")], binary_program(OP.DivideNotNull), {"x": x, "y": y})
    # floats: plain arithmetic, division by zero is an error
    for t in FLOATS:
        a, b = rng.normal(0, 100, 200).astype(t), (rng.normal(0, 100, 200).astype(t) + np.array(1000, dtype=t))
        for op, f in zip(CHECKED, (np.add, np.subtract, np.multiply, np.divide)):
            out = run([(1, "x"), (2, "y")], binary_program(op), {"x": a, "y": b})
            assert out["res"].to_pylist() == f(a, b).tolist(), (op, t)
        assert "divide by zero" in run_fails([(1, "x"), (2, "y")], binary_program(OP.DivideNotNull),
                                             {"x": _array(t, [1.0]), "y": _array(t, [0.0])})


def if_program(pred=1, then=2, otherwise=3):
    """The ternary YQL kernel as KQP sends it: function type YQL_KERNEL, a KernelIdx and three arguments, nothing else
    (ydb/core/kqp/query_compiler/kqp_olap_compiler.cpp:740-759)."""
    return Program(version=5).assign_yql(Column(name="res"), None, [pred, then, otherwise], kernel_idx=7).project(["res"])


def check_block_if(run, run_fails):
    res = [(1, "p"), (2, "a"), (3, "b")]
    p = np.array([1, 0, 1, 0, 0], dtype=np.uint8)
    a = Col(np.array([10, 11, 12, 13, 14], dtype=np.int32), valid=np.array([True, True, False, True, True]))
    b = Col(np.array([20, 21, 22, 23, 24], dtype=np.int32), valid=np.array([True, False, True, True, True]))
    out = run(res, if_program(), {"p": p, "a": a, "b": b})
    assert out["res"].type == "int32" and out["res"].to_pylist() == [10, None, None, 23, 24]
    # scalar branches broadcast
    prog = (Program(version=5).assign_const(Column(name="c1"), "Double", 1.5).assign_const(Column(name="c2"), "Double", -2.5)
            .assign_yql(Column(name="res"), None, [1, "c1", "c2"], kernel_idx=0).project(["res"]))
    out = run(res, prog, {"p": p, "a": a, "b": b})
    assert out["res"].to_pylist() == [1.5, -2.5, 1.5, -2.5, -2.5]
    # computed predicate and branches
    prog = (Program(version=5).assign_const(100, "Int32", 12).assign_yql(101, 13, [2, 100])      # a < 12
            .assign_yql(102, 3, [2, 3]).assign_yql(103, 4, [3, 2])                                   # a + b, b - a
            .assign_const(104, "Uint8", 0).assign_yql(105, 17, [101, 104])                          # coalesce(pred, false)
            .assign_yql(Column(name="res"), None, [105, 102, 103], kernel_idx=1).project(["res"]))
    out = run(res, prog, {"p": p, "a": a, "b": b})
    assert out["res"].to_pylist() == [30, None, None, 10, 10]
    assert "Different return types" in run_fails(res, if_program(1, 2, 1), {"p": p, "a": a, "b": b})
    assert "Expected bool" in run_fails(res, if_program(2, 2, 3), {"p": p, "a": a, "b": b})
    nullable_pred = Col(p, valid=np.array([True, True, False, True, True]))
    assert "Expected bool" in run_fails(res, if_program(), {"p": nullable_pred, "a": a, "b": b})

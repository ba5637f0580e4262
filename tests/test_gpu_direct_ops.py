"""Operations without a protobuf id on the GPU (through the C-ABI): the reference's known-answer vectors, and the CPU
oracle on seeded inputs -- bit-exact for every operation (sqrt included: both sides round correctly)."""
import numpy as np
import pytest

import direct_ops_cases as dc
from cases import compare_batches
from ydb_amd.ssa_pb import AGG, OP, Aggregate, Column, Program

pytestmark = pytest.mark.gpu


def gpu_run(resolver, program, batch):
    from ydb_amd.program import ProgramContainer
    pc = ProgramContainer()
    ok, err = pc.Init(resolver, program)
    assert ok, err
    return pc.ApplyProgram(batch)


def gpu_fails(resolver, program, batch):
    from ydb_amd.program import ProgramContainer, SsaError
    pc = ProgramContainer()
    ok, err = pc.Init(resolver, program)
    assert ok, err
    with pytest.raises(SsaError) as e:
        pc.ApplyProgram(batch)
    return str(e.value)


def both(resolver, program, batch):
    from oracle import ssa_oracle
    return gpu_run(resolver, program, batch), ssa_oracle.Oracle(resolver, program).apply(batch)


def test_reference_vectors():
    dc.check_vectors(gpu_run)


def test_reference_errors():
    dc.check_errors(gpu_fails)


def test_flag_is_required():
    from ydb_amd.program import ProgramContainer
    ok, err = ProgramContainer().Init([(1, "x")], dc.unary_program(OP.Abs).encode(), olap_wrapper=False)
    assert not ok


UNARY = [OP.Abs, OP.Negate, OP.Round, OP.RoundBankers, OP.RoundToExp2, OP.Floor, OP.Ceil, OP.Trunc, OP.Sqrt, OP.IsValid]
BINARY = [OP.Gcd, OP.Lcm, OP.Modulo, OP.ModuloOrZero]


@pytest.mark.parametrize("t", dc.INT_TYPES + dc.FLOATS)
@pytest.mark.parametrize("n,nulls", [(1, 0.0), (1025, 0.0), (70_001, 0.1)])
def test_unary_operations_against_oracle(t, n, nulls):
    rng = np.random.default_rng(100 + n)
    x = dc.random_column(rng, t, n, nulls=nulls)
    if t in dc.INT_TYPES and np.iinfo(t).bits == 64:
        # floor / ceil / trunc cast integers to double safely: values beyond 2^53 are an error on both sides
        ops = [o for o in UNARY if o not in (OP.Floor, OP.Ceil, OP.Trunc)]
    else:
        ops = UNARY
    p = Program()
    for i, op in enumerate(ops):
        p.assign_op(100 + i, op, [1])
    p.project([100 + i for i in range(len(ops))])
    got, exp = both([(1, "x")], p, {"x": x})
    compare_batches(got, exp)


@pytest.mark.parametrize("t", dc.INT_TYPES + dc.FLOATS)
@pytest.mark.parametrize("n,nulls", [(1, 0.0), (1025, 0.0), (70_001, 0.1)])
def test_binary_operations_against_oracle(t, n, nulls):
    rng = np.random.default_rng(200 + n)
    x = dc.random_column(rng, t, n, nulls=nulls)
    y = dc.random_column(rng, t, n, nonzero=True)
    ops = [o for o in BINARY if t in dc.INT_TYPES or o in (OP.Modulo, OP.ModuloOrZero)]
    p = Program()
    for i, op in enumerate(ops):
        p.assign_op(100 + i, op, [1, 2]).assign_op(110 + i, op, [2, 1] if op != OP.Modulo else [1, 2])
    p.project([100 + i for i in range(len(ops))] + [110 + i for i in range(len(ops))])
    got, exp = both([(1, "x"), (2, "y")], p, {"x": x, "y": y})
    compare_batches(got, exp)


def test_mixed_types_take_the_common_numeric_type():
    """ut_arithmetic.cpp:198-211,259-269: operands of different integer types are cast to their common type."""
    rng = np.random.default_rng(7)
    n = 5000
    for t0 in dc.INT_TYPES:
        for t1 in dc.INT_TYPES:
            if {t0, t1} == {"int64", "uint64"} or t0 == t1:
                continue
            x = rng.integers(1, 100, n).astype(t0)
            y = rng.integers(1, 100, n).astype(t1)
            p = (Program().assign_op(100, OP.Gcd, [1, 2]).assign_op(101, OP.Lcm, [1, 2]).assign_op(102, OP.Modulo, [1, 2])
                 .project([100, 101, 102]))
            got, exp = both([(1, "x"), (2, "y")], p, {"x": x, "y": y})
            compare_batches(got, exp)
            assert got["100"].type == exp["100"].type


def test_operations_inside_filters_and_aggregates():
    """The new operations feed filters, group-by keys and aggregates like any other assign; also through a kernel built
    for the program."""
    import os
    rng = np.random.default_rng(9)
    n = 300_000
    batch = {"a": rng.integers(-10**6, 10**6, n).astype(np.int32), "b": rng.integers(1, 50, n).astype(np.int32),
             "f": rng.normal(0, 100, n)}
    res = [(1, "a"), (2, "b"), (3, "f")]
    p = (Program().assign_op(100, OP.ModuloOrZero, [1, 2]).assign_op(101, OP.Abs, [100]).assign_op(102, OP.Round, [3])
         .assign_const(103, "Int32", 3).assign_func(104, 5, [101, 103]).filter(104)
         .group_by([Aggregate(110, AGG.SUM, [102]), Aggregate(111, AGG.COUNT, []), Aggregate(112, AGG.MAX, [100])], keys=[101]))
    for jit in ("0", "1"):
        os.environ["SSA_JIT"] = jit
        try:
            got, exp = both(res, p, batch)
        finally:
            del os.environ["SSA_JIT"]
        compare_batches(got, exp, key_names=["101"], float_sum_names=["110"])


def test_divide_by_zero_is_reported_from_any_row():
    n = 100_000
    x = np.arange(n, dtype=np.int64)
    y = np.ones(n, dtype=np.int64)
    y[n - 7] = 0
    msg = gpu_fails([(1, "x"), (2, "y")], dc.binary_program(OP.Modulo), {"x": x, "y": y})
    assert "divide by zero" in msg
    out = gpu_run([(1, "x"), (2, "y")], dc.binary_program(OP.ModuloOrZero), {"x": x, "y": y})
    assert out["res"].to_pylist() == [0] * n


MATH_RTOL64, MATH_RTOL32 = 1e-13, 1e-6   # the device's math library against the host's: a few units in the last place


@pytest.mark.parametrize("t", ["float64", "float32", "int32", "uint8", "int64"])
def test_math_functions_against_oracle(t):
    """func_math.h kernels: float arguments are computed in float, everything else in double; results within
    MATH_RTOL64 / MATH_RTOL32 of the oracle's (which calls the host C library like the reference)."""
    rng = np.random.default_rng(31)
    n = 20_000
    rtol = MATH_RTOL32 if t == "float32" else MATH_RTOL64
    dc.check_acosh_vectors(gpu_run, rtol64=MATH_RTOL64, rtol32=MATH_RTOL32)
    for op, _, (lo, hi) in dc.MATH_OPS:
        x = dc.math_argument(rng, t, n, lo, hi)
        got, exp = both([(1, "x")], dc.unary_program(op), {"x": x})
        assert got["res"].type == "double"
        g, e = got["res"].to_pylist(), exp["res"].to_pylist()
        atol = 1e-6 if op == OP.Lgamma and t == "float32" else 1e-13 if op == OP.Lgamma else 0.0   # lgamma crosses zero at 1 and 2
        bad = [(v, a, b) for v, a, b in zip(x, g, e) if not dc.close(a, b, rtol, atol)]
        assert not bad, (op, t, bad[:5])
    x, y = dc.math_argument(rng, t, n, -100, 100), dc.math_argument(rng, t, n, -100, 100)
    got, exp = both([(1, "x"), (2, "y")], dc.binary_program(OP.Hypot), {"x": x, "y": y})
    assert all(dc.close(a, b, rtol, 0) for a, b in zip(got["res"].to_pylist(), exp["res"].to_pylist()))
    assert "No function registered" in gpu_fails([(1, "x")], dc.unary_program(OP.Exp10), {"x": x})


def test_math_functions_in_a_built_kernel(monkeypatch):
    monkeypatch.setenv("SSA_JIT", "1")
    rng = np.random.default_rng(33)
    n = 100_000
    batch = {"x": rng.uniform(0.1, 20, n), "k": rng.integers(0, 5, n).astype(np.int32)}
    p = (Program().assign_op(100, OP.Lgamma, [1]).assign_op(101, OP.Erf, [1]).assign_op(102, OP.Hypot, [100, 101])
         .group_by([Aggregate(110, AGG.SUM, [102]), Aggregate(111, AGG.MAX, [100])], keys=[2]))
    got, exp = both([(1, "x"), (2, "k")], p, batch)
    compare_batches(got, exp, key_names=["k"], float_sum_names=["110", "111"], rtol=1e-9)


def test_math_functions_need_a_built_kernel(monkeypatch):
    """The generic interpreter does not carry the math library: with run-time builds switched off such a program is
    refused, it does not run on some other path."""
    from ydb_amd.program import SsaError
    monkeypatch.setenv("SSA_JIT", "0")
    with pytest.raises(SsaError) as e:
        gpu_run([(1, "x")], dc.unary_program(OP.Erf), {"x": np.linspace(-2, 2, 1000)})
    assert "kernel built for the program" in str(e.value)
    out = gpu_run([(1, "x")], dc.unary_program(OP.Sqrt), {"x": np.array([4.0, 9.0])})   # the exact operations do run
    assert out["res"].to_pylist() == [2.0, 3.0]


def test_checked_arithmetic():
    """add_checked ... divide_checked around every integer type's limits (Python integers as the model) and on floats."""
    dc.check_checked_arithmetic(gpu_run, gpu_fails)


@pytest.mark.parametrize("t", ["int16", "uint32", "int64", "float64"])
def test_checked_arithmetic_against_oracle(t):
    rng = np.random.default_rng(43)
    n = 50_000
    if t in dc.FLOATS:
        x, y = rng.normal(0, 100, n), rng.normal(0, 100, n) + 1000
    else:
        hi = 150 if t == "int16" else 60_000 if t == "uint32" else 3_000_000_000
        x = rng.integers(hi // 2, hi, n).astype(t)     # x >= y: an unsigned difference must not wrap
        y = rng.integers(1, hi // 2, n).astype(t)
    x = dc.Col(x, valid=rng.random(n) > 0.1)
    p = Program()
    outs = []
    for i, op in enumerate(dc.CHECKED):
        p.assign_op(100 + i, op, [1, 2])
        outs.append(100 + i)
        if not (t.startswith("uint") and op == OP.SubtractNotNull):
            p.assign_op(110 + i, op, [2, 1])
            outs.append(110 + i)
    p.project(outs)
    got, exp = both([(1, "x"), (2, "y")], p, {"x": x, "y": y})
    compare_batches(got, exp)


def test_block_if():
    dc.check_block_if(gpu_run, gpu_fails)


@pytest.mark.parametrize("t", ["int8", "uint16", "int64", "float32", "float64"])
@pytest.mark.parametrize("jit", ["0", "1"])
def test_block_if_against_oracle(t, jit, monkeypatch):
    monkeypatch.setenv("SSA_JIT", jit)
    rng = np.random.default_rng(51)
    n = 70_001
    batch = {"p": (rng.random(n) < 0.4).astype(np.uint8), "a": dc.random_column(rng, t, n, nulls=0.1),
             "b": dc.random_column(rng, t, n, nulls=0.2), "k": rng.integers(0, 9, n).astype(np.int16)}
    res = [(1, "p"), (2, "a"), (3, "b"), (4, "k")]
    got, exp = both(res, dc.if_program(), batch)
    compare_batches(got, exp)
    # feeding a filter and an aggregate: COUNT / MIN / MAX of the selected side per key, rows where it is null dropped by COUNT
    prog = (Program(version=5).assign_yql(100, None, [1, 2, 3], kernel_idx=0)
            .group_by([Aggregate(110, AGG.COUNT, [100]), Aggregate(111, AGG.MIN, [100]), Aggregate(112, AGG.MAX, [100])], keys=[4]))
    got, exp = both(res, prog, batch)
    compare_batches(got, exp, key_names=["k"])


def test_only_the_math_library_needs_a_built_kernel(monkeypatch):
    """Checked arithmetic, the rounding family and If run on the generic kernels too."""
    monkeypatch.setenv("SSA_JIT", "0")
    x = np.array([1, 2, 3], dtype=np.int32)
    for op in dc.CHECKED:
        out = gpu_run([(1, "x"), (2, "y")], dc.binary_program(op), {"x": x, "y": x})
        assert len(out["res"].to_pylist()) == 3


def test_modulo_of_nan_and_infinities_follows_the_x86_conversion():
    """mod / modOrZero convert floats to int64 first; for NaN and out-of-range values that conversion is what the
    reference's x86-64 build produces (INT64_MIN), not the GPU instruction's saturation."""
    x = np.array([743.0, np.inf, -np.inf, np.nan, 1e30, 7.5])
    y = np.array([np.nan, -np.inf, np.inf, 5.0, 3.0, -1e30])
    got, exp = both([(1, "x"), (2, "y")], dc.binary_program(OP.ModuloOrZero), {"x": x, "y": y})
    compare_batches(got, exp)
    assert got["res"].to_pylist()[:3] == [743.0, 0.0, 0.0]

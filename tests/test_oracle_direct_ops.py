"""Operations without a protobuf id (gcd lcm mod modOrZero abs negate is_valid round* floor ceil trunc sqrt e pi): the CPU
oracle against the known-answer vectors of the reference's unit tests (CPU only)."""
import math

import numpy as np
import pytest

import direct_ops_cases as dc
from oracle import ssa_oracle
from ydb_amd.ssa_pb import OP, Column, Program


def oracle_run(resolver, program, batch):
    return ssa_oracle.Oracle(resolver, program).apply(batch)


def oracle_fails(resolver, program, batch):
    with pytest.raises(ssa_oracle.OracleError) as e:
        ssa_oracle.Oracle(resolver, program).apply(batch)
    return str(e.value)


def test_oracle_matches_the_reference_vectors():
    dc.check_vectors(oracle_run)


def test_oracle_errors_like_the_reference():
    dc.check_errors(oracle_fails)


def test_direct_operations_need_the_flag():
    """A TFunction id above the protobuf's range is 'Unsupported function' unless the program is declared as one
    built from TAssign objects."""
    body = dc.unary_program(OP.Abs).encode()
    with pytest.raises(ssa_oracle.OracleError):
        ssa_oracle.Oracle([(1, "x")], body, olap_wrapper=False)


@pytest.mark.parametrize("t", dc.INT_TYPES)
def test_integer_kernels_against_python_integers(t):
    """gcd / lcm / mod follow C++'s truncating remainder in the column's type; the product of lcm wraps."""
    rng = np.random.default_rng(11)
    n = 2000
    x = dc.random_column(rng, t, n)
    y = dc.random_column(rng, t, n, nonzero=True)
    bits = np.iinfo(t).bits
    signed = np.iinfo(t).min < 0

    def wrap(v):
        v &= (1 << bits) - 1
        return v - (1 << bits) if signed and v >> (bits - 1) else v

    def rem(a, b):  # truncating remainder: the sign follows the dividend
        return (abs(a) % abs(b)) * (1 if a >= 0 else -1)

    def gcd(a, b):
        while b:
            a, b = b, rem(a, b)
        return a

    got = oracle_run([(1, "x"), (2, "y")], dc.binary_program(OP.Modulo), {"x": x, "y": y})["res"].to_pylist()
    assert got == [rem(int(a), int(b)) for a, b in zip(x, y)]
    got = oracle_run([(1, "x"), (2, "y")], dc.binary_program(OP.Gcd), {"x": x, "y": y})["res"].to_pylist()
    assert got == [gcd(int(a), int(b)) for a, b in zip(x, y)]
    got = oracle_run([(1, "x"), (2, "y")], dc.binary_program(OP.Lcm), {"x": x, "y": y})["res"].to_pylist()
    want = []
    for a, b in zip(x, y):
        g = gcd(int(a), int(b))
        m = wrap(int(a) * int(b))
        q = abs(m) // abs(g) * (1 if (m >= 0) == (g >= 0) else -1)
        want.append(wrap(q))
    assert got == want


def test_plan_compiler_lowers_the_direct_operations():
    """Host only: every operation type-checks against a schema and gets the reference's result type; the wrong types
    are refused with the reference's message."""
    from ydb_amd.program import ProgramContainer, SsaError
    p = Program()
    ops = [OP.Gcd, OP.Lcm, OP.Modulo, OP.ModuloOrZero]
    for i, op in enumerate(ops):
        p.assign_op(Column(name=f"b{i}"), op, [1, 2])
    unary = [OP.Abs, OP.Negate, OP.Round, OP.RoundBankers, OP.RoundToExp2, OP.Floor, OP.Ceil, OP.Trunc, OP.Sqrt, OP.IsValid]
    for i, op in enumerate(unary):
        p.assign_op(Column(name=f"u{i}"), op, [1])
    p.project([f"b{i}" for i in range(len(ops))] + [f"u{i}" for i in range(len(unary))][:6])
    pc = ProgramContainer()
    ok, err = pc.Init([(1, "x"), (2, "y")], p)
    assert ok, err
    text = pc.Explain({"x": np.zeros(4, dtype=np.int16), "y": np.zeros(4, dtype=np.uint8)})
    assert "b0" in text and "u5" in text
    pc = ProgramContainer()
    ok, err = pc.Init([(1, "x"), (2, "y")], dc.binary_program(OP.Gcd))
    assert ok, err
    with pytest.raises(SsaError) as e:
        pc.Explain({"x": np.zeros(4, dtype=np.float64), "y": np.zeros(4, dtype=np.float64)})
    assert "no kernel matching input types" in str(e.value)
    ok, err = ProgramContainer().Init([(1, "x")], dc.unary_program(OP.Abs).encode(), olap_wrapper=False)
    assert not ok


def test_oracle_math_functions_against_python_libm():
    """func_math.h kernels: the oracle calls the C library like the reference does; Python's math module is the same
    library for doubles.  ut_math.cpp:19-52 for the argument types."""
    dc.check_acosh_vectors(oracle_run)
    rng = np.random.default_rng(21)
    for op, fn, (lo, hi) in dc.MATH_OPS:
        x = dc.math_argument(rng, "float64", 500, lo, hi)
        got = oracle_run([(1, "x")], dc.unary_program(op), {"x": x})["res"].to_pylist()
        own = op in (OP.Lgamma, OP.Tgamma)   # Python computes these two itself
        assert all(dc.close(a, fn(float(v)), 1e-11 if own else 1e-13, 1e-12 if own else 1e-300) for a, v in zip(got, x)), op
    x, y = rng.normal(0, 1e3, 500), rng.normal(0, 1e3, 500)
    got = oracle_run([(1, "x"), (2, "y")], dc.binary_program(OP.Hypot), {"x": x, "y": y})["res"].to_pylist()
    assert all(dc.close(a, math.hypot(p, q), 1e-15, 0) for a, p, q in zip(got, x, y))
    with pytest.raises(ssa_oracle.OracleError) as e:
        oracle_run([(1, "x")], dc.unary_program(OP.Exp10), {"x": x})
    assert "No function registered" in str(e.value)


def test_oracle_checked_arithmetic_against_python_integers():
    dc.check_checked_arithmetic(oracle_run, oracle_fails)


def test_oracle_block_if():
    dc.check_block_if(oracle_run, oracle_fails)

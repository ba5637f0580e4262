"""The CPU oracle against the known-answer vectors of the reference's own unit tests (CPU only)."""
import pytest

from cases import GOLDEN_CHECKS
from oracle import ssa_oracle


def oracle_run(resolver, program, batch):
    return ssa_oracle.Oracle(resolver, program).apply(batch)


@pytest.mark.parametrize("name", sorted(GOLDEN_CHECKS))
def test_oracle_matches_reference_vector(name, golden):
    GOLDEN_CHECKS[name](golden[name], oracle_run)


def test_oracle_decimal128_sum_matches_python_integers():
    """decimal128 SUM has no counterpart in the reference (parity unpinned): the oracle's definition -- wrapping
    two's-complement 128-bit sum, null when no value is valid -- is checked against Python integers."""
    import numpy as np
    from ydb_amd.arrow_c import Col
    from ydb_amd.ssa_pb import AGG, Aggregate, Program
    rng = np.random.default_rng(5)
    n = 5000
    vals = [int(rng.integers(-(1 << 62), 1 << 62)) * int(rng.integers(1, 1 << 40)) for _ in range(n)]
    words = np.array([[v & ((1 << 64) - 1), (v >> 64) & ((1 << 64) - 1)] for v in vals], dtype=np.uint64)
    valid = rng.random(n) > 0.1
    key = rng.integers(0, 7, n).astype(np.int32)
    batch = {"d": Col(words, valid=valid, decimal=(38, 0)), "k": key}
    res = [(1, "d"), (2, "k")]
    out = oracle_run(res, Program().group_by([Aggregate(10, AGG.SUM, [1])]), batch)
    want = sum(v for v, ok in zip(vals, valid) if ok)
    assert out["10"].to_pylist() == [(want + (1 << 127)) % (1 << 128) - (1 << 127)]
    out = oracle_run(res, Program().group_by([Aggregate(10, AGG.SUM, [1])], keys=[2]), batch)
    for k, s in zip(out["k"].to_pylist(), out["10"].to_pylist()):
        assert s == sum(v for v, ok, kk in zip(vals, valid, key) if ok and kk == k)
    allnull = {"d": Col(words, valid=np.zeros(n, dtype=bool), decimal=(38, 0)), "k": key}
    assert oracle_run(res, Program().group_by([Aggregate(10, AGG.SUM, [1])]), allnull)["10"].to_pylist() == [None]


def test_general_like_patterns_against_the_reference_regex():
    """match_like with patterns Arrow hands to RE2: the oracle's token matcher against Python's `re` evaluating the
    regular expression the reference builds for the pattern (tests/like_cases.py)."""
    import numpy as np
    import like_cases
    from oracle import ssa_oracle
    from ydb_amd.arrow_c import Col
    from ydb_amd.ssa_pb import F, Column, Program
    rng = np.random.default_rng(3)
    values = like_cases.strings(rng, 4000) + ["abc", "a\nc", "ab%c", "ab_cd", "\\", "a", "日本", "x日本y", "aébc"]
    for pattern in like_cases.PATTERNS:
        prog = (Program().assign_const(Column(name="pat"), "Bytes", pattern.encode())
                .assign_func(Column(name="res"), F.STR_MATCH_LIKE, [1, "pat"]).project(["res"]))
        out = ssa_oracle.Oracle([(1, "x")], prog).apply({"x": Col(values, type="string")})
        got = [bool(v) for v in out["res"].to_pylist()]
        assert got == like_cases.expected(values, pattern), pattern

"""The unary YQL kernels on the GPU through ssa_program_create_with_kernels (parity unpinned: see yql_unary_cases)."""
import numpy as np
import pytest

import yql_unary_cases as uc
from ydb_amd.arrow_c import Col
from ydb_amd.ssa_pb import KERNEL_UNARY_BASE, YQL, YQL_UNARY, Program

pytestmark = pytest.mark.gpu


def gpu_run(resolver, program, batch, kernel_ops):
    from ydb_amd.program import ProgramContainer
    pc = ProgramContainer()
    ok, err = pc.Init(resolver, program, kernel_ops=kernel_ops)
    assert ok, err
    return pc.ApplyProgram(batch)


def gpu_fails(resolver, program, batch, kernel_ops):
    from ydb_amd.program import ProgramContainer, SsaError
    pc = ProgramContainer()
    ok, err = pc.Init(resolver, program, kernel_ops=kernel_ops)
    assert ok, err
    with pytest.raises(SsaError) as e:
        pc.ApplyProgram(batch)
    return str(e.value)


@pytest.mark.parametrize("check", uc.ALL, ids=lambda f: f.__name__)
def test_gpu(check):
    check(gpu_run, gpu_fails)


def test_a_pushed_down_not_like_over_a_large_batch_matches_the_oracle():
    """SELECT count(*), sum(v) WHERE NOT (s LIKE '%ab%') GROUP BY k -- the shape ClickBench's NOT LIKE filters take."""
    from cases import compare_batches
    from oracle import ssa_oracle
    from ydb_amd.ssa_pb import AGG, Aggregate
    rng = np.random.default_rng(9)
    n = 300_000
    words = ["", "ab", "xaby", "zzz", "ba", "a", "b", "abab", "q" * 40]
    s = Col([words[i] if i >= 0 else None for i in rng.integers(-1, len(words), n)], type="string")
    batch = {"s": s, "k": rng.integers(0, 5, n).astype(np.int16), "v": rng.integers(-100, 100, n).astype(np.int64)}
    prog = (Program(version=5).assign_const(100, "Bytes", b"ab").assign_yql(110, YQL.StringContains, [1, 100], kernel_idx=0)
            .assign_yql(111, None, [110], kernel_idx=1).assign_const(112, "Uint8", 0).assign_yql(113, YQL.Coalesce, [111, 112], kernel_idx=2)
            .filter(113).group_by([Aggregate(120, AGG.COUNT, []), Aggregate(121, AGG.SUM, [3])], keys=[2]))
    resolver = [(1, "s"), (2, "k"), (3, "v")]
    ops = [YQL.StringContains, KERNEL_UNARY_BASE + YQL_UNARY.Not, YQL.Coalesce]
    got = gpu_run(resolver, prog, batch, ops)
    want = ssa_oracle.Oracle(resolver, prog, kernel_ops=ops).apply(batch)
    compare_batches(got, want, key_names=["k"])
    keep = np.array([w is not None and "ab" not in w for w in s.values])
    assert sum(got["120"].to_pylist()) == int(keep.sum())

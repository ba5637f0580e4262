"""The unary YQL kernels in the CPU oracle (parity unpinned: see yql_unary_cases)."""
import pytest

import yql_unary_cases as uc
from oracle import ssa_oracle


def oracle_run(resolver, program, batch, kernel_ops):
    return ssa_oracle.Oracle(resolver, program, kernel_ops=kernel_ops).apply(batch)


def oracle_fails(resolver, program, batch, kernel_ops):
    with pytest.raises(ssa_oracle.OracleError) as e:
        ssa_oracle.Oracle(resolver, program, kernel_ops=kernel_ops).apply(batch)
    return str(e.value)


@pytest.mark.parametrize("check", uc.ALL, ids=lambda f: f.__name__)
def test_oracle(check):
    check(oracle_run, oracle_fails)

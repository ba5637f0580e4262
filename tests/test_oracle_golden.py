"""The CPU oracle against the known-answer vectors of the reference's own unit tests (CPU only)."""
import pytest

from cases import GOLDEN_CHECKS
from oracle import ssa_oracle


def oracle_run(resolver, program, batch):
    return ssa_oracle.Oracle(resolver, program).apply(batch)


@pytest.mark.parametrize("name", sorted(GOLDEN_CHECKS))
def test_oracle_matches_reference_vector(name, golden):
    GOLDEN_CHECKS[name](golden[name], oracle_run)

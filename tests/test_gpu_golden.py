"""The GPU executor, through the C-ABI, against the known-answer vectors of the reference's own unit tests."""
import pytest

from cases import GOLDEN_CHECKS

pytestmark = pytest.mark.gpu


def gpu_run(resolver, program, batch):
    from ydb_amd.program import ProgramContainer
    pc = ProgramContainer()
    ok, err = pc.Init(resolver, program)
    assert ok, err
    return pc.ApplyProgram(batch)


@pytest.mark.parametrize("name", sorted(GOLDEN_CHECKS))
def test_gpu_matches_reference_vector(name, golden):
    GOLDEN_CHECKS[name](golden[name], gpu_run)

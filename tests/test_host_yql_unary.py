"""ssa_program_create_with_kernels without a GPU: parsing, the table's checks, planning against a schema."""
import numpy as np
import pytest

from ydb_amd.arrow_c import Col
from ydb_amd.program import ProgramContainer, SsaError
from ydb_amd.ssa_pb import KERNEL_UNARY_BASE, KERNEL_UNKNOWN, YQL_UNARY, Column, Program


def unary(kernel_idx):
    return Program(version=5).assign_yql(Column(name="res"), None, [1], kernel_idx=kernel_idx).project(["res"])


def test_a_unary_kernel_is_unknown_until_the_table_names_it():
    pc = ProgramContainer()
    ok, err = pc.Init([(1, "x")], unary(3))
    assert not ok and "Unknown kernel" in err and "kernel_idx=3" in err
    ok, err = pc.Init([(1, "x")], unary(3), kernel_ops=[KERNEL_UNKNOWN] * 4)
    assert not ok and "Unknown kernel" in err
    ok, err = pc.Init([(1, "x")], unary(3), kernel_ops=[KERNEL_UNKNOWN] * 3 + [KERNEL_UNARY_BASE + YQL_UNARY.Abs])
    assert ok, err
    assert "kernel=3" in pc.DebugString()


def test_a_table_entry_that_names_nothing_is_refused():
    pc = ProgramContainer()
    with pytest.raises(SsaError):
        pc.Init([(1, "x")], unary(0), kernel_ops=[KERNEL_UNARY_BASE + 5])
    with pytest.raises(SsaError):
        pc.Init([(1, "x")], unary(0), kernel_ops=[18])


def test_planning_types_the_results():
    pc = ProgramContainer()
    for op, batch, want in ((YQL_UNARY.Size, {"x": Col(["a"], type="string")}, "uint32"), (YQL_UNARY.Minus, {"x": np.zeros(1, np.int8)}, "int8"),
                            (YQL_UNARY.Abs, {"x": np.zeros(1, np.float32)}, "float"), (YQL_UNARY.Not, {"x": np.zeros(1, np.uint8)}, "uint8")):
        ok, err = pc.Init([(1, "x")], unary(0), kernel_ops=[KERNEL_UNARY_BASE + op])
        assert ok, err
        assert want in pc.Explain(batch), (op, pc.Explain(batch))

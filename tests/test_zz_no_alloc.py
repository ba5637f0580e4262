"""ssa_filter_from_program_noalloc (the entry point that allocates no device memory) against ssa_filter_from_program,
through the C-ABI: same bitmap and same count on two calls in a row, the second over a smaller input (the static
scratch buffer keeps the first call's contents)."""
import ctypes

import numpy as np
import pytest

from ydb_amd.arrow_c import Col
from ydb_amd.ssa_pb import F, Column, Program

pytestmark = pytest.mark.gpu


def _program():
    p = Program()
    p.assign_const(10, "Int32", 5).assign_const(12, "Bytes", b"ab")
    p.assign_func(11, F.CMP_GREATER_EQUAL, [Column(id=1), Column(id=10)])
    p.assign_func(13, F.STR_MATCH, [3, 12])
    p.filter(11).filter(13)
    p.project([2])
    return p


def _batch(rng, n):
    words = ["", "ab", "xxabyy", "ba", "a" * 20 + "b"]
    return {"a": Col(rng.integers(0, 10, n).astype(np.int32), valid=rng.random(n) > 0.1), "b": rng.random(n),
            "s": [words[i] for i in rng.integers(0, len(words), n)]}


def _noalloc(pc, dev, rows):
    import torch
    from ydb_amd import _lib
    nwords = (rows + 63) // 64
    out = torch.full((nwords + 8,), -1, dtype=torch.int64, device="cuda:0")   # guard words behind the bitmap
    torch.cuda.synchronize()
    selected, err = ctypes.c_int64(-1), ctypes.c_void_p()
    rc = pc._lib.ssa_filter_from_program_noalloc(pc._h, 0, dev._h, out.data_ptr(), ctypes.byref(selected), ctypes.byref(err))
    msg = _lib.take_string(pc._lib, err.value)
    assert rc == _lib.SSA_OK, msg
    host = out.cpu().numpy()
    assert (host[nwords:] == -1).all(), "wrote past the bitmap"
    bits = np.unpackbits(host[:nwords].view(np.uint8), bitorder="little")[:rows].astype(np.bool_)
    return bits, selected.value


def test_noalloc_filter_equals_filter_from_program():
    from ydb_amd.column_filter import ColumnFilter
    from ydb_amd.program import ProgramContainer
    rng = np.random.default_rng(11)
    pc = ProgramContainer()
    ok, err = pc.Init([(1, "a"), (2, "b"), (3, "s")], _program())
    assert ok, err
    for n in (300_000, 70_001, 1025, 1):   # every call after the first is smaller than the one before
        batch = _batch(rng, n)
        dev = pc.Import(batch)
        want = ColumnFilter.FromProgram(pc, dev, 0)
        want_bits = want.BuildSimpleFilter()
        got_bits, got_count = _noalloc(pc, dev, n)
        assert got_count == want.GetFilteredCount() == int(want_bits.sum())
        assert np.array_equal(got_bits, want_bits)
        mask, count = pc.BuildFilter(batch, 0)      # and the host-side seam
        assert count == got_count and np.array_equal(mask, got_bits)


def test_noalloc_filter_limits():
    import torch
    from ydb_amd import _lib
    from ydb_amd.program import ProgramContainer, device_batch
    pc = ProgramContainer()
    p = Program().assign_const(10, "Int32", 5).assign_func(11, F.CMP_GREATER_EQUAL, [1, 10]).filter(11).project([1])
    assert pc.Init([(1, "a")], p)[0]
    rows = (1 << 26) + 1
    a = torch.zeros(rows + 16, dtype=torch.int32, device="cuda:0")
    dev = device_batch([("a", _lib.SSA_T_INT32 if hasattr(_lib, "SSA_T_INT32") else 4, a.data_ptr(), 0, 0, 0, 0)], rows, keep=a)
    out = torch.zeros(((rows + 63) // 64) + 8, dtype=torch.int64, device="cuda:0")
    torch.cuda.synchronize()
    selected, err = ctypes.c_int64(-1), ctypes.c_void_p()
    rc = pc._lib.ssa_filter_from_program_noalloc(pc._h, 0, dev._h, out.data_ptr(), ctypes.byref(selected), ctypes.byref(err))
    msg = _lib.take_string(pc._lib, err.value)
    assert rc == _lib.SSA_INVALID and "at most" in msg, (rc, msg)
    # the largest batch it takes
    rows = 1 << 26
    a[:rows] = torch.arange(rows, dtype=torch.int32, device="cuda:0") % 10
    dev = device_batch([("a", 4, a.data_ptr(), 0, 0, 0, 0)], rows, keep=a)
    torch.cuda.synchronize()
    rc = pc._lib.ssa_filter_from_program_noalloc(pc._h, 0, dev._h, out.data_ptr(), ctypes.byref(selected), ctypes.byref(err))
    assert rc == _lib.SSA_OK, _lib.take_string(pc._lib, err.value)
    assert selected.value == int((torch.arange(rows, dtype=torch.int32, device="cuda:0") % 10 >= 5).sum().item())
    # a step without a Filter command
    pc2 = ProgramContainer()
    assert pc2.Init([(1, "a")], Program().project([1]))[0]
    rc = pc2._lib.ssa_filter_from_program_noalloc(pc2._h, 0, dev._h, out.data_ptr(), ctypes.byref(selected), ctypes.byref(err))
    assert rc == _lib.SSA_INVALID, _lib.take_string(pc2._lib, err.value)

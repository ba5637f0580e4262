"""Q1 (utf8 flags, 1-byte codes) and Q6 over 64 slices of 2^20 rows of one resident table: ssa_program_apply_many against
one ssa_program_apply_device per slice.  Run from the repository root: PYTHONPATH=. python scripts/bench_slices.py"""
import sys, time
import numpy as np
import torch
from ydb_amd import tpch
from ydb_amd.program import ProgramContainer
per, nslices = 1 << 20, 64
li = tpch.DeviceLineitem(per * nslices)
for name, codes, cols in (("utf8", False, tpch.Q1_COLUMNS_UTF8), ("codes", True, tpch.Q1_COLUMNS_CODES), ("q6", None, tpch.Q6_COLUMNS)):
    prog = tpch.q6_program() if codes is None else tpch.q1_program(codes)
    pc = ProgramContainer()
    ok, err = pc.Init(tpch.LINEITEM_RESOLVER, prog)
    assert ok, err
    slices = [li.batch(cols, row0=i * per, rows=per) for i in range(nslices)]
    for _ in range(4):
        res = pc.ApplyDeviceMany(slices)
    torch.cuda.synchronize()
    pc.EnableProfile(True); pc.ReadProfile()
    t0 = time.perf_counter()
    K = 20
    for _ in range(K):
        res = pc.ApplyDeviceMany(slices)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    ms, nl, ab = pc.ReadProfile()
    print(name, "many: %.3f ms/step, %.1f G rows/s; kernel %.3f ms avg, %.0f GB/s" % (dt * 1e3, per * nslices / dt / 1e9, ms / max(nl, 1), ab / max(ms, 1e-9) / 1e6), flush=True)
    pc.EnableProfile(False)
    t0 = time.perf_counter()
    for _ in range(3):
        for s in slices:
            r = pc.ApplyDevice(s)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print(name, "one by one: %.3f ms/step, %.1f G rows/s" % (dt * 1e3, per * nslices / dt / 1e9), flush=True)

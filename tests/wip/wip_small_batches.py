"""Cost of one apply at the batch sizes a ColumnShard hands over (1e4 .. 1e7 rows), Q1 and Q6, HBM-resident input."""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import time
import torch
from ydb_amd.program import ProgramContainer
from ydb_amd import tpch
os.environ.setdefault("SSA_JIT", "1")
for name, prog, cols, kw in (("q6", tpch.q6_program(), tpch.Q6_COLUMNS, dict(utf8_flags=False, code_flags=False)),
                             ("q1", tpch.q1_program(False), tpch.Q1_COLUMNS_UTF8, dict(utf8_flags=True, code_flags=False))):
    pc = ProgramContainer()
    ok, err = pc.Init(tpch.LINEITEM_RESOLVER, prog); assert ok, err
    for n in (10_000, 100_000, 1_000_000, 10_000_000):
        li = tpch.DeviceLineitem(n, **kw)
        b = li.batch(cols)
        for _ in range(5):
            pc.ApplyDevice(b)
        pc.EnableProfile(True); pc.ReadProfile()
        torch.cuda.synchronize()
        reps = 200
        t0 = time.perf_counter()
        for _ in range(reps):
            pc.ApplyDevice(b)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        ms, launches, _ = pc.ReadProfile()
        pc.EnableProfile(False)
        t0 = time.perf_counter()
        for _ in range(reps):
            pc.ApplyDevice(b)
        torch.cuda.synchronize()
        dt2 = (time.perf_counter() - t0) / reps
        print(f"{name} rows {n}: {1e6*dt2:.0f} us per apply ({1e6*dt:.0f} with event timing), scan kernel {1e3*ms/max(1,launches):.0f} us, {n/dt2/1e9:.2f} G rows/s")

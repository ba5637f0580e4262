"""Throughput of the drop-in seam with host batches (ssa_program_apply: upload + program + download), Q1 and Q6."""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import time
import numpy as np
from ydb_amd.program import ProgramContainer
from ydb_amd.arrow_c import HostBatch
from ydb_amd import tpch
os.environ.setdefault("SSA_JIT", "1")
for name, prog, cols, bpr in (("q6", tpch.q6_program(), tpch.Q6_COLUMNS, 28), ("q1_codes", tpch.q1_program(True), tpch.Q1_COLUMNS_CODES, 38),
                              ("q1", tpch.q1_program(False), tpch.Q1_COLUMNS_UTF8, 46)):
    pc = ProgramContainer()
    ok, err = pc.Init(tpch.LINEITEM_RESOLVER, prog); assert ok, err
    for n in (100_000, 1_000_000, 8_000_000):
        li = tpch.lineitem_host(0, n)
        hb = HostBatch({c: li[c] for c in cols})
        for _ in range(3):
            pc.ApplyProgram(hb)
        reps = max(3, min(50, 20_000_000 // n))
        t0 = time.perf_counter()
        for _ in range(reps):
            pc.ApplyProgram(hb)
        dt = (time.perf_counter() - t0) / reps
        t0 = time.perf_counter()
        for _ in range(reps):
            d = pc.Import(hb)
        dti = (time.perf_counter() - t0) / reps
        print(f"{name} rows {n}: apply {1e3*dt:.3f} ms = {n/dt/1e9:.3f} G rows/s, {n*bpr/dt/1e9:.1f} GB/s over PCIe; import alone {1e3*dti:.3f} ms")

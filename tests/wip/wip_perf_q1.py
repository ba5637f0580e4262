import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import time, sys
import numpy as np, torch
from ydb_amd.program import ProgramContainer
from ydb_amd import tpch

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64_000_000
codes = (sys.argv[2] == 'codes') if len(sys.argv) > 2 else True
li = tpch.DeviceLineitem(n, utf8_flags=not codes, code_flags=codes)
pc = ProgramContainer()
ok, err = pc.Init(tpch.LINEITEM_RESOLVER, tpch.q1_program(codes)); assert ok, err
cols = tpch.Q1_COLUMNS_CODES if codes else tpch.Q1_COLUMNS_UTF8
b = li.batch(cols)
out = pc.Export(pc.ApplyDevice(b))
print("q1 groups", len(out['130'].values), out['138'].to_pylist())
pc.EnableProfile(True)
for it in range(3):
    t0 = time.time()
    r = pc.ApplyDevice(b)
    t1 = time.time()
    ms, launches, nbytes = pc.ReadProfile()
    print(f"iter {it}: wall {1e3*(t1-t0):.3f} ms kernel {ms:.3f} ms launches {launches} bytes {nbytes} -> {nbytes/ms/1e6:.1f} GB/s, {n/ms/1e6:.2f} Grows/s")

import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import time, sys
import numpy as np, torch
from ydb_amd.program import ProgramContainer
from ydb_amd import tpch

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64_000_000
li = tpch.DeviceLineitem(n, utf8_flags=False, code_flags=False)
# parity of generator on a prefix
h = tpch.lineitem_host(0, 4096)
for k in ("l_shipdate", "l_quantity", "l_extendedprice", "l_discount", "l_tax"):
    d = li.t[k][:4096].cpu().numpy()
    assert (d == h[k]).all(), k
print("generator parity ok")
pc = ProgramContainer()
ok, err = pc.Init(tpch.LINEITEM_RESOLVER, tpch.q6_program()); assert ok, err
b = li.batch(tpch.Q6_COLUMNS)
out = pc.Export(pc.ApplyDevice(b))
print("q6 result", out['130'].to_pylist())
pc.EnableProfile(True)
for it in range(3):
    t0 = time.time()
    r = pc.ApplyDevice(b)
    t1 = time.time()
    ms, launches, nbytes = pc.ReadProfile()
    print(f"iter {it}: wall {1e3*(t1-t0):.3f} ms kernel {ms:.3f} ms launches {launches} bytes {nbytes} -> {nbytes/ms/1e6:.1f} GB/s, {n/ms/1e6:.2f} Grows/s")

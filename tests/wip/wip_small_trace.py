import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ydb_amd.program import ProgramContainer
from ydb_amd import tpch
os.environ.setdefault("SSA_JIT", "1")
pc = ProgramContainer()
ok, err = pc.Init(tpch.LINEITEM_RESOLVER, tpch.q1_program(False)); assert ok, err
li = tpch.DeviceLineitem(100_000, utf8_flags=True, code_flags=False)
b = li.batch(tpch.Q1_COLUMNS_UTF8)
for _ in range(300):
    pc.ApplyDevice(b)
torch.cuda.synchronize()

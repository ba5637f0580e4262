import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ydb_amd.program import ProgramContainer
from ydb_amd import tpch
n = int(sys.argv[1])
li = tpch.DeviceLineitem(n, utf8_flags=False, code_flags=False)
pc = ProgramContainer()
ok, err = pc.Init(tpch.LINEITEM_RESOLVER, tpch.q6_program()); assert ok, err
b = li.batch(tpch.Q6_COLUMNS)
for _ in range(2):
    pc.ApplyDevice(b)

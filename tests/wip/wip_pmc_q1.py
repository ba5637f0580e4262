import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import sys
from ydb_amd.program import ProgramContainer
from ydb_amd import tpch
n = int(sys.argv[1]); codes = sys.argv[2] == 'codes'
li = tpch.DeviceLineitem(n, utf8_flags=not codes, code_flags=codes)
pc = ProgramContainer()
ok, err = pc.Init(tpch.LINEITEM_RESOLVER, tpch.q1_program(codes)); assert ok, err
b = li.batch(tpch.Q1_COLUMNS_CODES if codes else tpch.Q1_COLUMNS_UTF8)
for _ in range(2):
    pc.ApplyDevice(b)

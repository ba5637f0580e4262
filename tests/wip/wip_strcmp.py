import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import time, torch
from ydb_amd import clickbench
from ydb_amd.program import ProgramContainer
from ydb_amd.ssa_pb import AGG, YQL, F, Aggregate, Column, Program
n = 99_997_497
t = clickbench.DeviceHits(n, columns=["SearchPhrase"])
b = t.batch(["SearchPhrase"])
progs = {
 "phrase <> ''": Program(version=5).assign_const(106, "Bytes", b"").assign_yql(107, YQL.NotEquals, [40, 106]).filter(107).group_by([Aggregate(108, AGG.COUNT, [])]).project([108]),
 "phrase == 'abc'": Program(version=5).assign_const(106, "Bytes", b"abcdefghij").assign_yql(107, YQL.Equals, [40, 106]).filter(107).group_by([Aggregate(108, AGG.COUNT, [])]).project([108]),
 "starts_with 'ab'": Program(version=5).assign_const(106, "Bytes", b"ab").assign_yql(107, YQL.StartsWith, [40, 106]).filter(107).group_by([Aggregate(108, AGG.COUNT, [])]).project([108]),
 "length": Program().assign_func(107, F.STR_LENGTH, [40]).group_by([Aggregate(108, AGG.SUM, [107])]),
 "contains 'goo'": Program(version=5).assign_const(106, "Bytes", b"goo").assign_yql(107, YQL.StringContains, [40, 106]).filter(107).group_by([Aggregate(108, AGG.COUNT, [])]).project([108]),
}
nbytes = 4 * n + t.phrase_bytes
for name, prog in progs.items():
    pc = ProgramContainer(); ok, err = pc.Init(clickbench.HITS_RESOLVER, prog); assert ok, err
    for _ in range(3): r = pc.ApplyDevice(b)
    pc.EnableProfile(True); pc.ReadProfile()
    for _ in range(5): r = pc.ApplyDevice(b)
    ms, launches, ab = pc.ReadProfile()
    out = pc.Export(r)
    print(f"{name}: kernel {ms/launches:.3f} ms, {ab/launches/(ms/launches)/1e6:.0f} GB/s, {n/(ms/launches)/1e6:.1f} G rows/s, result {list(out.values())[0].to_pylist()}")

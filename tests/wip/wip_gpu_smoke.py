import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
"""Bring-up checks of the scan path against numpy (run by hand on the GPU box)."""
import numpy as np
from ydb_amd.program import ProgramContainer
from ydb_amd.ssa_pb import *
from ydb_amd.arrow_c import Col

rng = np.random.default_rng(1)

def run(prog, resolver, batch):
    pc = ProgramContainer()
    ok, err = pc.Init(resolver, prog)
    assert ok, err
    return pc.ApplyProgram(batch)

for n in (0, 1, 5, 1000, 1024, 5000, 1_000_003):
    ship = rng.integers(8036, 10562, n).astype(np.int32)
    disc = (rng.integers(0, 11, n) / 100.0)
    qty = rng.integers(1, 51, n).astype(np.float64)
    price = rng.integers(90000, 200000, n) / 100.0
    res = [(1, 'l_shipdate'), (2, 'l_discount'), (3, 'l_quantity'), (4, 'l_extendedprice')]
    p = (Program()
         .assign_const(100, 'Int32', 9131).assign_const(101, 'Int32', 9496)
         .assign_const(102, 'Double', 0.07 - 0.01).assign_const(103, 'Double', 0.07 + 0.01).assign_const(104, 'Double', 25.0)
         .assign_func(110, F.CMP_GREATER_EQUAL, [1, 100]).assign_func(111, F.CMP_LESS, [1, 101])
         .assign_func(112, F.CMP_GREATER_EQUAL, [2, 102]).assign_func(113, F.CMP_LESS_EQUAL, [2, 103])
         .assign_func(114, F.CMP_LESS, [3, 104])
         .filter(110).filter(111).filter(112).filter(113).filter(114)
         .assign_func(120, F.MATH_MULTIPLY, [4, 2])
         .group_by([Aggregate(130, AGG.SUM, [120]), Aggregate(131, AGG.COUNT, []), Aggregate(132, AGG.MIN, [1]), Aggregate(133, AGG.MAX, [4])])
         .project([130, 131, 132, 133]))
    out = run(p, res, {'l_shipdate': ship, 'l_discount': disc, 'l_quantity': qty, 'l_extendedprice': price})
    m = (ship >= 9131) & (ship < 9496) & (disc >= 0.07 - 0.01) & (disc <= 0.07 + 0.01) & (qty < 25.0)
    exp = float((price[m] * disc[m]).sum())
    got = out['130']
    print(n, 'sum', got.to_pylist(), exp, 'count', out['131'].to_pylist(), int(m.sum()), 'min', out['132'].to_pylist(),
          ship[m].min() if m.any() else None, 'max', out['133'].to_pylist(), price[m].max() if m.any() else None)
    if m.any():
        assert abs(got.values[0] - exp) <= 1e-12 * abs(exp), (got.values[0], exp)
        assert out['131'].values[0] == m.sum()
        assert out['132'].values[0] == ship[m].min() and out['133'].values[0] == price[m].max()
        assert out['130'].type == 'double' and out['131'].type == 'uint64' and out['132'].type == 'int32'
    else:
        assert not got.valid[0] and out['131'].values[0] == 0

    # filter + projection (compaction) with a computed column and a nullable input
    valid = rng.random(n) > 0.1
    p2 = (Program().assign_const(100, 'Int32', 9131).assign_func(110, F.CMP_GREATER_EQUAL, [1, 100]).filter(110)
          .assign_func(120, F.MATH_ADD, [1, 100]).project([120, 2, 1]))
    out2 = run(p2, res, {'l_shipdate': ship, 'l_discount': Col(disc, valid=valid), 'l_quantity': qty})
    m2 = ship >= 9131
    assert len(out2['120'].values) == m2.sum(), (len(out2['120'].values), m2.sum())
    assert (out2['120'].values == ship[m2] + 9131).all()
    assert (out2['l_shipdate'].values == ship[m2]).all()
    assert (out2['l_discount'].valid == valid[m2]).all()
    assert (out2['l_discount'].values[valid[m2]] == disc[m2][valid[m2]]).all()
    print('  project ok', m2.sum())
    # strings: filter on starts_with + projection of the string column
    strs = [("ab" if i % 3 == 0 else "xyz") + str(i % 7) for i in range(min(n, 5000))]
    if strs:
        p3 = (Program().assign_const(100, 'Bytes', b'ab').assign_func(110, F.STR_STARTS_WITH, ['s', 100]).filter(110)
              .assign_func(111, F.STR_LENGTH, ['s']).project(['s', 111]))
        out3 = run(p3, [(1, 's')], {'s': strs})
        exp3 = [s.encode() for s in strs if s.startswith('ab')]
        assert out3['s'].values == exp3, (out3['s'].values[:5], exp3[:5])
        assert (out3['111'].values == [len(s) for s in exp3]).all()
        print('  strings ok', len(exp3))
print("ALL OK")

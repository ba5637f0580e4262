import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
"""Bring-up checks of keyed GROUP BY against pandas (run by hand on the GPU box)."""
import numpy as np
import pandas as pd
from ydb_amd.program import ProgramContainer
from ydb_amd.ssa_pb import *
from ydb_amd.arrow_c import Col
from ydb_amd import tpch

rng = np.random.default_rng(7)


def run(prog, resolver, batch):
    pc = ProgramContainer()
    ok, err = pc.Init(resolver, prog)
    assert ok, err
    return pc.ApplyProgram(batch)


def frame(out, names):
    d = {}
    for n in names:
        c = out[n]
        v = c.values
        if isinstance(v, list):
            v = [x.decode() if x is not None else None for x in v]
            d[n] = v
        else:
            v = np.array(v).astype(object)
            if c.valid is not None:
                v[~np.asarray(c.valid, dtype=bool)] = None
            d[n] = v
    return pd.DataFrame(d)


# ---- Q1 on the synthetic lineitem, both key encodings
for n in (1, 1000, 300_000):
    li = tpch.lineitem_host(0, n)
    for codes in (True, False):
        cols = tpch.Q1_COLUMNS_CODES if codes else tpch.Q1_COLUMNS_UTF8
        out = run(tpch.q1_program(codes), tpch.LINEITEM_RESOLVER, {c: li[c] for c in cols})
        k1, k2 = cols[-2], cols[-1]
        df = pd.DataFrame({c: li[c] for c in cols})
        df = df[df.l_shipdate <= tpch.Q1_DATE]
        df['dp'] = df.l_extendedprice * (1.0 - df.l_discount)
        df['ch'] = df.dp * (1.0 + df.l_tax)
        exp = df.groupby([k1, k2]).agg(s_qty=('l_quantity', 'sum'), s_price=('l_extendedprice', 'sum'), s_dp=('dp', 'sum'),
                                       s_ch=('ch', 'sum'), s_disc=('l_discount', 'sum'), cnt=('dp', 'size')).reset_index()
        got = frame(out, ['130', '131', '132', '133', '134', '135', '136', '137', '138', k1, k2])
        got = got.sort_values([k1, k2]).reset_index(drop=True)
        exp = exp.sort_values([k1, k2]).reset_index(drop=True)
        assert len(got) == len(exp), (len(got), len(exp), got)
        assert (got[k1].values == exp[k1].values).all() and (got[k2].values == exp[k2].values).all(), (got, exp)
        for g, e in (('130', 's_qty'), ('131', 's_price'), ('132', 's_dp'), ('133', 's_ch'), ('136', 's_disc')):
            gv = got[g].values.astype(float); ev = exp[e].values.astype(float)
            assert (np.abs(gv - ev) <= 1e-12 * np.abs(ev)).all(), (g, gv, ev)
        for g in ('134', '135', '137', '138'):
            assert (got[g].values.astype(np.int64) == exp['cnt'].values).all(), (g, got[g].values, exp['cnt'].values)
        print('q1', n, 'codes' if codes else 'utf8', 'groups', len(got), 'ok')

# ---- high cardinality int64 key with nullable arguments; run twice (second run uses the learnt sizes)
n = 200_000
key = rng.integers(-25_000, 25_000, n).astype(np.int64)
val = rng.integers(-1000, 1000, n).astype(np.int32)
valid = rng.random(n) > 0.3
fval = rng.normal(size=n)
res = [(1, 'k'), (2, 'v'), (3, 'f')]
p = (Program().group_by([Aggregate(10, AGG.SUM, [2]), Aggregate(11, AGG.MIN, [2]), Aggregate(12, AGG.MAX, [2]),
                         Aggregate(13, AGG.COUNT, [2]), Aggregate(14, AGG.COUNT, []), Aggregate(15, AGG.MIN, [3]),
                         Aggregate(16, AGG.MAX, [3]), Aggregate(17, AGG.SUM, [3]), Aggregate(18, AGG.SOME, [2])], keys=[1]))
pc = ProgramContainer()
ok, err = pc.Init(res, p)
assert ok, err
for it in range(2):
    out = pc.ApplyProgram({'k': key, 'v': Col(val, valid=valid), 'f': fval})
    got = frame(out, ['10', '11', '12', '13', '14', '15', '16', '17', '18', 'k']).sort_values('k').reset_index(drop=True)
    df = pd.DataFrame({'k': key, 'v': np.where(valid, val, np.nan), 'f': fval})
    exp = df.groupby('k').agg(s=('v', lambda x: x.sum(min_count=1)), mn=('v', 'min'), mx=('v', 'max'), c=('v', 'count'),
                              n=('v', 'size'), fmn=('f', 'min'), fmx=('f', 'max'), fs=('f', 'sum'),
                              some=('v', 'first')).reset_index().sort_values('k').reset_index(drop=True)
    assert len(got) == len(exp), (len(got), len(exp))
    assert (got['k'].values.astype(np.int64) == exp['k'].values).all()
    for g, e in (('10', 's'), ('11', 'mn'), ('12', 'mx'), ('18', 'some')):
        gv = np.array([np.nan if x is None else float(x) for x in got[g].values]); ev = exp[e].values.astype(float)
        assert ((gv == ev) | (np.isnan(gv) & np.isnan(ev))).all(), (g, gv[:10], ev[:10])
    assert (got['13'].values.astype(np.int64) == exp['c'].values).all()
    assert (got['14'].values.astype(np.int64) == exp['n'].values).all()
    assert (got['15'].values.astype(float) == exp['fmn'].values).all() and (got['16'].values.astype(float) == exp['fmx'].values).all()
    gv = got['17'].values.astype(float); ev = exp['fs'].values
    assert (np.abs(gv - ev) <= 1e-9 * np.maximum(1.0, np.abs(ev))).all()
    print('high-cardinality pass', it, 'groups', len(got), 'ok')

# ---- nullable keys (null-aware packing), two keys
n = 50_000
k1 = rng.integers(0, 5, n).astype(np.int32); k1v = rng.random(n) > 0.2
k2 = rng.integers(0, 3, n).astype(np.int16)
v = rng.integers(0, 100, n).astype(np.uint32)
p = Program().group_by([Aggregate(10, AGG.SUM, [3]), Aggregate(11, AGG.COUNT, [])], keys=[1, 2])
out = run(p, [(1, 'a'), (2, 'b'), (3, 'v')], {'a': Col(k1, valid=k1v), 'b': k2, 'v': v})
got = frame(out, ['10', '11', 'a', 'b'])
df = pd.DataFrame({'a': np.where(k1v, k1, -1), 'b': k2, 'v': v.astype(np.int64)})
exp = df.groupby(['a', 'b']).agg(s=('v', 'sum'), n=('v', 'size')).reset_index()
got['a'] = [(-1 if x is None else int(x)) for x in got['a'].values]
got = got.sort_values(['a', 'b']).reset_index(drop=True); exp = exp.sort_values(['a', 'b']).reset_index(drop=True)
assert len(got) == len(exp) == 18, (len(got), len(exp))
assert (got['a'].values == exp['a'].values).all() and (got['b'].values.astype(int) == exp['b'].values).all()
# COUNT(*) is rewritten to COUNT(first key): the null-key groups count 0 rows
exp.loc[exp['a'] == -1, 'n'] = 0
assert (got['10'].values.astype(np.int64) == exp['s'].values).all() and (got['11'].values.astype(np.int64) == exp['n'].values).all()
assert out['10'].type == 'uint64', out['10'].type
print('nullable keys ok')

# ---- wide keys (hashed, fixed width) and a string + int key
n = 100_000
a = rng.integers(0, 7, n).astype(np.int64) * (1 << 40); b = rng.integers(0, 5, n).astype(np.uint64); c = rng.integers(0, 3, n).astype(np.int32)
p = Program().group_by([Aggregate(10, AGG.COUNT, []), Aggregate(11, AGG.MAX, [3])], keys=[1, 2, 3])
out = run(p, [(1, 'a'), (2, 'b'), (3, 'c')], {'a': a, 'b': b, 'c': c})
got = frame(out, ['10', '11', 'a', 'b', 'c']).astype(np.int64).sort_values(['a', 'b', 'c']).reset_index(drop=True)
exp = pd.DataFrame({'a': a, 'b': b.astype(np.int64), 'c': c}).groupby(['a', 'b', 'c']).agg(n=('c', 'size'), m=('c', 'max')).reset_index()
assert len(got) == len(exp) == 105
assert (got[['a', 'b', 'c']].values == exp[['a', 'b', 'c']].values).all() and (got['10'].values == exp['n'].values).all()
print('wide keys ok')
strs = ["key%d" % (i % 1000) + "x" * (i % 13) for i in range(n)]
out = run(Program().group_by([Aggregate(10, AGG.COUNT, []), Aggregate(11, AGG.SUM, [2])], keys=[1, 2]), [(1, 's'), (2, 'c')], {'s': strs, 'c': c})
got = frame(out, ['10', '11', 's', 'c']).sort_values(['s', 'c']).reset_index(drop=True)
exp = pd.DataFrame({'s': strs, 'c': c}).groupby(['s', 'c']).agg(n=('c', 'size'), m=('c', 'sum')).reset_index().sort_values(['s', 'c']).reset_index(drop=True)
assert len(got) == len(exp), (len(got), len(exp))
assert (got['s'].values == exp['s'].values).all() and (got['c'].values.astype(int) == exp['c'].values).all()
assert (got['10'].values.astype(np.int64) == exp['n'].values).all() and (got['11'].values.astype(np.int64) == exp['m'].values).all()
print('string keys ok', len(got))
# empty input
out = run(Program().group_by([Aggregate(10, AGG.COUNT, [])], keys=[1]), [(1, 'k')], {'k': np.zeros(0, dtype=np.int32)})
assert len(out['10'].values) == 0 and len(out['k'].values) == 0
print("ALL OK")

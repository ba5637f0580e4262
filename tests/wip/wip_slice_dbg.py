import numpy as np
from ydb_amd.arrow_c import Col
from ydb_amd.program import ProgramContainer
from ydb_amd.ssa_pb import AGG, Aggregate, Program
rng = np.random.default_rng(1)
n = 6000
v = rng.integers(-100, 100, n).astype(np.int64)
ok = rng.random(n) > 0.3
k = rng.integers(0, 5, n).astype(np.int32)
words = ["", "a", "bb", "a longer string than the rest of them", "ccc"]
sidx = rng.integers(0, 5, n)
sok = rng.random(n) > 0.2
s = [words[i] if o else None for i, o in zip(sidx, sok)]
RES = [(1, "k"), (2, "v"), (3, "s")]
for prog_name, p in (("nokey", Program().group_by([Aggregate(120, AGG.COUNT, [2]), Aggregate(121, AGG.SUM, [2])])),
                     ("key k", Program().group_by([Aggregate(120, AGG.COUNT, [2]), Aggregate(121, AGG.SUM, [2])], keys=[1])),
                     ("key s", Program().group_by([Aggregate(120, AGG.COUNT, [2]), Aggregate(121, AGG.SUM, [2])], keys=[3]))):
    pc = ProgramContainer()
    okk, err = pc.Init(RES, p)
    assert okk, err
    b = pc.Import({"k": k, "v": Col(v, valid=ok), "s": Col(s, type="string")})
    for row0, rows in ((0, 6000), (8, 4000), (64, 4000), (500, 4000), (1024, 3000), (3, 100)):
        got = pc.Export(pc.ApplyDevice(pc.Slice(b, row0, rows)))
        cnt = int(np.sum(got["120"].values))
        sm = int(np.sum(np.asarray(got["121"].values)[np.asarray(got["121"].valid, dtype=bool)]))
        sel = slice(row0, row0 + rows)
        print(prog_name, row0, rows, "count", cnt, int(ok[sel].sum()), "sum", sm, int(v[sel][ok[sel]].sum()), "groups", len(got["120"].values), flush=True)

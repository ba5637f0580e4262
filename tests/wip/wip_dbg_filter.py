import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from ydb_amd.program import ProgramContainer, device_batch
from ydb_amd.column_filter import ColumnFilter
from ydb_amd.ssa_pb import Program
pc = ProgramContainer(); ok, err = pc.Init([(1, "pk")], Program().project([1])); assert ok, err
dev = torch.device("cuda", 0)
for n in (100_000_000, 200_000_000, 600_037_902):
    pk = torch.arange(n, dtype=torch.int64, device=dev) // 7
    keys = device_batch([("pk", 5, pk.data_ptr(), 0, 0, 0, 0)], n, keep=pk)
    hiv = min(50_000_000, n // 14)
    lo = ColumnFilter.MakePredicateFilter(pc, keys, {"pk": np.array([1000], dtype=np.int64)}, 4)
    hi = ColumnFilter.MakePredicateFilter(pc, keys, {"pk": np.array([hiv], dtype=np.int64)}, 1)
    both = lo.And(hi)
    print(n, "lo count", lo.GetFilteredCount(), n - 7000, "hi count", hi.GetFilteredCount(), 7 * hiv, "and", both.GetFilteredCount(), 7 * (hiv - 1000))
    for name, f in (("lo", lo), ("hi", hi), ("and", both)):
        runs, last = f.Runs()
        print("  ", name, len(runs), runs[:6].tolist(), last)
    bits = both.BuildSimpleFilter()
    exp = np.zeros(n, dtype=bool); exp[7000:7 * hiv] = True
    bad = np.flatnonzero(bits != exp)
    print("   bits wrong at", len(bad), bad[:5])
    ref = torch.from_numpy(exp).to(dev)
    chk = (pk >= 1000) & (pk < hiv)
    print("   torch agrees with numpy:", bool((chk == ref).all().item()))

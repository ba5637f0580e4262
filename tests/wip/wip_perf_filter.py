"""TColumnFilter kernels at scale: predicate filter over a sorted int64 key, And, CombineSequentialAnd, runs, Apply."""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import time
import numpy as np, torch
from ydb_amd.program import ProgramContainer, device_batch
from ydb_amd.column_filter import ColumnFilter
from ydb_amd.ssa_pb import Program

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256_000_000
dev = torch.device("cuda", 0)
pk = torch.arange(n, dtype=torch.int64, device=dev) // 3
val = torch.randint(0, 100, (n,), dtype=torch.int32, device=dev)
payload = torch.rand(n, dtype=torch.float64, device=dev)
torch.cuda.synchronize()
pc = ProgramContainer()
ok, err = pc.Init([(1, "pk")], Program().project([1])); assert ok, err
b = device_batch([("pk", 5, pk.data_ptr(), 0, 0, 0, 0), ("val", 4, val.data_ptr(), 0, 0, 0, 0), ("payload", 11, payload.data_ptr(), 0, 0, 0, 0)], n, keep=(pk, val, payload))
def t(label, fn, nbytes):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{label}: {1e3*dt:.3f} ms, {nbytes/dt/1e9:.0f} GB/s of algorithmic bytes")
    return r
lo = t("predicate >= (8 B/row + 1 bit)", lambda: ColumnFilter.MakePredicateFilter(pc, b, {"pk": np.array([n // 12], dtype=np.int64)}, 4), n * 8 + n // 8)
hi = t("predicate <", lambda: ColumnFilter.MakePredicateFilter(pc, b, {"pk": np.array([n // 4], dtype=np.int64)}, 1), n * 8 + n // 8)
rng = t("and (3 bitmaps)", lambda: lo.And(hi), 3 * n // 8)
kept = rng.GetFilteredCount()
bits = torch.rand(kept, device=dev) < 0.3
ext = ColumnFilter.FromValues(pc, bits.cpu().numpy())
both = t("combine sequential and", lambda: rng.CombineSequentialAnd(ext), 2 * n // 8 + kept // 8)
t("runs", lambda: both.Runs(), n // 8)
out = t("apply (3 columns)", lambda: both.Apply(b), n // 8 + 3 * n // 8 + both.GetFilteredCount() * 20 * 2)
print("rows", n, "range keeps", kept, "final", out.rows, "runs", len(both.Runs()[0]))

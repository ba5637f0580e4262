"""Where the time of one Q1 step over an SF100-sized resident shard goes on the host's clock: the apply (scan, finalize,
their syncs) and the export of the result to the host, timed apart.
Run from the repository root: PYTHONPATH=. python scripts/step_phases.py [rows]"""
import sys
import time

import torch

from ydb_amd import tpch
from ydb_amd.shard import ShardedScan


def main():
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 600_037_902
    li = tpch.DeviceLineitem(rows, device=0, row0=0, utf8_flags=True, code_flags=False)
    scan = ShardedScan(tpch.q1_program(False), tpch.LINEITEM_RESOLVER, device=0)
    batch = li.batch(tpch.Q1_COLUMNS_UTF8)
    first = []
    for _ in range(8):  # (the first scans build the step's kernel and settle how it is run: their times one by one)
        t0 = time.perf_counter()
        scan.run(batch)
        first.append(1e3 * (time.perf_counter() - t0))
    print("the first scans, ms: " + " ".join(f"{x:.2f}" for x in first), flush=True)
    torch.cuda.synchronize()
    K = 10
    t_apply = t_export = t_run = 0.0
    for _ in range(K):
        t0 = time.perf_counter()
        r = scan.local.ApplyDevice(batch)
        t1 = time.perf_counter()
        scan.local.Export(r)
        t2 = time.perf_counter()
        t_apply += t1 - t0
        t_export += t2 - t1
    t0 = time.perf_counter()
    for _ in range(K):
        scan.run(batch)
    t_run = time.perf_counter() - t0
    print(f"rows {rows}: apply {1e3 * t_apply / K:.3f} ms, export {1e3 * t_export / K:.3f} ms, run() {1e3 * t_run / K:.3f} ms per step", flush=True)
    scan.local.EnableProfile(True)
    scan.local.ReadProfile()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        scan.run(batch)
    torch.cuda.synchronize()
    t_prof = time.perf_counter() - t0
    ms, n, _ = scan.local.ReadProfile()
    print(f"with the kernel timed by HIP events: run() {1e3 * t_prof / K:.3f} ms per step, the scan kernel {ms / max(1, n):.3f} ms", flush=True)


if __name__ == "__main__":
    main()

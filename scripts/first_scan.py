"""Time of a step's first scans over a hits-sized batch (the table is sized by trial: a scan that finds it too small
gives up and is repeated), then the steady state.  Usage: python scripts/first_scan.py cb_userid [scans]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from ydb_amd import clickbench  # noqa: E402
from ydb_amd.program import ProgramContainer  # noqa: E402


def main():
    workload = sys.argv[1]
    scans = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    rows = int(os.environ.get("ROWS", clickbench.HITS_ROWS))
    cols, _ = clickbench.COLUMNS[workload]
    hits = clickbench.DeviceHits(rows, columns=cols)
    batch = hits.batch(cols)
    pc = ProgramContainer(0)
    ok, err = pc.Init(clickbench.HITS_RESOLVER, clickbench.programs()[workload])
    assert ok, err
    for i in range(scans):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = pc.ApplyDevice(batch)
        torch.cuda.synchronize()
        print(f"{workload} scan {i + 1}: {1e3 * (time.perf_counter() - t0):.2f} ms, {out.rows} groups", flush=True)


if __name__ == "__main__":
    main()

"""Time the region-by-region GROUP BY pass (ClickBench UserID / SearchPhrase over a hits-sized batch) for a list of
settings of its environment switches.  Usage: python scripts/sweep_partition.py cb_userid "PARTS=256,APPLY=8" ..."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from ydb_amd import clickbench  # noqa: E402
from ydb_amd.program import ProgramContainer  # noqa: E402

NAMES = {"PARTS": "SSA_KEYED_PARTS", "APPLY": "SSA_KEYED_APPLY_BLOCKS", "ON": "SSA_KEYED_PARTITION", "LDS": "SSA_KEYED_LDS"}


def main():
    workload = sys.argv[1]
    rows = int(os.environ.get("ROWS", clickbench.HITS_ROWS))
    cols, _ = clickbench.COLUMNS[workload]
    hits = clickbench.DeviceHits(rows, columns=cols)
    batch = hits.batch(cols)
    pc = ProgramContainer(0)
    ok, err = pc.Init(clickbench.HITS_RESOLVER, clickbench.programs()[workload])
    assert ok, err
    for setting in sys.argv[2:] or ["ON=0"]:
        for v in NAMES.values():
            os.environ.pop(v, None)
        for kv in filter(None, setting.split(",")):
            k, v = kv.split("=")
            os.environ[NAMES[k]] = v.replace(":", ",")
        for _ in range(3):
            out = pc.ApplyDevice(batch)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            out = pc.ApplyDevice(batch)
        torch.cuda.synchronize()
        print(f"{workload} {setting or 'default'}: {1e3 * (time.perf_counter() - t0) / 5:.2f} ms, {out.rows} groups", flush=True)


if __name__ == "__main__":
    main()

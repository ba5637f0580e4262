"""Applies of shard-sized batches, one at a time: wall time per apply of a TPC-H program over `rows` rows of a resident
table.  Usage: python scripts/small_apply.py q1|q1_codes|q6 [rows] [applies]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from ydb_amd import tpch  # noqa: E402
from ydb_amd.program import ProgramContainer  # noqa: E402


def main():
    which = sys.argv[1]
    rows = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 20
    applies = int(sys.argv[3]) if len(sys.argv) > 3 else 200
    li = tpch.DeviceLineitem(rows, utf8_flags=which == "q1", code_flags=which == "q1_codes")
    prog, cols = {"q1": (tpch.q1_program(False), tpch.Q1_COLUMNS_UTF8), "q1_codes": (tpch.q1_program(True), tpch.Q1_COLUMNS_CODES),
                  "q6": (tpch.q6_program(), tpch.Q6_COLUMNS)}[which]
    pc = ProgramContainer(0)
    ok, err = pc.Init(tpch.LINEITEM_RESOLVER, prog)
    assert ok, err
    batch = li.batch(cols)
    os.environ["SSA_JIT"] = os.environ.get("SSA_JIT", "1")
    for _ in range(20):
        out = pc.ApplyDevice(batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(applies):
        out = pc.ApplyDevice(batch)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / applies
    print(f"{which}: {rows} rows, {1e6 * dt:.1f} us per apply, {rows / dt / 1e9:.2f} G rows/s, {out.rows} groups", flush=True)


if __name__ == "__main__":
    main()

"""Blob -> HBM-resident batch decode rate: lineitem-like portions of 2^20 rows written by pyarrow as IPC streams with
LZ4_FRAME bodies (what TNativeSerializer writes by default, ydb/core/formats/arrow/serializer/native.h:54-57), imported in
one ssa_batches_from_blobs call.  Reports decompressed GB/s (column bytes produced) and rows/s, next to pyarrow's own
single-thread read of the same blobs on the host.
Run from the repository root: PYTHONPATH=. python scripts/bench_blob.py [portions]
(BLOB_COLUMNS=l_quantity,l_tax keeps those columns only: which kind of column costs what.)"""
import io
import os
import sys
import time

import numpy as np
import pyarrow as pa
import pyarrow.ipc as ipc
import torch

from ydb_amd import tpch
from ydb_amd.program import ProgramContainer
from ydb_amd.ssa_pb import Program


def portion(row0, rows):
    li = tpch.lineitem_host(row0, rows)
    cols = {"l_shipdate": pa.array(li["l_shipdate"]), "l_quantity": pa.array(li["l_quantity"]), "l_extendedprice": pa.array(li["l_extendedprice"]),
            "l_discount": pa.array(li["l_discount"]), "l_tax": pa.array(li["l_tax"]),
            "l_returnflag": pa.array(list(li["l_returnflag"]), type=pa.utf8()), "l_linestatus": pa.array(list(li["l_linestatus"]), type=pa.utf8())}
    keep = os.environ.get("BLOB_COLUMNS")
    if keep:
        cols = {k: v for k, v in cols.items() if k in keep.split(",")}
    t = pa.table(cols)
    sink = io.BytesIO()
    with ipc.new_stream(sink, t.schema, options=ipc.IpcWriteOptions(compression="lz4")) as w:
        w.write_table(t, max_chunksize=rows)
    raw = sum(c.nbytes for c in t.columns)
    return sink.getvalue(), raw


def main():
    nportions = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    rows = 1 << 20
    blobs, raw = [], 0
    distinct = [portion(i * rows, rows) for i in range(4)]  # (generating on the host is the slow part: four portions, repeated)
    for i in range(nportions):
        b, r = distinct[i % 4]
        blobs.append(b)
        raw += r
    packed = sum(len(b) for b in blobs)
    pc = ProgramContainer()
    ok, err = pc.Init(tpch.LINEITEM_RESOLVER, Program().project([tpch.L_SHIPDATE]))
    assert ok, err
    for _ in range(2):
        out = pc.ImportBlobs(blobs)
    torch.cuda.synchronize()
    pc.EnableProfile(True)
    pc.ReadProfile()
    t0 = time.perf_counter()
    K = 5
    for _ in range(K):
        out = pc.ImportBlobs(blobs)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    kernel_ms, launches, _ = pc.ReadProfile()
    pc.EnableProfile(False)
    print(f"decode kernel: {kernel_ms / max(1, launches):.1f} ms per import (HIP events)", flush=True)
    print(f"{nportions} portions x {rows} rows: {packed / 1e6:.1f} MB stored -> {raw / 1e6:.1f} MB of columns; "
          f"{dt * 1e3:.2f} ms per import (host copy of the blobs included), {raw / dt / 1e9:.1f} GB/s of columns, {nportions * rows / dt / 1e9:.2f} G rows/s", flush=True)
    t0 = time.perf_counter()
    for b in blobs[:8]:
        ipc.open_stream(b).read_all()
    hdt = (time.perf_counter() - t0) / 8 * nportions
    print(f"pyarrow on one host thread: {hdt * 1e3:.1f} ms for the same blobs, {raw / hdt / 1e9:.2f} GB/s of columns", flush=True)
    del out


if __name__ == "__main__":
    main()

"""One rank of a rehearsal of the multi-GPU path on a single GPU: every rank uses cuda:0, the process group is gloo, so the
packed partial tables cross the host on their way (on a node with one GPU per rank the same code hands RCCL the device
buffers).  Each rank scans its own row range, the ranks exchange and merge, and every rank writes what it ends up with
to `out_dir`; the parent test compares the union with a scan of the whole table.

usage (under torch.distributed.run): rehearse_exchange.py <out_dir> <rows_per_rank>"""
import os
import pickle
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))

from ydb_amd import tpch  # noqa: E402
from ydb_amd.shard import ShardedScan  # noqa: E402
from ydb_amd.ssa_pb import AGG, F, Aggregate, Program  # noqa: E402


def programs():
    many_groups = (Program(version=5).assign_const(100, "Int32", tpch.Q1_DATE).assign_func(110, F.CMP_LESS_EQUAL, [tpch.L_SHIPDATE, 100]).filter(110)
                   .group_by([Aggregate(130, AGG.SUM, [tpch.L_QUANTITY]), Aggregate(131, AGG.COUNT, []), Aggregate(132, AGG.MIN, [tpch.L_DISCOUNT]),
                              Aggregate(133, AGG.MAX, [tpch.L_TAX])],
                             keys=[tpch.L_SHIPDATE, tpch.L_EXTENDEDPRICE, tpch.L_RETURNFLAG]))
    cols = ["l_shipdate", "l_quantity", "l_extendedprice", "l_discount", "l_tax", "l_returnflag", "l_linestatus"]
    return {"q1": (tpch.q1_program(False), tpch.Q1_COLUMNS_UTF8), "q6": (tpch.q6_program(), tpch.Q6_COLUMNS), "many_groups": (many_groups, cols)}


def main():
    out_dir, rows = sys.argv[1], int(sys.argv[2])
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cpu = torch.device("cpu")
    li = tpch.DeviceLineitem(rows, device=0, row0=rank * rows)
    out = {}
    for name, (prog, cols) in programs().items():
        scan = ShardedScan(prog, tpch.LINEITEM_RESOLVER, device=0)
        batch = li.batch(cols)
        for exchange in (("allgather", "alltoall", "auto") if name != "q6" else ("allgather",)):
            res = scan.run(batch, gather_device=cpu, exchange=exchange)
            out[(name, exchange)] = {k: (c.type, list(c.values) if isinstance(c.values, list) else np.asarray(c.values), np.asarray(c.valid)) for k, c in res.items()}
    # the mixed stream of bench.py (--workload mixed): Q1 and Q6 alternate over one resident shard, each through its own
    # scan object whose kernels, tables and exchange buffers live across the steps
    stream = {name: (ShardedScan(prog, tpch.LINEITEM_RESOLVER, device=0), li.batch(cols)) for name, (prog, cols) in programs().items() if name != "many_groups"}
    for step in range(3):
        for name, (scan, batch) in stream.items():
            res = scan.run(batch, gather_device=cpu)
            out[("mixed", step, name)] = {k: (c.type, list(c.values) if isinstance(c.values, list) else np.asarray(c.values), np.asarray(c.valid)) for k, c in res.items()}
    with open(os.path.join(out_dir, f"rank{rank}.pkl"), "wb") as f:
        pickle.dump(out, f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()

"""One rank over the RCCL backend on one GPU: the device exchange of a sharded scan (headers, all-gather, all-to-all by
key hash, concatenate, merge program) with the collectives run on device buffers, checked against the plain scan.
Child process of tests/test_gpu_exchange.py (a process group lives as long as its process)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cases import compare_batches  # noqa: E402
from ydb_amd import tpch  # noqa: E402
from ydb_amd.shard import ShardedScan  # noqa: E402

import rehearse_exchange  # noqa: E402


def main():
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    rows = int(sys.argv[1])
    li = tpch.DeviceLineitem(rows)
    checked = 0
    for name, (prog, cols) in rehearse_exchange.programs().items():
        scan = ShardedScan(prog, tpch.LINEITEM_RESOLVER)
        batch = li.batch(cols)
        want = scan.local.Export(scan.local.ApplyDevice(batch))
        key_names = [k for k in want if not k.isdigit()]
        for exchange in (("allgather", "alltoall", "auto") if key_names else ("allgather",)):
            partial = scan.head.ApplyDevice(batch)
            merged = scan.merge.ApplyDevice(scan.exchange_device(partial, None, exchange))
            got = scan.merge.Export(merged)
            floats = [k for k, c in want.items() if k.isdigit() and np.asarray(c.values).dtype.kind == "f"]
            compare_batches(got, want, key_names=key_names, float_sum_names=floats)
            checked += 1
    dist.barrier()
    dist.destroy_process_group()
    print(f"rccl one rank: {checked} exchanges equal the plain scan", flush=True)


if __name__ == "__main__":
    main()

import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import time
import numpy as np, torch
from ydb_amd.program import ProgramContainer
from ydb_amd import tpch

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64_000_000
shards = int(sys.argv[2]) if len(sys.argv) > 2 else 8
li = tpch.DeviceLineitem(n, utf8_flags=True)
pc = ProgramContainer()
ok, err = pc.Init(tpch.LINEITEM_RESOLVER, tpch.q6_program()); assert ok, err
cols = ["l_shipdate", "l_quantity", "l_extendedprice", "l_discount"]
dev = li.batch(cols)
for keys in (["l_shipdate", "l_quantity"], ["l_shipdate"]):
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.time()
        h = pc.ShardingHash(dev, keys, 34323543)
        torch.cuda.synchronize(); t1 = time.time()
        out, sizes = pc.ShardingSplit(dev, keys, shards)
        torch.cuda.synchronize(); t2 = time.time()
    print(f"keys {keys}: hash+copy {1e3*(t1-t0):.2f} ms, split {1e3*(t2-t1):.2f} ms ({n/(t2-t1)/1e9:.2f} Grows/s), sizes {sizes[:4]}...")

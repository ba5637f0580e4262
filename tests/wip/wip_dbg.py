import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from ydb_amd import tpch
li = tpch.DeviceLineitem(4096, utf8_flags=True, code_flags=True)
h = tpch.lineitem_host(0, 4096)
for k in ("l_shipdate", "l_quantity", "l_extendedprice", "l_discount", "l_tax"):
    d = li.t[k][:4096].cpu().numpy()
    bad = np.nonzero(d != h[k])[0]
    print(k, len(bad), [(int(i), d[i], h[k][i]) for i in bad[:5]])
print(li.t['rf_chars'][:16].cpu().numpy().tobytes(), h['l_returnflag'][:16])
print(li.t['ls_chars'][:16].cpu().numpy().tobytes(), h['l_linestatus'][:16])
print((li.t['rf_code'].cpu().numpy()==h['l_returnflag_code']).all(), (li.t['ls_code'].cpu().numpy()==h['l_linestatus_code']).all())

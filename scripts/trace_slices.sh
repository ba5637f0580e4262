#!/bin/bash
# Kernel trace of ssa_program_apply_many over 64 slices of 2^20 rows (Q1 utf8): the kernels of the last step with start times.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/trace_slices
mkdir -p $O
cat > /tmp/slices_once.py <<PY
import sys, time
sys.path.insert(0, "$R")
import torch
from ydb_amd import tpch
from ydb_amd.program import ProgramContainer
per, nslices = 1 << 20, 64
li = tpch.DeviceLineitem(per * nslices)
pc = ProgramContainer()
ok, err = pc.Init(tpch.LINEITEM_RESOLVER, tpch.q1_program(False))
slices = [li.batch(tpch.Q1_COLUMNS_UTF8, row0=i * per, rows=per) for i in range(nslices)]
for _ in range(6):
    res = pc.ApplyDeviceMany(slices)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    res = pc.ApplyDeviceMany(slices)
torch.cuda.synchronize()
print("ms per step", (time.perf_counter() - t0) / 20 * 1e3)
PY
python3 /tmp/slices_once.py
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $O -o t -- python3 /tmp/slices_once.py > $O/out.txt 2> $O/err.txt
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$O/t_kernel_trace.csv")))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
rows=[r for r in rows if not r['Kernel_Name'].startswith('void at::')]
scans=[i for i,r in enumerate(rows) if r['Kernel_Name'].startswith('ssa_spec')]
start=scans[-2]+1
t0=int(rows[start]['Start_Timestamp'])
for r in rows[start:]:
    d=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
    print(f"{(int(r['Start_Timestamp'])-t0)/1e3:9.1f} us +{d:7.1f} us  {r['Kernel_Name'][:70]}")
PY

#!/bin/bash
# Kernel trace of single applies of a shard-sized batch: the kernels of the last apply with their start times.
# Usage: bash scripts/trace_small.sh q1|q1_codes|q6 [rows]
W=$1; ROWS=${2:-1048576}
R=${GRAFT_REPO_ROOT:-$(pwd)}
python3 $R/scripts/small_apply.py $W $ROWS 200
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/trace_small_$W
mkdir -p $O
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $O -o t -- python3 $R/scripts/small_apply.py $W $ROWS 30 > $O/out.txt 2> $O/err.txt
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$O/t_kernel_trace.csv")))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
rows=[r for r in rows if not r['Kernel_Name'].startswith('void at::')]
n=len(rows)
# the last apply: kernels after the second to last scan kernel
scans=[i for i,r in enumerate(rows) if r['Kernel_Name'].startswith('ssa_spec') or 'Scan' in r['Kernel_Name'] and 'Kernel<' not in r['Kernel_Name'] and 'ScanKeyed' in r['Kernel_Name']]
start=scans[-2]+1 if len(scans)>1 else 0
t0=int(rows[start]['Start_Timestamp'])
for r in rows[start:]:
    d=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
    print(f"{(int(r['Start_Timestamp'])-t0)/1e3:9.1f} us +{d:7.1f} us  {r['Kernel_Name'][:70]}")
PY

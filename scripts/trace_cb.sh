#!/bin/bash
# Kernel trace of the steady state of one ClickBench grouping.  Usage: bash scripts/trace_cb.sh <workload> <tag> [ENV=VALUE ...]
W=$1; TAG=$2; shift; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/trace_$TAG
mkdir -p $O
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o t -- python3 $R/scripts/first_scan.py $W 8 > $O/out.txt 2> $O/err.txt
tail -3 $O/out.txt
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$O/t_kernel_trace.csv")))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
rows=[r for r in rows if not r['Kernel_Name'].startswith('void at::') and 'rocclr' not in r['Kernel_Name']]
# the last step: from the last KeyedTableInitKernel on
idx=[i for i,r in enumerate(rows) if 'KeyedTableInit' in r['Kernel_Name']]
for r in rows[idx[-1]:]:
    d=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6
    print(f"{d:9.3f} ms grid {r['Grid_Size_X']:>8s} lds {r['LDS_Block_Size']:>6s} vgpr {r['VGPR_Count']:>4s} {r['Kernel_Name'][:60]}")
PY

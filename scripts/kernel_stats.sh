#!/bin/bash
# rocprofv3 kernel statistics of one bench workload: bash scripts/kernel_stats.sh <workload> [bench.py arguments...]
# (run on the GPU box from the repository root; the table lands in gpurun_out/ks_<workload>/ and its head on stdout)
set -e
W=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/ks_$W
mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o $W -- python3 $R/bench.py --steps 5 --warmup 3 --workload $W "$@" --no-cpu-baseline --no-also > $O/bench_line.json 2> $O/err.txt
python3 - "$O/${W}_kernel_stats.csv" <<'PY'
import csv, sys
for i, r in enumerate(csv.reader(open(sys.argv[1]))):
    if i == 0 or "at::native" in r[0] or "rocprim" in r[0]: continue
    if i > 16: break
    print("%-60s calls %4s avg %10.1f us min %10.1f us" % (r[0].replace("ssa::(anonymous namespace)::", "")[:60], r[1], float(r[3]) / 1e3, float(r[5]) / 1e3))
PY
